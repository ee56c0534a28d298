"""Seeded random configurations through the C ABI against the oracle, all three quantizers:
shapes the hand-picked cases do not list (odd dims, tiny and ragged stores, every distance /
invert combination), encode -> rows -> query -> score_all / score_ids / top-k, bit for bit."""
import numpy as np
import pytest

import quantization_amd as qa
from util import assert_bits_equal

pytestmark = pytest.mark.gpu
D = qa.DistanceType


def _topk_want(scores, k, largest):
    n = scores.size
    order = np.lexsort((np.arange(n), -scores if largest else scores))[: min(k, n)]
    return order.astype(np.uint32), scores[order]


def test_fuzz_u8(qo):
    rng = np.random.default_rng(20261004)
    for case in range(40):
        n = int(rng.integers(1, 3000))
        dim = int(rng.integers(1, 260))
        dist = [D.Dot, D.L1, D.L2][int(rng.integers(0, 3))]
        invert = bool(rng.integers(0, 2))
        scale, shift = float(rng.choice([1.0, 1e-3, 50.0])), float(rng.choice([0.0, -0.5, 3.0]))
        data = (rng.random((n, dim), dtype=np.float32) + np.float32(shift)) * np.float32(scale)
        query = (rng.random(dim, dtype=np.float32) + np.float32(shift)) * np.float32(scale)
        tag = f"case {case}: n={n} dim={dim} {dist.name} invert={invert} scale={scale} shift={shift}"
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
        rows, meta = qo.u8_encode(data, int(dist), invert)
        assert np.array_equal(enc.storage_bytes(), rows), tag + " rows"
        q = enc.encode_query(query)
        codes, qoff = qo.u8_encode_query(meta, query)
        assert np.array_equal(q.encoded_query, codes), tag + " query codes"
        want = qo.u8_score_all(meta, rows, codes, qoff)
        assert_bits_equal(enc.score_all(q), want, tag + " score_all")
        ids = rng.integers(0, n, size=min(n, 17)).astype(np.uint32)
        assert_bits_equal(enc.score_ids(q, ids), want[ids], tag + " score_ids")
        k = int(rng.integers(1, 40))
        largest = bool(rng.integers(0, 2))
        gi, gs = enc.topk(q, k, largest=largest)
        wi, ws = _topk_want(want, k, largest)
        assert np.array_equal(gi[: wi.size], wi), tag + " topk ids"
        assert_bits_equal(gs[: wi.size], ws, tag + " topk scores")
        i, j = int(rng.integers(0, n)), int(rng.integers(0, n))
        assert_bits_equal([enc.score_internal(i, j)], [qo.u8_score_internal(meta, rows, i, j)], tag + " score_internal")


def test_fuzz_binary(qo):
    rng = np.random.default_rng(77)
    S = qa.BitsStoreType if hasattr(qa, "BitsStoreType") else None
    for case in range(30):
        n = int(rng.integers(1, 4000))
        dim = int(rng.integers(1, 700))
        dist = [D.Dot, D.L1, D.L2][int(rng.integers(0, 3))]
        invert = bool(rng.integers(0, 2))
        data = rng.standard_normal((n, dim)).astype(np.float32)
        query = rng.standard_normal(dim).astype(np.float32)
        tag = f"case {case}: n={n} dim={dim} {dist.name} invert={invert}"
        enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, dist, invert))
        rows = qo.bin_encode(data)
        assert np.array_equal(enc.storage_bytes(), rows), tag + " rows"
        q = enc.encode_query(query)
        qbits = qo.bin_encode(query[None, :])[0]
        want = qo.bin_score_all(rows, qbits, dim, int(dist), invert)
        assert_bits_equal(enc.score_all(q), want, tag + " score_all")
        k = int(rng.integers(1, 30))
        gi, gs = enc.topk(q, k, largest=True)
        wi, ws = _topk_want(want, k, True)
        assert np.array_equal(gi[: wi.size], wi), tag + " topk ids (ties to the lower id)"
        assert_bits_equal(gs[: wi.size], ws, tag + " topk scores")


def test_fuzz_pq(qo):
    rng = np.random.default_rng(4242)
    for case in range(24):
        n = int(rng.integers(1, 2500))
        dim = int(rng.integers(1, 200))
        chunk = int(rng.integers(1, min(dim, 32) + 1))
        dist = [D.Dot, D.L1, D.L2][int(rng.integers(0, 3))]
        invert = bool(rng.integers(0, 2))
        data = rng.random((n, dim), dtype=np.float32)
        query = rng.random(dim, dtype=np.float32)
        cen = rng.random((256, dim), dtype=np.float32)
        tag = f"case {case}: n={n} dim={dim} chunk={chunk} {dist.name} invert={invert}"
        enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, dist, invert), chunk, centroids=cen)
        rows = qo.pq_encode(data, chunk, cen)
        assert np.array_equal(enc.storage_bytes(), rows), tag + " codes"
        q = enc.encode_query(query)
        lut = qo.pq_encode_query(query, chunk, cen, int(dist), invert)
        assert_bits_equal(q.lut, lut, tag + " LUT")
        want = qo.pq_score_all(rows, lut, order=qo.ORDER_SSE)
        assert_bits_equal(enc.score_all(q), want, tag + " score_all")
        k = int(rng.integers(1, 30))
        gi, gs = enc.topk(q, k, largest=False)
        wi, ws = _topk_want(want, k, False)
        assert np.array_equal(gi[: wi.size], wi), tag + " topk ids"
        assert_bits_equal(gs[: wi.size], ws, tag + " topk scores")


def test_fuzz_u8_batch():
    """score_batch / topk_batch (both MFMA kernels and both tile shapes, chosen by nq and dim)
    against the single-query path on random shapes."""
    rng = np.random.default_rng(31337)
    for case in range(30):
        n = int(rng.integers(1, 3000))
        dim = int(rng.integers(1, 600))
        nq = int(rng.choice([1, 2, 7, 64, 128, 129, 200, 257, 400]))
        dist = [D.Dot, D.L2][int(rng.integers(0, 2))]
        invert = bool(rng.integers(0, 2))
        data = rng.random((n, dim), dtype=np.float32) - np.float32(rng.choice([0.0, 0.4]))
        queries = rng.random((nq, dim), dtype=np.float32)
        tag = f"case {case}: n={n} dim={dim} nq={nq} {dist.name} invert={invert}"
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
        b = enc.encode_query_batch(queries)
        got = enc.score_batch(b)
        k = int(rng.integers(1, 20))
        largest = bool(rng.integers(0, 2))
        ids, sc = enc.topk_batch(b, k, largest=largest)
        for qi in sorted({0, nq // 2, nq - 1}):
            want = enc.score_all(enc.encode_query(queries[qi]))
            assert_bits_equal(got[qi], want, tag + f" scores of query {qi}")
            wi, ws = _topk_want(want, k, largest)
            assert np.array_equal(ids[qi][: wi.size], wi), tag + f" topk ids of query {qi}"
            assert_bits_equal(sc[qi][: wi.size], ws, tag + f" topk scores of query {qi}")
