"""Bursts of random-access pairs in one launch (csrc/lists.hpp): `score_internal_ids`,
`score_internal_ids_batch`, `score_ids_batch` for the three quantizers.  Every score must equal the
reference's per-pair call restated by the oracle — `score_point` (encoded_vectors_u8.rs:331-384,
encoded_vectors_pq.rs:549-561, encoded_vectors_binary.rs:293-300) and `score_internal`
(encoded_vectors_u8.rs:386-453, encoded_vectors_pq.rs:566-593, encoded_vectors_binary.rs:302-314) —
bit for bit, for host lists (mapped-scratch and staged paths) and device lists, ragged and empty lists.
Also the ranged row export (`storage_rows`), the caller-owned-storage half of `encode`."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
torch = pytest.importorskip("torch")
D = qa.DistanceType


def make_lists(rng, n_lists, count, max_len, empty_every=5):
    lens = rng.integers(1, max_len + 1, n_lists)
    lens[::empty_every] = 0  # empty lists are legal
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    ids = rng.integers(0, count, int(offs[-1])).astype(np.uint32)
    return offs, ids


def list_of(offs, p):
    return int(np.searchsorted(offs, p, side="right") - 1)


def both_ways(call, offs, ids, rows=None):
    """The same burst through host lists (numpy) and device lists (CUDA tensors, device output)."""
    host = call(offs, ids, rows, None)
    t = lambda a: torch.from_numpy(a.view(np.int32)).cuda()
    out = torch.empty(ids.size, dtype=torch.float32, device="cuda")
    call(t(offs), t(ids), t(rows) if rows is not None else None, out)
    torch.cuda.synchronize()
    assert_bits_equal(out.cpu().numpy(), host, "device lists vs host lists")
    return host


@pytest.mark.parametrize("dim", [65, 768, 1536])
@pytest.mark.parametrize("dist", [D.Dot, D.L1, D.L2])
@pytest.mark.parametrize("invert", [False, True])
def test_u8_bursts_equal_the_per_pair_reference_calls(dim, dist, invert, qo):
    rng = np.random.default_rng(dim * 7 + int(dist) * 3 + invert)
    n = 3000
    data = rng.random((n, dim), dtype=np.float32) - (0.5 if dist == D.L1 else 0.0)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    rows, meta = qo.u8_encode(data, int(dist), invert)
    assert np.array_equal(enc.storage_bytes(), rows)
    order = qo.ORDER_SIMPLE
    # score_internal_ids: one row against many (few ids -> mapped scratch; many -> staged buffers)
    for n_ids in (1, 37, 1500):
        ids = rng.integers(0, n, n_ids).astype(np.uint32)
        got = enc.score_internal_ids(17, ids)
        want = np.array([qo.u8_score_internal(meta, rows, 17, int(j), order) for j in ids[:200]], dtype=np.float32)
        assert_bits_equal(got[:200], want, f"score_internal_ids n={n_ids}")
        d_out = torch.empty(n_ids, dtype=torch.float32, device="cuda")
        enc.score_internal_ids(17, torch.from_numpy(ids.view(np.int32)).cuda(), out=d_out)
        assert_bits_equal(d_out.cpu().numpy(), got, "device ids")
    assert np.float32(enc.score_internal(17, int(ids[0]))).view(np.uint32) == got[:1].view(np.uint32)[0]
    # score_internal_ids_batch: many rows, each with its own list
    for n_lists, max_len in ((9, 12), (64, 40)):
        offs, ids = make_lists(rng, n_lists, n, max_len)
        qrows = rng.integers(0, n, n_lists).astype(np.uint32)
        got = both_ways(lambda o, i, r, out: enc.score_internal_ids_batch(r, o, i, out=out), offs, ids, qrows)
        pick = rng.integers(0, ids.size, min(300, ids.size))
        want = np.array([qo.u8_score_internal(meta, rows, int(qrows[list_of(offs, p)]), int(ids[p]), order) for p in pick],
                        dtype=np.float32)
        assert_bits_equal(got[pick], want, "score_internal_ids_batch")
    # score_ids_batch: query l against list l
    for n_lists, max_len in ((7, 9), (64, 32), (200, 50)):
        offs, ids = make_lists(rng, n_lists, n, max_len)
        queries = rng.random((n_lists, dim), dtype=np.float32)
        batch = enc.encode_query_batch(queries)
        got = both_ways(lambda o, i, r, out: enc.score_ids_batch(batch, o, i, out=out), offs, ids)
        pick = rng.integers(0, ids.size, min(300, ids.size))
        enc_q = {}
        want = []
        for p in pick:
            l = list_of(offs, p)
            if l not in enc_q:
                enc_q[l] = qo.u8_encode_query(meta, queries[l])
            codes, qoff = enc_q[l]
            want.append(qo.u8_score_point(meta, rows, codes, qoff, int(ids[p]), order))
        assert_bits_equal(got[pick], np.array(want, dtype=np.float32), "score_ids_batch")
        # and it is what looping score_point gives
        q3 = enc.encode_query(queries[list_of(offs, pick[0])])
        assert np.float32(enc.score_point(q3, int(ids[pick[0]]))).view(np.uint32) == got[pick[:1]].view(np.uint32)[0]


@pytest.mark.parametrize("dim", [387, 1024, 20])
@pytest.mark.parametrize("dist,invert", [(D.Dot, False), (D.Dot, True), (D.L1, False), (D.L2, True)])
def test_binary_bursts_equal_the_per_pair_reference_calls(dim, dist, invert, qo):
    rng = np.random.default_rng(dim + int(dist) * 11 + invert)
    n = 4000
    data = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, dist, invert))
    rows = qo.bin_encode(data)
    assert np.array_equal(enc.storage_bytes(), rows)
    ids = rng.integers(0, n, 1100).astype(np.uint32)
    got = enc.score_internal_ids(5, ids)
    want = np.array([qo.bin_score_internal(rows, dim, int(dist), invert, 5, int(j)) for j in ids[:300]], dtype=np.float32)
    assert_bits_equal(got[:300], want, "bin score_internal_ids")
    if dist == D.Dot and not invert:  # the reference's known-answer property: +-1 data -> the exact f32 dot
        assert np.array_equal(got, (data[ids] @ data[5]).astype(np.float32))
    offs, ids = make_lists(rng, 64, n, 32)
    qrows = rng.integers(0, n, 64).astype(np.uint32)
    got = both_ways(lambda o, i, r, out: enc.score_internal_ids_batch(r, o, i, out=out), offs, ids, qrows)
    want = np.array([qo.bin_score_internal(rows, dim, int(dist), invert, int(qrows[list_of(offs, p)]), int(ids[p]))
                     for p in range(ids.size)], dtype=np.float32)
    assert_bits_equal(got, want, "bin score_internal_ids_batch")
    queries = np.where(rng.random((64, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    batch = enc.encode_query_batch(queries)
    got = both_ways(lambda o, i, r, out: enc.score_ids_batch(batch, o, i, out=out), offs, ids)
    qbits = qo.bin_encode(queries)
    want = np.empty(ids.size, dtype=np.float32)
    for l in range(64):
        sl = slice(int(offs[l]), int(offs[l + 1]))
        if sl.stop > sl.start:
            want[sl] = qo.bin_score_all(rows[ids[sl]], qbits[l], dim, int(dist), invert)
    assert_bits_equal(got, want, "bin score_ids_batch")


@pytest.mark.parametrize("dim,chunk", [(768, 8), (1536, 8), (65, 1), (100, 7)])
@pytest.mark.parametrize("dist,invert", [(D.Dot, False), (D.L2, False), (D.L1, True)])
def test_pq_bursts_equal_the_per_pair_reference_calls(dim, chunk, dist, invert, qo):
    rng = np.random.default_rng(dim + chunk + int(dist))
    n = 2500
    data = rng.random((n, dim), dtype=np.float32)
    cen = rng.random((256, dim), dtype=np.float32)
    enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, dist, invert), chunk, centroids=cen)
    rows = qo.pq_encode(data, chunk, cen)
    assert np.array_equal(enc.storage_bytes(), rows)
    ids = rng.integers(0, n, 300).astype(np.uint32)
    got = enc.score_internal_ids(9, ids)
    want = np.array([qo.pq_score_internal(rows, dim, chunk, cen, int(dist), invert, 9, int(j)) for j in ids], dtype=np.float32)
    assert_bits_equal(got, want, "pq score_internal_ids")
    assert np.float32(enc.score_internal(9, int(ids[3]))).view(np.uint32) == got[3:4].view(np.uint32)[0]
    d_out = torch.empty(300, dtype=torch.float32, device="cuda")
    enc.score_internal_ids(9, torch.from_numpy(ids.view(np.int32)).cuda(), out=d_out)
    assert_bits_equal(d_out.cpu().numpy(), got, "pq device ids")
    offs, ids = make_lists(rng, 40, n, 20)
    qrows = rng.integers(0, n, 40).astype(np.uint32)
    got = both_ways(lambda o, i, r, out: enc.score_internal_ids_batch(r, o, i, out=out), offs, ids, qrows)
    want = np.array([qo.pq_score_internal(rows, dim, chunk, cen, int(dist), invert, int(qrows[list_of(offs, p)]), int(ids[p]))
                     for p in range(ids.size)], dtype=np.float32)
    assert_bits_equal(got, want, "pq score_internal_ids_batch")
    queries = rng.random((40, dim), dtype=np.float32)
    batch = enc.encode_query_batch(queries)
    got = both_ways(lambda o, i, r, out: enc.score_ids_batch(batch, o, i, out=out), offs, ids)
    want = np.empty(ids.size, dtype=np.float32)
    for l in range(40):
        sl = slice(int(offs[l]), int(offs[l + 1]))
        if sl.stop > sl.start:
            lut = qo.pq_encode_query(queries[l], chunk, cen, int(dist), invert)
            want[sl] = qo.pq_score_all(rows[ids[sl]], lut, order=qo.ORDER_SSE)
    assert_bits_equal(got, want, "pq score_ids_batch")


@pytest.mark.parametrize("dim,chunk", [(768, 8), (1536, 8), (130, 2), (100, 7)])
def test_pq_bursts_with_long_lists_stage_the_lut_once_per_segment(dim, chunk, qo):
    """Long lists (the reference's PQ bench shape, demos/benches/pq.rs:12-46: a query against many random rows) take
    pq_lists_staged_kernel: a workgroup stages a list's LUT in LDS once per segment of >= 128 pairs and gathers the shorter
    segments through the caches; list boundaries fall inside, on and across the 1024-pair workgroup ranges, empty lists and
    an out-of-range id (device lists: NaN) included.  Same bits as the per-pair score_point_sse order."""
    rng = np.random.default_rng(dim * 7 + chunk)
    n = 3000
    m = -(-dim // chunk)
    cen = (rng.random((256, dim), dtype=np.float32) - 0.5).astype(np.float32)
    rows = rng.integers(0, 256, size=(n, m), dtype=np.uint8)
    enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, D.L2, True), chunk, cen)
    lens = np.array([3000, 0, 130, 127, 1500, 1, 0, 700, 1024, 1024, 4000, 128, 5], dtype=np.uint32)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    ids = rng.integers(0, n, int(offs[-1])).astype(np.uint32)
    queries = (rng.random((len(lens), dim), dtype=np.float32) - 0.5).astype(np.float32)
    batch = enc.encode_query_batch(queries)
    want = np.empty(ids.size, dtype=np.float32)
    for l in range(len(lens)):
        sl = slice(int(offs[l]), int(offs[l + 1]))
        if sl.stop > sl.start:
            lut = qo.pq_encode_query(queries[l], chunk, cen, qo.L2, True)
            want[sl] = qo.pq_score_all(rows[ids[sl]], lut, order=qo.ORDER_SSE)
    assert_bits_equal(both_ways(lambda o, i, r, out: enc.score_ids_batch(batch, o, i, out=out), offs, ids), want,
                      "pq score_ids_batch, long lists")
    bad = ids.copy()
    bad[4000] = n + 7
    t = lambda a: torch.from_numpy(a.view(np.int32)).cuda()
    out = torch.empty(ids.size, dtype=torch.float32, device="cuda")
    enc.score_ids_batch(batch, t(offs), t(bad), out=out)
    got = out.cpu().numpy()
    assert np.isnan(got[4000])
    keep = np.arange(ids.size) != 4000
    assert_bits_equal(got[keep], want[keep], "device lists with one id out of range")


def test_burst_argument_errors():
    rng = np.random.default_rng(0)
    n, dim = 500, 32
    enc = qa.EncodedVectorsU8.encode(rng.random((n, dim), dtype=np.float32), qa.VectorParameters(dim, n, D.Dot, False))
    batch = enc.encode_query_batch(rng.random((3, dim), dtype=np.float32))
    offs = np.array([0, 2, 4], dtype=np.uint32)
    ids = np.array([1, 2, 3, 4], dtype=np.uint32)
    assert enc.score_ids_batch(batch, offs, ids).shape == (4,)
    with pytest.raises(IndexError):  # an id past the store: the reference panics on the slice index
        enc.score_ids_batch(batch, offs, np.array([1, 2, 3, n], dtype=np.uint32))
    with pytest.raises(qa.EncodingError):  # offsets must end at n_ids
        enc.score_ids_batch(batch, np.array([0, 2, 3], dtype=np.uint32), ids)
    with pytest.raises(qa.EncodingError):  # more lists than queries
        enc.score_ids_batch(batch, np.array([0, 1, 2, 3, 4], dtype=np.uint32), ids)
    with pytest.raises(IndexError):
        enc.score_internal_ids(n, ids)
    with pytest.raises(IndexError):
        enc.score_internal_ids_batch(np.array([0, n], dtype=np.uint32), offs, ids)
    # device lists cannot be validated: an id out of range scores NaN
    t = lambda a: torch.from_numpy(a.view(np.int32)).cuda()
    out = torch.empty(4, dtype=torch.float32, device="cuda")
    enc.score_ids_batch(batch, t(offs), t(np.array([1, 2, n + 5, 4], dtype=np.uint32)), out=out)
    got = out.cpu().numpy()
    assert np.isnan(got[2]) and not np.isnan(got[[0, 1, 3]]).any()
    assert enc.score_ids_batch(batch, np.array([0, 0, 0], dtype=np.uint32), np.zeros(0, dtype=np.uint32)).size == 0


def test_ranged_row_export_is_the_whole_export_in_pieces(qo):
    """storage_rows(first, n) = rows [first, first + n) of storage_bytes(): how a binding feeds the reference's
    storage_builder.push_vector_data (encoded_storage.rs:17-25) without holding the whole store."""
    rng = np.random.default_rng(3)
    n = 10_000
    data = rng.random((n, 100), dtype=np.float32)
    cen = rng.random((256, 100), dtype=np.float32)
    stores = [
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(100, n, D.L2, False)),
        qa.EncodedVectorsBin.encode(data - 0.5, qa.VectorParameters(100, n, D.Dot, False)),
        qa.EncodedVectorsBin.encode(data[:, :20] - 0.5, qa.VectorParameters(20, n, D.Dot, False)),  # 4-byte device stride
        qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(100, n, D.Dot, False), 7, centroids=cen),  # m = 15: re-strided
        qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(100, n, D.Dot, False), 5, centroids=cen),  # m = 20
    ]
    rows_u8, _ = qo.u8_encode(data, qo.L2, False)
    for enc in stores:
        whole = enc.storage_bytes()
        pieces = [enc.storage_rows(r0, min(1337, n - r0)) for r0 in range(0, n, 1337)]
        assert np.array_equal(np.concatenate(pieces), whole), type(enc).__name__
        dev = torch.empty(whole[100:300].size, dtype=torch.uint8, device="cuda")
        enc.storage_rows(100, 200, out=dev)
        assert np.array_equal(dev.cpu().numpy().reshape(200, -1), whole[100:300])
        assert enc.storage_rows(n, 0).size == 0
        with pytest.raises(IndexError):
            enc.storage_rows(n - 1, 2)
    assert np.array_equal(stores[0].storage_bytes(), rows_u8)
