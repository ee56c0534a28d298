"""HIP u8 path vs the oracle, through the C ABI (python mirror = thin ctypes).

Bit-exact everywhere: codes, vector offsets, metadata, query codes/offset and final f32 scores.
Mirrors quantization/tests/test_simple.rs, test_avx2.rs, empty_storage.rs, stop_condition.rs.
"""
import os

import numpy as np
import pytest

from util import assert_bits_equal, bits

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
D = qa.DistanceType

SHAPES = [(129, 65), (1000, 128), (257, 768), (70, 1536), (33, 17), (5, 1), (300, 16), (64, 100)]


def _data(n, dim, seed=42, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    return (rng.random((n, dim), dtype=np.float32) * np.float32(hi - lo) + np.float32(lo)), \
        (rng.random(dim, dtype=np.float32) * np.float32(hi - lo) + np.float32(lo))


def _order_for(qo, dim):
    # For actual_dim <= 1040 every reference kernel returns the same exact integer; above,
    # the default GPU mode equals the scalar path (integer rounded once).
    return qo.ORDER_AVX2 if qo.u8_actual_dim(dim) <= 1040 else qo.ORDER_SIMPLE


@pytest.mark.parametrize("n,dim", SHAPES)
@pytest.mark.parametrize("dist", [D.Dot, D.L1, D.L2])
@pytest.mark.parametrize("invert", [False, True])
def test_encode_query_score_bit_exact(qo, n, dim, dist, invert):
    data, query = _data(n, dim, lo=-1.0 if dist == D.L1 else 0.0)
    vp = qa.VectorParameters(dim, n, dist, invert)
    enc = qa.EncodedVectorsU8.encode(data, vp)
    rows, meta = qo.u8_encode(data, int(dist), invert)
    md = enc.metadata
    assert md["actual_dim"] == meta.actual_dim
    assert_bits_equal([md["alpha"], md["offset"], md["multiplier"]],
                      [meta.alpha, meta.offset, meta.multiplier], "metadata")
    assert np.array_equal(enc.storage_bytes(), rows), "encoded rows differ"

    q = enc.encode_query(query)
    codes, qoff = qo.u8_encode_query(meta, query)
    assert np.array_equal(q.encoded_query, codes)
    assert_bits_equal([q.offset], [qoff], "query offset")

    want = qo.u8_score_all(meta, rows, codes, qoff, order=_order_for(qo, dim))
    assert_bits_equal(enc.score_all(q), want, "score_all")
    # reference pair kernels (compiled from the reference's C) where available
    if qo.ref() is not None and qo.u8_actual_dim(dim) <= 1040:
        assert_bits_equal(enc.score_all(q), qo.u8_score_all(meta, rows, codes, qoff, use_ref=True), "vs _ref")

    for i in (0, n // 2, n - 1):
        assert_bits_equal([enc.score_point(q, i)], [want[i]], f"score_point {i}")
    ids = np.array([n - 1, 0, n // 3, 0], dtype=np.uint32)
    assert_bits_equal(enc.score_ids(q, ids), want[ids], "score_ids")
    for (i, j) in ((0, n - 1), (n // 2, n // 2), (1 % n, 0)):
        w = qo.u8_score_internal(meta, rows, i, j, order=_order_for(qo, dim))
        assert_bits_equal([enc.score_internal(i, j)], [w], f"score_internal {i},{j}")


@pytest.mark.parametrize("dist,name", [(D.Dot, "dot"), (D.L2, "l2"), (D.L1, "l1")])
@pytest.mark.parametrize("invert", [False, True])
def test_reference_tolerance_spec(qo, dist, name, invert):
    """quantization/tests/test_simple.rs:15-49: 129 x 65, |score - f32 metric| < dim*0.1."""
    n, dim = 129, 65
    data, query = _data(n, dim, lo=-1.0 if dist == D.L1 else 0.0)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    scores = enc.score_all(enc.encode_query(query))
    for i in range(n):
        orig = qo.metric_f32(int(dist), query, data[i])
        orig = -orig if invert else orig
        assert abs(scores[i] - orig) < dim * 0.1
    # score_internal spec: the reference tests it for Dot only (test_simple.rs:237-305); for
    # L1/L2 its formula keeps a spurious actual_dim*offset^2 term (encoded_vectors_u8.rs:389-395),
    # reproduced verbatim and pinned bit-for-bit in test_encode_query_score_bit_exact.
    if dist != D.Dot:
        return
    for i in range(0, n - 1, 16):
        orig = qo.metric_f32(int(dist), data[i], data[i + 1])
        orig = -orig if invert else orig
        assert abs(enc.score_internal(i, i + 1) - orig) < dim * 0.1


def test_quantile_deterministic_case(qo):
    """count <= 100 000: the reference's sample is every vector (quantile.rs:31-34)."""
    n, dim = 1000, 48
    data, query = _data(n, dim, seed=3)
    data[5, 7] = 40.0  # outliers that the quantile must cut
    data[9, 1] = -30.0
    for quantile in (0.99, 0.9, 1.0 - np.finfo(np.float32).eps, 1.0):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False), quantile=quantile)
        rows, meta = qo.u8_encode(data, qo.DOT, False, quantile=quantile)
        md = enc.metadata
        assert_bits_equal([md["alpha"], md["offset"]], [meta.alpha, meta.offset], f"quantile {quantile}")
        assert np.array_equal(enc.storage_bytes(), rows)


def test_alpha_offset_override(qo):
    n, dim = 64, 32
    data, query = _data(n, dim, seed=5)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L2, False), alpha_offset=(0.01, -0.2))
    rows, meta = qo.u8_encode_with(data, qo.L2, False, 0.01, -0.2)
    assert np.array_equal(enc.storage_bytes(), rows)
    q = enc.encode_query(query)
    codes, qoff = qo.u8_encode_query(meta, query)
    assert_bits_equal(enc.score_all(q), qo.u8_score_all(meta, rows, codes, qoff), "scores")


def test_edge_values(qo):
    """NaN / inf / denormal / negative-zero inputs; all-zero store (alpha = 0 -> NaN -> code 0,
    quantization/tests/stop_condition.rs uses exactly that data)."""
    dim = 40
    special = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-42, -1e-42, 3.4e38, -3.4e38, 1.0, -1.0, 0.5],
                       dtype=np.float32)
    rng = np.random.default_rng(1)
    data = rng.standard_normal((50, dim)).astype(np.float32)
    data[:, :12] = special
    for dist in (D.Dot, D.L2, D.L1):
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, 50, dist, False))
        rows, meta = qo.u8_encode(data, int(dist), False)
        assert np.array_equal(enc.storage_bytes(), rows)
    finite = np.nan_to_num(data, nan=0.0, posinf=2.0, neginf=-2.0)
    enc = qa.EncodedVectorsU8.encode(finite, qa.VectorParameters(dim, 50, D.Dot, False))
    rows, meta = qo.u8_encode(finite, qo.DOT, False)
    assert np.array_equal(enc.storage_bytes(), rows)
    q = enc.encode_query(data[0])  # query with NaN/inf entries
    codes, qoff = qo.u8_encode_query(meta, data[0])
    assert np.array_equal(q.encoded_query, codes)
    assert_bits_equal(enc.score_all(q), qo.u8_score_all(meta, rows, codes, qoff), "scores")

    zeros = np.zeros((100, 8), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(zeros, qa.VectorParameters(8, 100, D.Dot, False))
    rows, meta = qo.u8_encode(zeros, qo.DOT, False)
    assert np.array_equal(enc.storage_bytes(), rows)
    q = enc.encode_query(zeros[0])
    codes, qoff = qo.u8_encode_query(meta, zeros[0])
    assert_bits_equal(enc.score_all(q), qo.u8_score_all(meta, rows, codes, qoff), "all-zero store")


def test_empty_storage_roundtrip(tmp_path):
    """quantization/tests/empty_storage.rs: count = 0 encode -> save -> load."""
    vp = qa.VectorParameters(256, 0, D.Dot, False)
    enc = qa.EncodedVectorsU8.encode(np.zeros((0, 256), np.float32), vp)
    md = enc.metadata
    assert (md["alpha"], md["offset"], md["multiplier"]) == (0.0, 0.0, 0.0)
    enc.save(tmp_path / "d" / "data.bin", tmp_path / "m" / "meta.json")
    assert os.path.getsize(tmp_path / "d" / "data.bin") == 0
    back = qa.EncodedVectorsU8.load(tmp_path / "d" / "data.bin", tmp_path / "m" / "meta.json", vp)
    assert back.score_all(back.encode_query(np.zeros(256, np.float32))).size == 0


def test_save_load_roundtrip_and_format(qo, tmp_path):
    import json
    n, dim = 200, 65
    data, query = _data(n, dim, seed=9)
    vp = qa.VectorParameters(dim, n, D.L2, True)
    enc = qa.EncodedVectorsU8.encode(data, vp)
    dp, mp = tmp_path / "data.bin", tmp_path / "meta.json"
    enc.save(dp, mp)
    rows, meta = qo.u8_encode(data, qo.L2, True)
    assert open(dp, "rb").read() == rows.tobytes()  # raw row file (encoded_storage.rs:54-59)
    js = json.load(open(mp))
    assert list(js.keys()) == ["actual_dim", "alpha", "offset", "multiplier", "vector_parameters"]
    assert js["vector_parameters"] == {"dim": dim, "count": n, "distance_type": "L2", "invert": True}
    assert np.float32(js["alpha"]) == np.float32(meta.alpha)
    back = qa.EncodedVectorsU8.load(dp, mp, vp)
    assert_bits_equal(back.score_all(back.encode_query(query)), enc.score_all(enc.encode_query(query)), "reload")
    with pytest.raises(OSError):  # encoded_storage.rs:40-51 size check
        qa.EncodedVectorsU8.load(dp, mp, qa.VectorParameters(dim, n + 1, D.L2, True))
    with pytest.raises(OSError):
        qa.EncodedVectorsU8.load(tmp_path / "nope.bin", mp, vp)


def test_from_storage_reference_rows(qo):
    """A store encoded elsewhere (reference row format) scores identically after upload."""
    n, dim = 500, 768
    data, query = _data(n, dim, seed=11)
    rows, meta = qo.u8_encode(data, qo.DOT, False)
    md = {"actual_dim": meta.actual_dim, "alpha": meta.alpha, "offset": meta.offset,
          "multiplier": meta.multiplier,
          "vector_parameters": qa.VectorParameters(dim, n, D.Dot, False)}
    enc = qa.EncodedVectorsU8.from_storage(rows, md)
    assert np.array_equal(enc.storage_bytes(), rows)
    codes, qoff = qo.u8_encode_query(meta, query)
    assert_bits_equal(enc.score_all(enc.encode_query(query)), qo.u8_score_all(meta, rows, codes, qoff), "scores")


def test_stop_condition():
    """quantization/tests/stop_condition.rs: Err(EncodingError::Stopped)."""
    data = np.zeros((10000, 8), dtype=np.float32)
    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(8, 10000, D.Dot, False), stop_condition=lambda: True)
    assert e.value.stopped
    calls = {"n": 0}

    def later():
        calls["n"] += 1
        return calls["n"] > 1  # lets pass 1 start, stops before pass 2

    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(8, 10000, D.Dot, False), stop_condition=later)
    assert e.value.stopped
    qa.EncodedVectorsU8.encode(data, qa.VectorParameters(8, 10000, D.Dot, False), stop_condition=lambda: False)


def test_argument_errors():
    data = np.zeros((4, 8), dtype=np.float32)
    with pytest.raises(qa.EncodingError) as e:  # validate_vector_parameters, encoded_vectors.rs:47-70
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(8, 5, D.Dot, False))
    assert e.value.kind == "ArgumentsError"
    with pytest.raises(qa.EncodingError):
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(9, 4, D.Dot, False))
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(8, 4, D.Dot, False))
    q = enc.encode_query(np.zeros(8, np.float32))
    with pytest.raises(IndexError):  # the reference panics on the slice index
        enc.score_point(q, 4)
    with pytest.raises(IndexError):
        enc.score_internal(0, 9)
    with pytest.raises(qa.EncodingError):
        enc.score_all(enc.encode_query(np.zeros(40, np.float32)))  # wrong query length


@pytest.mark.parametrize("dim", [2048, 2064, 4096, 5000])
def test_large_dims_generic_kernel_and_lane_modes(qo, dim):
    """actual_dim > 1040: sums can exceed 2^24.  Mode 0 = exact integer rounded once (the
    reference's scalar path); mode 1 = avx2.c's 8-lane f32 summation, bit for bit — checked on
    adversarial all-127 rows against the restated AVX2 order and the compiled reference."""
    n = 37
    rng = np.random.default_rng(dim)
    ad = qo.u8_actual_dim(dim)
    codes = rng.integers(0, 128, size=(n, ad), dtype=np.uint8)
    codes[0] = 127
    codes[1, ::2] = 127
    rows = np.zeros((n, ad + 4), dtype=np.uint8)
    rows[:, 4:] = codes
    rows[:, :4] = rng.standard_normal(n).astype(np.float32).view(np.uint8).reshape(n, 4)
    md = {"actual_dim": ad, "alpha": 1.0, "offset": 0.0, "multiplier": 1.0,
          "vector_parameters": qa.VectorParameters(dim, n, D.Dot, False)}
    enc = qa.EncodedVectorsU8.from_storage(rows, md)
    meta = qo.Meta(ad, 1.0, 0.0, 1.0, dim, n, qo.DOT, 0)
    qcodes = np.full(ad, 127, dtype=np.uint8)
    query = qcodes[:dim].astype(np.float32)  # alpha 1, offset 0: codes == values
    q = enc.encode_query(query)
    _, qoff = qo.u8_encode_query(meta, query)
    got_codes = q.encoded_query
    want0 = qo.u8_score_all(meta, rows, got_codes, qoff, order=qo.ORDER_SIMPLE)
    assert_bits_equal(enc.score_all(q), want0, "mode 0")
    enc.set_lane_mode(1)
    want1 = qo.u8_score_all(meta, rows, got_codes, qoff, order=qo.ORDER_AVX2)
    assert_bits_equal(enc.score_all(q), want1, "mode 1 (avx2 lanes)")
    if qo.ref() is not None:
        assert_bits_equal(enc.score_all(q), qo.u8_score_all(meta, rows, got_codes, qoff, use_ref=True), "vs _ref")


def test_device_resident_inputs_and_outputs(qo):
    """Vectors, query and scores all in HBM (torch = plumbing): same bits as the host path."""
    torch = pytest.importorskip("torch")
    n, dim = 3000, 768
    data, query = _data(n, dim, seed=21)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    d_data = torch.from_numpy(data).cuda()
    d_query = torch.from_numpy(query).cuda()
    enc = qa.EncodedVectorsU8.encode(d_data, vp)
    rows, meta = qo.u8_encode(data, qo.DOT, False)
    assert np.array_equal(enc.storage_bytes(), rows)
    q_dev = enc.encode_query(d_query)
    q_host = enc.encode_query(query)
    assert np.array_equal(q_dev.encoded_query, q_host.encoded_query)
    assert_bits_equal([q_dev.offset], [q_host.offset], "device-encoded query offset")
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    enc.score_all(q_dev, out=out)
    torch.cuda.synchronize()
    codes, qoff = qo.u8_encode_query(meta, query)
    assert_bits_equal(out.cpu().numpy(), qo.u8_score_all(meta, rows, codes, qoff), "device scores")
    # reuse of the query object
    q2 = enc.encode_query(d_data[5], reuse=q_dev)
    assert q2 is q_dev
    codes, qoff = qo.u8_encode_query(meta, data[5])
    torch.cuda.synchronize()
    assert_bits_equal(enc.score_all(q_dev), qo.u8_score_all(meta, rows, codes, qoff), "reused query")


def test_quantile_sampled_case_is_conditionally_exact(qo):
    """count > 100 000: the reference's interval comes from a RANDOM 100k-vector sample
    (quantile.rs:31-34) — parity unpinned for (alpha, offset); everything after them is exact."""
    torch = pytest.importorskip("torch")
    n, dim = 150_000, 8
    rng = np.random.default_rng(77)
    data = rng.standard_normal((n, dim)).astype(np.float32)
    for src in (data, torch.from_numpy(data).cuda()):
        enc = qa.EncodedVectorsU8.encode(src, qa.VectorParameters(dim, n, D.Dot, False), quantile=0.98)
        md = enc.metadata
        lo, hi = float(md["offset"]), float(md["offset"]) + 127.0 * float(md["alpha"])
        # the reference cuts slice_size*(1-q)/2 VALUES (not vectors' worth) per side
        # (quantile.rs:52-55): a (1-q)/(2*dim) = 0.125 % tail here, near -/+3.02 sigma
        frac = (1.0 - 0.98) / (2 * dim)
        assert abs(lo - np.quantile(data, frac)) < 0.05 and abs(hi - np.quantile(data, 1 - frac)) < 0.05, (lo, hi)
        rows, meta = qo.u8_encode_with(data, qo.DOT, False, float(md["alpha"]), float(md["offset"]))
        assert np.array_equal(enc.storage_bytes(), rows)


def test_cosine_is_normalised_dot(qo):
    """The reference has no Cosine distance: callers L2-normalise vectors and queries
    (cosine_preprocess, demos/src/ann_benchmark_data.rs:223-230) and use Dot
    (demos/src/ann_benchmark.rs:114-116).  Bit-exact against the oracle; against the true cosine
    within the reference's u8 tolerance scaled to unit vectors."""
    rng = np.random.default_rng(13)
    n, dim = 2000, 768
    data = rng.standard_normal((n, dim)).astype(np.float32)
    query = rng.standard_normal(dim).astype(np.float32)

    def cosine_preprocess(v):  # ann_benchmark_data.rs:223-230, f32 arithmetic
        length = np.float32(0.0)
        for x in v:
            length = np.float32(length + x * x)
        if length < np.finfo(np.float32).eps:
            return v
        return (v / np.sqrt(length, dtype=np.float32)).astype(np.float32)

    nd = np.stack([cosine_preprocess(v) for v in data[:64]] + [(v / np.linalg.norm(v)).astype(np.float32) for v in data[64:]])
    nq = cosine_preprocess(query)
    enc = qa.EncodedVectorsU8.encode(nd, qa.VectorParameters(dim, n, D.Dot, False))
    rows, meta = qo.u8_encode(nd, qo.DOT, False)
    assert np.array_equal(enc.storage_bytes(), rows)
    codes, qoff = qo.u8_encode_query(meta, nq)
    got = enc.score_all(enc.encode_query(nq))
    assert_bits_equal(got, qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2), "cosine (normalised dot)")
    true_cos = (data @ query) / (np.linalg.norm(data, axis=1) * np.linalg.norm(query))
    # stated f32 tolerance: per-dimension quantisation step alpha ~ (max-min)/127 of unit vectors
    assert np.max(np.abs(got - true_cos)) < 0.01
    top = np.argsort(-got)[:10]
    assert len(set(top) & set(np.argsort(-true_cos)[:20])) >= 8  # recall sanity, as ann_benchmark's same_10


def test_division_free_quantize_is_exact_at_every_code_boundary(qo):
    """quantize16_kernel converts with (v - offset) * (1 / alpha) and redoes the reference's division
    ((v - offset) / alpha, encoded_vectors_u8.rs:234-237) only inside a 2^-15 band around the integers.
    Adversarial sweep: for several (alpha, offset), every boundary k = -2 .. 130, the f32 values within +-6 ulp of
    offset + alpha * k (where a reciprocal-multiply alone WOULD flip codes), plus specials; alphas outside
    2^-60 .. 2^60 take the exact kernel.  Codes and vector_offset must equal the oracle's byte for byte."""
    rng = np.random.default_rng(5)
    pairs = [(1.0 / 127.0, 0.0), (float(np.float32(0.999999) / np.float32(127.0)), 1e-7), (0.0078125, -0.5),
             (3.1415927e-3, 2.7182817), (1.7e-5, -1234.5), (9.313226e-10, 0.25), (123456.789, -1e7),
             (1e-25, 0.0), (1e25, 3.0), (float(np.float32(2.0) ** -70), 0.0)]
    for alpha, offset in pairs:
        a32, o32 = np.float32(alpha), np.float32(offset)
        vals = []
        for k in range(-2, 131):
            centre = np.float32(np.float64(o32) + np.float64(a32) * k)
            bits = centre.view(np.int32)
            for d in range(-6, 7):
                vals.append(np.int32(bits + d).view(np.float32))
        vals += [np.float32(x) for x in (np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-45, -1e-45, 3.4e38, -3.4e38)]
        vals = np.array(vals, dtype=np.float32)
        vals = vals[np.isfinite(vals) | np.isnan(vals) | np.isinf(vals)]
        dim = 64
        n = (vals.size + dim - 1) // dim + 64
        data = rng.uniform(float(o32) - float(a32), float(o32) + 128 * float(a32), (n, dim)).astype(np.float32)
        data.reshape(-1)[:vals.size] = vals
        for dist in (D.Dot, D.L2):
            enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, False), alpha_offset=(float(a32), float(o32)))
            rows, _meta = qo.u8_encode_with(data, int(dist), False, float(a32), float(o32))
            got = enc.storage_bytes()
            assert np.array_equal(got[:, 4:], rows[:, 4:]), f"codes differ for alpha={alpha} offset={offset}"
            assert np.array_equal(got, rows), f"vector_offset differs for alpha={alpha} offset={offset}"
    # random data at a realistic scale, 2M values: the fast and the exact conversion must agree everywhere
    data = rng.random((4096, 512), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(512, 4096, D.Dot, False))
    rows, _ = qo.u8_encode(data, qo.DOT, False)
    assert np.array_equal(enc.storage_bytes(), rows)


def test_pass1_statistics_on_their_own(qo):
    """qamd_u8_find_min_max / qamd_u8_find_quantile_interval / qamd_pq_find_centroids - the global statistics of `encode` as
    separate entry points for hosts that hold the rows in several places: equal to the oracle's find_min_max_from_iter
    (quantile.rs:5-19; NaN never wins, no rows -> (f32::MAX, f32::MIN)), find_quantile_interval (:21-71) and to what the
    one-shot encoders use."""
    rng = np.random.default_rng(12)
    data = (rng.standard_normal((5000, 24)) * 3).astype(np.float32)
    data[7, 3] = np.nan
    data[100, 0] = np.float32(-0.0)
    for rows in (data, data[:1], data[:0], data[:4097, :17].copy()):
        mn, mx = qa.EncodedVectorsU8.find_min_max(rows)
        wmn, wmx = qo.find_min_max(rows)
        assert (np.float32(mn).view(np.uint32), np.float32(mx).view(np.uint32)) == (wmn.view(np.uint32), wmx.view(np.uint32))
    assert qa.EncodedVectorsU8.find_min_max(data[:0]) == (np.finfo(np.float32).max, -np.finfo(np.float32).max)
    clean = np.nan_to_num(data)
    for q in (0.99, 0.9, 0.5):
        got = qa.EncodedVectorsU8.find_quantile_interval(clean, q)
        want = qo.find_quantile_interval(clean, q)
        assert (got is None) == (want is None)
        if got is not None:
            assert np.float32(got[0]).view(np.uint32) == want[0].view(np.uint32) and np.float32(got[1]).view(np.uint32) == want[1].view(np.uint32)
    assert qa.EncodedVectorsU8.find_quantile_interval(clean[:100], 0.99) is None  # fewer than 127 vectors (quantile.rs:27-29)
    assert qa.EncodedVectorsU8.find_quantile_interval(clean, 1.0) is None
    # the interval from the two statistics = the one-shot encode's
    enc = qa.EncodedVectorsU8.encode(clean, qa.VectorParameters(24, 5000, D.Dot, False), 0.95)
    mn, mx = qa.EncodedVectorsU8.find_quantile_interval(clean, 0.95)
    a, o = qo.alpha_offset(mn, mx)
    assert np.float32(enc.metadata["alpha"]).view(np.uint32) == a.view(np.uint32)
    assert np.float32(enc.metadata["offset"]).view(np.uint32) == o.view(np.uint32)
    # PQ: find_centroids on its own = the centroids `encode` trains
    pdata = rng.random((3000, 16), dtype=np.float32)
    cen = qa.EncodedVectorsPQ.find_centroids(pdata, 4, 2)
    penc = qa.EncodedVectorsPQ.encode(pdata, qa.VectorParameters(16, 3000, D.L2, False), 4, max_kmeans_threads=2)
    assert np.array_equal(cen.view(np.uint32), penc.centroids.view(np.uint32))
    small = qa.EncodedVectorsPQ.find_centroids(pdata[:100], 4)  # count <= 256: the vectors themselves (:290-297)
    assert np.array_equal(small, qo.pq_centroids_small(pdata[:100]))
