"""The reference's own test assertions, re-run on the oracle (CPU).

The Rust half of the reference cannot be compiled here; these are the properties its tests
pin (quantization/tests/*.rs), plus the exactness facts of SURVEY 8a.  Together with
test_oracle_golden.py (reference C kernels) this is what pins oracle/qoracle.c.
"""
import numpy as np
import pytest

D = {"dot": 0, "l1": 1, "l2": 2}


def _gen(n, dim, seed=42, lo=0.0):
    rng = np.random.default_rng(seed)
    scale = np.float32(1.0 - lo)
    return rng.random((n, dim), dtype=np.float32) * scale + np.float32(lo), \
        rng.random(dim, dtype=np.float32) * scale + np.float32(lo)


@pytest.mark.parametrize("dist", ["dot", "l2", "l1"])
@pytest.mark.parametrize("invert", [False, True])
@pytest.mark.parametrize("order", [0, 1, 2])
def test_u8_tolerance_spec(qo, dist, invert, order):
    """test_simple.rs / test_avx2.rs / test_sse.rs: 129 x 65 (padding 65 -> 80), error < dim*0.1."""
    if dist == "l1" and order == 2:
        pytest.skip("impl_score_l1_sse wraps mod 2^16 (SURVEY 2.1): not an oracle")
    n, dim = 129, 65
    data, query = _gen(n, dim, lo=-1.0 if dist == "l1" else 0.0)
    rows, meta = qo.u8_encode(data, D[dist], invert)
    assert meta.actual_dim == 80 and rows.shape == (n, 84)
    codes, qoff = qo.u8_encode_query(meta, query)
    scores = qo.u8_score_all(meta, rows, codes, qoff, order=order)
    for i in range(n):
        orig = qo.metric_f32(D[dist], query, data[i])
        orig = -orig if invert else orig
        assert abs(scores[i] - orig) < dim * 0.1
    assert rows[:, 4:].max() <= 127  # codes are 7-bit (encoded_vectors_u8.rs:236)
    if dist == "dot":  # test_dot_internal_simple
        for i in range(1, n):
            s = qo.u8_score_internal(meta, rows, 0, i, order=order)
            orig = qo.metric_f32(0, data[0], data[i])
            assert abs(s - (-orig if invert else orig)) < dim * 0.1


def test_u8_orders_agree_up_to_dim_1040(qo):
    """Every reference kernel returns the exact integer for actual_dim <= 1040."""
    for dim in (16, 65, 768, 1040):
        data, query = _gen(64, dim, seed=dim)
        rows, meta = qo.u8_encode(data, 0, False)
        codes, qoff = qo.u8_encode_query(meta, query)
        a = qo.u8_score_all(meta, rows, codes, qoff, order=0)
        for order in (1, 2):
            assert np.array_equal(a.view(np.uint32), qo.u8_score_all(meta, rows, codes, qoff, order=order).view(np.uint32))
        if qo.ref() is not None:
            assert np.array_equal(a.view(np.uint32), qo.u8_score_all(meta, rows, codes, qoff, use_ref=True).view(np.uint32))


def test_u8_large_quantile(qo):
    """test_simple.rs test_u8_large_quantile: quantile = 1 - eps still encodes sanely."""
    n, dim = 129, 65
    data, query = _gen(n, dim)
    q = float(np.float32(1.0) - np.finfo(np.float32).eps)
    rows, meta = qo.u8_encode(data, 0, False, quantile=q)
    codes, qoff = qo.u8_encode_query(meta, query)
    scores = qo.u8_score_all(meta, rows, codes, qoff)
    for i in range(n):
        assert abs(scores[i] - qo.metric_f32(0, query, data[i])) < dim * 0.1


def test_u8_empty_and_all_zero(qo):
    rows, meta = qo.u8_encode_empty(256, 0, False)
    assert (meta.alpha, meta.offset, meta.multiplier, meta.actual_dim) == (0.0, 0.0, 0.0, 256)
    zeros = np.zeros((10, 8), dtype=np.float32)  # stop_condition.rs data: alpha 0 -> NaN -> code 0
    rows, meta = qo.u8_encode(zeros, 0, False)
    assert meta.alpha == 0.0 and not rows[:, 4:].any()


@pytest.mark.parametrize("store", [0, 1])
@pytest.mark.parametrize("invert", [False, True])
def test_binary_known_answer(qo, store, invert):
    """test_binary.rs:14-71: on +-1 vectors the Dot score is exactly the f32 dot product."""
    rng = np.random.default_rng(42)
    for dim in ([0, 1, 8, 33, 65, 387] if store == 0 else [1, 387]):
        data = np.where(rng.random((128, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
        query = np.where(rng.random(dim) < 0.5, -1.0, 1.0).astype(np.float32)
        rows = qo.bin_encode(data, store)
        qb = qo.bin_encode(query[None, :], store)[0]
        got = qo.bin_score_all(rows, qb, dim, 0, invert, store)
        want = (data @ query).astype(np.float32)
        assert np.array_equal(got, -want if invert else want), dim
        if qo.ref() is not None:
            assert np.array_equal(got, qo.bin_score_all(rows, qb, dim, 0, invert, store, use_ref=True))


def test_binary_row_sizes(qo):
    """BitsStoreType::get_storage_size (encoded_vectors_binary.rs:99-116, :152-159)."""
    want_u8 = {0: 0, 1: 1, 8: 1, 9: 2, 32: 4, 33: 8, 64: 8, 65: 16, 128: 16, 129: 32, 387: 64, 1024: 128}
    for dim, nb in want_u8.items():
        assert qo.bin_row_bytes(dim, 0) == nb, dim
    want_u128 = {0: 0, 1: 16, 128: 16, 129: 32, 387: 64, 1024: 128}
    for dim, nb in want_u128.items():
        assert qo.bin_row_bytes(dim, 1) == nb, dim


@pytest.mark.parametrize("dist", ["l1", "l2"])
def test_binary_ordering(qo, dist):
    """test_binary.rs:243-263."""
    rng = np.random.default_rng(1)
    for dim in (8, 33, 65, 387):
        data = np.where(rng.random((128, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
        query = np.where(rng.random(dim) < 0.5, -1.0, 1.0).astype(np.float32)
        rows = qo.bin_encode(data)
        got = qo.bin_score_all(rows, qo.bin_encode(query[None])[0], dim, D[dist], False)
        true = np.array([qo.metric_f32(D[dist], query, data[i]) for i in range(128)])
        assert np.array_equal(got[np.argsort(true, kind="stable")], np.sort(got))


@pytest.mark.parametrize("dist", ["dot", "l1", "l2"])
@pytest.mark.parametrize("invert", [False, True])
def test_pq_tolerance_spec_chunk1(qo, dist, invert):
    """test_pq.rs:16-50: 513 x 65, chunk_size 1, error < dim*0.05 — with a good 1-D codebook
    (256 per-dimension quantiles standing in for the reference's randomised k-means)."""
    n, dim = 513, 65
    data, query = _gen(n, dim)
    cen = np.quantile(data, (np.arange(256) + 0.5) / 256, axis=0).astype(np.float32)
    rows = qo.pq_encode(data, 1, cen)
    lut = qo.pq_encode_query(query, 1, cen, D[dist], invert)
    for order in (0, 2):
        scores = qo.pq_score_all(rows, lut, order=order)
        for i in range(n):
            orig = qo.metric_f32(D[dist], query, data[i])
            assert abs(scores[i] - (-orig if invert else orig)) < dim * 0.05
    for i in range(0, n - 1, 32):
        orig = qo.metric_f32(D[dist], data[i], data[i + 1])
        s = qo.pq_score_internal(rows, dim, 1, cen, D[dist], invert, i, i + 1)
        assert abs(s - (-orig if invert else orig)) < dim * 0.05


def test_pq_division_and_small_count(qo):
    assert qo.pq_chunks(768, 8) == 96 and qo.pq_chunks(65, 8) == 9 and qo.pq_chunks(65, 1) == 65
    data, _ = _gen(100, 24, seed=2)
    cen = qo.pq_centroids_small(data)
    assert np.array_equal(cen[:100], data) and not cen[100:].any()
    rows = qo.pq_encode(data, 5, cen)
    assert rows.shape == (100, 5)
    assert np.array_equal(rows[:, 0], np.arange(100))  # each vector is its own nearest centroid


def test_f32_to_u8_truncates_and_saturates(qo):
    L = qo.lib()
    assert L.qo_f32_to_u8(0.999, 1.0, 0.0) == 0       # truncation toward zero, not rounding
    assert L.qo_f32_to_u8(126.999, 1.0, 0.0) == 126
    assert L.qo_f32_to_u8(500.0, 1.0, 0.0) == 127
    assert L.qo_f32_to_u8(-3.0, 1.0, 0.0) == 0
    assert L.qo_f32_to_u8(float("nan"), 1.0, 0.0) == 0
    assert L.qo_f32_to_u8(1.0, 0.0, 0.0) == 127      # x/0 = +inf -> clamp
    assert L.qo_f32_to_u8(0.0, 0.0, 0.0) == 0        # 0/0 = NaN -> 0
