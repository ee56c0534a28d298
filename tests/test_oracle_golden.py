"""Pin the oracle's pair kernels to the reference's own C kernels.

pair_kernels.npz holds outputs of quantization/cpp/avx2.c + cpp/sse.c compiled with
build.rs's flags (tests/golden/make_golden.py).  Bit-exact comparisons.
"""
import ctypes as C

import numpy as np
import pytest


def _each(golden):
    for name in golden["case_names"]:
        name = str(name)
        yield name, golden[f"{name}__q"], golden[f"{name}__v"]


def test_dot_avx2_order_matches_reference_bits(qo, golden):
    L = qo.lib()
    for name, q, v in _each(golden):
        want = golden[f"{name}__dot_avx"]
        for i in range(q.shape[0]):
            got = np.float32(L.qo_dot_avx2_order(q[i].ctypes.data, v[i].ctypes.data, q.shape[1]))
            assert got.view(np.uint32) == want[i].view(np.uint32), (name, i, got, want[i])


def test_dot_sse_order_matches_reference_bits(qo, golden):
    L = qo.lib()
    for name, q, v in _each(golden):
        want = golden[f"{name}__dot_sse"]
        for i in range(q.shape[0]):
            got = np.float32(L.qo_dot_sse_order(q[i].ctypes.data, v[i].ctypes.data, q.shape[1]))
            assert got.view(np.uint32) == want[i].view(np.uint32), (name, i)


def test_l1_avx2_order_matches_reference_bits(qo, golden):
    L = qo.lib()
    for name, q, v in _each(golden):
        want = golden[f"{name}__l1_avx"]
        for i in range(q.shape[0]):
            got = np.float32(L.qo_l1_avx2_order(q[i].ctypes.data, v[i].ctypes.data, q.shape[1]))
            assert got.view(np.uint32) == want[i].view(np.uint32), (name, i)


def test_xor_popcnt_matches_reference(qo, golden):
    L = qo.lib()
    for name, q, v in _each(golden):
        want = golden[f"{name}__popcnt128"]
        for i in range(q.shape[0]):
            assert L.qo_xor_popcnt(q[i].ctypes.data, v[i].ctypes.data, q.shape[1]) == want[i]
    q, v = golden["small__q"], golden["small__v"]
    for i in range(q.shape[0]):
        assert L.qo_xor_popcnt(q[i].ctypes.data, v[i].ctypes.data, 16) == golden["small__popcnt64x2"][i]
        assert L.qo_xor_popcnt(q[i].ctypes.data, v[i].ctypes.data, 8) == golden["small__popcnt32x2"][i]


def test_exact_integer_for_encoder_codes_up_to_dim_1040(qo, golden):
    """codes <= 127 and actual_dim <= 1040 => every kernel returns the exact
    integer, so summation order is irrelevant there (simple == AVX2 == SSE)."""
    L = qo.lib()
    for name, q, v in _each(golden):
        if not (name.startswith("codes127") or name.startswith("all127") or name.startswith("sparse")):
            continue
        dim = q.shape[1]
        if dim > 1040:
            continue
        for i in range(q.shape[0]):
            exact = int(np.dot(q[i].astype(np.int64), v[i].astype(np.int64)))
            assert exact < 2 ** 24
            assert golden[f"{name}__dot_avx"][i] == np.float32(exact)
            assert golden[f"{name}__dot_sse"][i] == np.float32(exact)
            assert L.qo_dot_i32(q[i].ctypes.data, v[i].ctypes.data, dim) == exact


def test_live_reference_differential(qo):
    """When oracle/_ref is present (built here, shipped prebuilt to the GPU box), fuzz the
    restatement against it on fresh inputs."""
    R = qo.ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    L = qo.lib()
    rng = np.random.default_rng(7)
    for dim in (16, 48, 80, 768, 1536, 3072):
        for hi in (128, 256):
            for _ in range(25):
                q = rng.integers(0, hi, size=dim, dtype=np.uint8)
                v = rng.integers(0, hi, size=dim, dtype=np.uint8)
                qp, vp = q.ctypes.data, v.ctypes.data
                a = np.float32(R.impl_score_dot_avx(qp, vp, dim))
                b = np.float32(L.qo_dot_avx2_order(qp, vp, dim))
                assert a.view(np.uint32) == b.view(np.uint32)
                a = np.float32(R.impl_score_dot_sse(qp, vp, dim))
                b = np.float32(L.qo_dot_sse_order(qp, vp, dim))
                assert a.view(np.uint32) == b.view(np.uint32)
                a = np.float32(R.impl_score_l1_avx(qp, vp, dim))
                b = np.float32(L.qo_l1_avx2_order(qp, vp, dim))
                assert a.view(np.uint32) == b.view(np.uint32)
                assert R.impl_xor_popcnt_sse_uint128(qp, vp, dim // 16) == L.qo_xor_popcnt(qp, vp, dim)
