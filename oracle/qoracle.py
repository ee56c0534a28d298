"""ctypes/numpy front-end of the parity oracle (oracle/qoracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the CPU-baseline legs of the
two measurement harnesses (bench.py's cpu_baseline; the "oracle CPU loop" column of tools/ann_protocol.py,
the reference's own ann_benchmark protocol) — as the checker / the timed CPU side, never by quantization_amd/.  See the header of
qoracle.c for what pins it (compiled reference C kernels in oracle/_ref +
committed golden vectors + the reference tests' own assertions).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libqoracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libsimd_utils_ref.so")

DOT, L1, L2 = 0, 1, 2
ORDER_SIMPLE, ORDER_AVX2, ORDER_SSE = 0, 1, 2
STORE_U8, STORE_U128 = 0, 1


class Meta(C.Structure):
    """encoded_vectors_u8.rs:24-31 Metadata + VectorParameters."""

    _fields_ = [
        ("actual_dim", C.c_uint64),
        ("alpha", C.c_float),
        ("offset", C.c_float),
        ("multiplier", C.c_float),
        ("dim", C.c_uint64),
        ("count", C.c_uint64),
        ("distance_type", C.c_int32),
        ("invert", C.c_int32),
    ]


def build(force: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when the reference sources are present)."""
    if force or not os.path.exists(LIB_PATH) or (
        os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "qoracle.c"))
    ):
        subprocess.check_call(["make", "-C", HERE, "all"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/quantization/cpp") and (force or not os.path.exists(REF_PATH)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None
_ref = None

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
PAIR_F32 = C.CFUNCTYPE(C.c_float, _u8p, _u8p, C.c_uint32)
PAIR_U32 = C.CFUNCTYPE(C.c_uint32, _u8p, _u8p, C.c_uint32)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        u64, i32, f32, u32 = C.c_uint64, C.c_int, C.c_float, C.c_uint32
        sig = {
            "qo_find_min_max": (None, [vp, u64, vp, vp]),
            "qo_u8_alpha_offset": (None, [f32, f32, vp, vp]),
            "qo_f32_to_u8": (C.c_uint8, [f32, f32, f32]),
            "qo_u8_actual_dim": (u64, [u64]),
            "qo_u8_quantized_vector_size": (u64, [u64]),
            "qo_find_quantile_interval": (i32, [vp, u64, u64, f32, vp, vp]),
            "qo_u8_encode": (i32, [vp, u64, u64, i32, i32, f32, vp, vp]),
            "qo_u8_encode_with": (None, [vp, u64, u64, i32, i32, f32, f32, vp, vp]),
            "qo_u8_encode_query": (f32, [vp, vp, u64, vp]),
            "qo_dot_i32": (C.c_int32, [vp, vp, u32]),
            "qo_l1_i32": (C.c_int32, [vp, vp, u32]),
            "qo_dot_simple": (f32, [vp, vp, u32]),
            "qo_l1_simple": (f32, [vp, vp, u32]),
            "qo_dot_avx2_order": (f32, [vp, vp, u32]),
            "qo_dot_sse_order": (f32, [vp, vp, u32]),
            "qo_l1_avx2_order": (f32, [vp, vp, u32]),
            "qo_xor_popcnt": (u32, [vp, vp, u32]),
            "qo_u8_score_point": (f32, [vp, vp, vp, f32, u64, i32]),
            "qo_u8_score_all": (None, [vp, vp, vp, f32, u64, u64, i32, vp, vp, vp]),
            "qo_u8_score_internal": (f32, [vp, vp, u64, u64, i32]),
            "qo_bin_row_bytes": (u64, [u64, i32]),
            "qo_bin_encode": (None, [vp, u64, u64, i32, vp]),
            "qo_bin_metric": (f32, [u32, u64, i32, i32]),
            "qo_bin_score_point": (f32, [vp, vp, u64, i32, i32, i32, u64]),
            "qo_bin_score_all": (None, [vp, vp, u64, i32, i32, i32, u64, u64, vp, vp]),
            "qo_bin_score_internal": (f32, [vp, u64, i32, i32, i32, u64, u64]),
            "qo_pq_chunks": (u64, [u64, u64]),
            "qo_pq_centroids_small": (None, [vp, u64, u64, vp]),
            "qo_pq_encode": (None, [vp, u64, u64, u64, vp, vp]),
            "qo_pq_encode_query": (None, [vp, u64, u64, vp, i32, i32, vp]),
            "qo_pq_score_simple": (f32, [vp, u64, vp]),
            "qo_pq_score_sse_order": (f32, [vp, u64, vp]),
            "qo_pq_score_all": (None, [vp, u64, vp, u64, u64, i32, vp]),
            "qo_pq_score_internal": (f32, [vp, u64, u64, vp, i32, i32, u64, u64]),
            "qo_metric_f32": (f32, [i32, vp, vp, u64]),
            "qo_kmeans": (i32, [vp, u64, u64, u64, u32, u32, f32, u32, vp, vp, vp, vp]),
            "qo_find_centroids": (i32, [vp, u64, u64, u64, vp, u64, u32, vp, vp, vp]),
            "qo_topk_heap": (u64, [vp, u64, u64, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ref() -> C.CDLL | None:
    """The reference's own C kernels (cpp/avx2.c, cpp/sse.c) compiled into oracle/_ref."""
    global _ref
    if _ref is None:
        build()
        if not os.path.exists(REF_PATH):
            return None
        R = C.CDLL(REF_PATH)
        for n in ("impl_score_dot_avx", "impl_score_l1_avx", "impl_score_dot_sse", "impl_score_l1_sse"):
            f = getattr(R, n)
            f.restype = C.c_float
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        for n in ("impl_xor_popcnt_sse_uint128", "impl_xor_popcnt_sse_uint64", "impl_xor_popcnt_sse_uint32"):
            f = getattr(R, n)
            f.restype = C.c_uint32
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        _ref = R
    return _ref


def _p(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


# ----------------------------------------------------------------------------- u8
def u8_actual_dim(dim: int) -> int:
    return int(lib().qo_u8_actual_dim(dim))


def u8_encode(data, distance: int, invert: bool, quantile: float | None = None):
    """EncodedVectorsU8::encode -> (rows [count, actual_dim+4] u8, Meta)."""
    data = _f32(data)
    count, dim = data.shape if data.ndim == 2 else (0, 0)
    stride = u8_actual_dim(dim) + 4
    rows = np.zeros((count, stride), dtype=np.uint8)
    meta = Meta()
    r = lib().qo_u8_encode(_p(data), count, dim, distance, int(invert),
                           -1.0 if quantile is None else float(quantile), _p(rows), C.byref(meta))
    if r != 0:
        raise ValueError("quantile with count > 100000 is a random sample in the reference; "
                         "use u8_encode_with(alpha, offset)")
    return rows, meta


def find_min_max(values) -> tuple[np.float32, np.float32]:
    """quantile.rs:5-19 find_min_max_from_iter over all values (no values: (f32::MAX, f32::MIN))."""
    v = _f32(values).reshape(-1)
    mn, mx = C.c_float(), C.c_float()
    lib().qo_find_min_max(_p(v), v.size, C.byref(mn), C.byref(mx))
    return np.float32(mn.value), np.float32(mx.value)


def alpha_offset(mn, mx) -> tuple[np.float32, np.float32]:
    """encoded_vectors_u8.rs:228-232 alpha_offset_from_min_max."""
    a, o = C.c_float(), C.c_float()
    lib().qo_u8_alpha_offset(C.c_float(mn), C.c_float(mx), C.byref(a), C.byref(o))
    return np.float32(a.value), np.float32(o.value)


def find_quantile_interval(data, quantile: float):
    """quantile.rs:21-71 for count <= 100 000 (every vector is in the sample) -> (min, max) or None."""
    data = _f32(data)
    count, dim = data.shape
    mn, mx = C.c_float(), C.c_float()
    r = lib().qo_find_quantile_interval(_p(data), dim, count, C.c_float(quantile), C.byref(mn), C.byref(mx))
    if r < 0:
        raise ValueError("count > 100000: a random sample in the reference")
    return (np.float32(mn.value), np.float32(mx.value)) if r == 1 else None


def u8_encode_empty(dim: int, distance: int, invert: bool):
    rows = np.zeros((0, u8_actual_dim(dim) + 4), dtype=np.uint8)
    meta = Meta()
    lib().qo_u8_encode(None, 0, dim, distance, int(invert), -1.0, None, C.byref(meta))
    return rows, meta


def u8_encode_with(data, distance: int, invert: bool, alpha: float, offset: float):
    data = _f32(data)
    count, dim = data.shape
    rows = np.zeros((count, u8_actual_dim(dim) + 4), dtype=np.uint8)
    meta = Meta()
    lib().qo_u8_encode_with(_p(data), count, dim, distance, int(invert),
                            C.c_float(alpha), C.c_float(offset), _p(rows), C.byref(meta))
    return rows, meta


def u8_encode_query(meta: Meta, query):
    query = _f32(query)
    codes = np.zeros(u8_actual_dim(query.shape[0]), dtype=np.uint8)
    off = lib().qo_u8_encode_query(C.byref(meta), _p(query), query.shape[0], _p(codes))
    return codes, np.float32(off)


def u8_score_all(meta: Meta, rows, codes, qoffset, order: int = ORDER_SIMPLE,
                 use_ref: bool = False, begin: int = 0, end: int | None = None) -> np.ndarray:
    rows = _u8(rows)
    codes = _u8(codes)
    end = int(meta.count) if end is None else end
    out = np.zeros(end - begin, dtype=np.float32)
    rd = rl = None
    if use_ref:
        R = ref()
        if R is None:
            raise RuntimeError("oracle/_ref is not built")
        rd = C.cast(R.impl_score_dot_avx, C.c_void_p)
        rl = C.cast(R.impl_score_l1_avx, C.c_void_p)
    lib().qo_u8_score_all(C.byref(meta), _p(rows), _p(codes), C.c_float(qoffset), begin, end,
                          order, rd, rl, _p(out))
    return out


def u8_score_point(meta: Meta, rows, codes, qoffset, i: int, order: int = ORDER_SIMPLE) -> np.float32:
    rows = _u8(rows)
    codes = _u8(codes)
    return np.float32(lib().qo_u8_score_point(C.byref(meta), _p(rows), _p(codes),
                                              C.c_float(qoffset), i, order))


def u8_score_internal(meta: Meta, rows, i: int, j: int, order: int = ORDER_SIMPLE) -> np.float32:
    rows = _u8(rows)
    return np.float32(lib().qo_u8_score_internal(C.byref(meta), _p(rows), i, j, order))


# ------------------------------------------------------------------------- binary
def bin_row_bytes(dim: int, store: int = STORE_U8) -> int:
    return int(lib().qo_bin_row_bytes(dim, store))


def bin_encode(data, store: int = STORE_U8) -> np.ndarray:
    data = _f32(data)
    count, dim = data.shape
    rows = np.zeros((count, bin_row_bytes(dim, store)), dtype=np.uint8)
    lib().qo_bin_encode(_p(data), count, dim, store, _p(rows))
    return rows


def bin_score_all(rows, q, dim: int, distance: int, invert: bool, store: int = STORE_U8,
                  use_ref: bool = False, begin: int = 0, end: int | None = None) -> np.ndarray:
    rows = _u8(rows)
    q = _u8(q)
    n = rows.shape[0] if end is None else end
    out = np.zeros(n - begin, dtype=np.float32)
    rp = None
    if use_ref:
        R = ref()
        if R is None:
            raise RuntimeError("oracle/_ref is not built")
        rp = C.cast(R.impl_xor_popcnt_sse_uint128, C.c_void_p)
    lib().qo_bin_score_all(_p(rows), _p(q), dim, store, distance, int(invert), begin, n, rp, _p(out))
    return out


def bin_score_internal(rows, dim: int, distance: int, invert: bool, i: int, j: int,
                       store: int = STORE_U8) -> np.float32:
    rows = _u8(rows)
    return np.float32(lib().qo_bin_score_internal(_p(rows), dim, store, distance, int(invert), i, j))


# ----------------------------------------------------------------------------- PQ
def pq_chunks(dim: int, chunk_size: int) -> int:
    return int(lib().qo_pq_chunks(dim, chunk_size))


def pq_centroids_small(data) -> np.ndarray:
    data = _f32(data)
    count, dim = data.shape
    assert count <= 256
    cen = np.zeros((256, dim), dtype=np.float32)
    lib().qo_pq_centroids_small(_p(data), count, dim, _p(cen))
    return cen


def pq_encode(data, chunk_size: int, centroids) -> np.ndarray:
    data = _f32(data)
    centroids = _f32(centroids)
    count, dim = data.shape
    rows = np.zeros((count, pq_chunks(dim, chunk_size)), dtype=np.uint8)
    lib().qo_pq_encode(_p(data), count, dim, chunk_size, _p(centroids), _p(rows))
    return rows


def pq_encode_query(query, chunk_size: int, centroids, distance: int, invert: bool) -> np.ndarray:
    query = _f32(query)
    centroids = _f32(centroids)
    dim = query.shape[0]
    lut = np.zeros(pq_chunks(dim, chunk_size) * 256, dtype=np.float32)
    lib().qo_pq_encode_query(_p(query), dim, chunk_size, _p(centroids), distance, int(invert), _p(lut))
    return lut


def pq_score_all(rows, lut, order: int = ORDER_SSE, begin: int = 0, end: int | None = None) -> np.ndarray:
    rows = _u8(rows)
    lut = _f32(lut)
    n, m = rows.shape
    n = n if end is None else end
    out = np.zeros(n - begin, dtype=np.float32)
    lib().qo_pq_score_all(_p(rows), m, _p(lut), begin, n, order, _p(out))
    return out


def pq_score_internal(rows, dim: int, chunk_size: int, centroids, distance: int, invert: bool,
                      i: int, j: int) -> np.float32:
    rows = _u8(rows)
    centroids = _f32(centroids)
    return np.float32(lib().qo_pq_score_internal(_p(rows), dim, chunk_size, _p(centroids),
                                                 distance, int(invert), i, j))


def kmeans(data, max_threads: int = 1, chunk_index: int = 0, max_iterations: int = 100, accuracy: float = 1e-5,
           trace: bool = False):
    """kmeans.rs:7-47 on [n, dim] sub-vectors -> (centroids [256, dim], iterations, empty re-seeds[, trace])."""
    data = _f32(data)
    n, dim = data.shape
    cen = np.zeros((256, dim), dtype=np.float32)
    it, em = C.c_uint32(), C.c_uint32()
    tr = np.zeros((max_iterations, n), dtype=np.uint32) if trace else None
    rc = lib().qo_kmeans(_p(data), n, dim, 256, max_iterations, max_threads, accuracy, chunk_index, _p(cen),
                         C.byref(it), C.byref(em), _p(tr) if trace else None)
    assert rc == 0
    return (cen, it.value, em.value, tr[: it.value]) if trace else (cen, it.value, em.value)


def pq_sample_rows(count: int, sample_size: int = 10_000) -> np.ndarray:
    """The product's deterministic stand-in for the reference's random Permutor sample
    (encoded_vectors_pq.rs:300-307): evenly strided rows floor(k * count / S), ascending."""
    S = min(sample_size, count)
    return (np.arange(S, dtype=np.uint64) * np.uint64(count)) // np.uint64(S)


def find_centroids(data, chunk_size: int, sample_rows=None, max_threads: int = 1):
    """encoded_vectors_pq.rs:278-342 given the sampled rows -> (centroids [256, dim],
    iterations per chunk, empty re-seeds)."""
    data = _f32(data)
    count, dim = data.shape
    rows = np.ascontiguousarray(pq_sample_rows(count) if sample_rows is None else sample_rows, dtype=np.uint64)
    cen = np.zeros((256, dim), dtype=np.float32)
    its = np.zeros(pq_chunks(dim, chunk_size), dtype=np.uint32)
    em = C.c_uint32()
    rc = lib().qo_find_centroids(_p(data), count, dim, chunk_size, _p(rows), rows.size, max_threads, _p(cen), _p(its),
                                 C.byref(em))
    assert rc == 0
    return cen, its, em.value


def topk_heap(scores, k: int = 30):
    """The caller's k-entry max-heap (demos/src/ann_benchmark_data.rs:151-167): k SMALLEST scores,
    ascending -> (ids, scores)."""
    scores = _f32(scores)
    ids = np.zeros(k, dtype=np.uint32)
    sc = np.zeros(k, dtype=np.float32)
    n = lib().qo_topk_heap(_p(scores), scores.size, k, _p(ids), _p(sc))
    return ids[:n], sc[:n]


def metric_f32(distance: int, a, b) -> np.float32:
    a = _f32(a)
    b = _f32(b)
    return np.float32(lib().qo_metric_f32(distance, _p(a), _p(b), a.shape[0]))
