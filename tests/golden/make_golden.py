#!/usr/bin/env python3
"""Generate tests/golden/pair_kernels.npz from the reference's own C kernels.

Run in the build container only (needs /root/reference to compile oracle/_ref):

    python tests/golden/make_golden.py

The fixture holds DATA only: seeded input byte rows and the outputs that the
compiled reference functions (quantization/cpp/avx2.c, cpp/sse.c) returned for
them.  It pins oracle/qoracle.c's pair kernels; the GPU box replays it without
the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import qoracle as qo  # noqa: E402

DIMS = [16, 32, 48, 80, 128, 768, 1024, 1040, 1056, 1536, 2048, 4096]
PAIRS_PER_CASE = 6


def main() -> None:
    R = qo.ref()
    if R is None:
        raise SystemExit("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(20241004)
    out = {}
    case_names = []
    for dim in DIMS:
        for kind in ("codes127", "bytes255", "all127", "sparse"):
            name = f"{kind}_d{dim}"
            if kind == "codes127":
                q = rng.integers(0, 128, size=(PAIRS_PER_CASE, dim), dtype=np.uint8)
                v = rng.integers(0, 128, size=(PAIRS_PER_CASE, dim), dtype=np.uint8)
            elif kind == "bytes255":  # outside the encoder's range: maddubs sign/saturation
                q = rng.integers(0, 256, size=(PAIRS_PER_CASE, dim), dtype=np.uint8)
                v = rng.integers(0, 256, size=(PAIRS_PER_CASE, dim), dtype=np.uint8)
            elif kind == "all127":
                q = np.full((1, dim), 127, dtype=np.uint8)
                v = np.full((1, dim), 127, dtype=np.uint8)
            else:
                q = (rng.integers(0, 128, size=(PAIRS_PER_CASE, dim)) *
                     (rng.random((PAIRS_PER_CASE, dim)) < 0.05)).astype(np.uint8)
                v = rng.integers(0, 128, size=(PAIRS_PER_CASE, dim), dtype=np.uint8)
            n = q.shape[0]
            res = {k: np.zeros(n, dtype=np.float32) for k in ("dot_avx", "dot_sse", "l1_avx")}
            pop = np.zeros(n, dtype=np.uint32)
            for i in range(n):
                qp, vp = q[i].ctypes.data, v[i].ctypes.data
                res["dot_avx"][i] = R.impl_score_dot_avx(qp, vp, dim)
                res["dot_sse"][i] = R.impl_score_dot_sse(qp, vp, dim)
                res["l1_avx"][i] = R.impl_score_l1_avx(qp, vp, dim)
                pop[i] = R.impl_xor_popcnt_sse_uint128(qp, vp, dim // 16)
            out[f"{name}__q"] = q
            out[f"{name}__v"] = v
            for k, a in res.items():
                out[f"{name}__{k}"] = a
            out[f"{name}__popcnt128"] = pop
            case_names.append(name)
    # small-row popcount entry points (encoded_vectors_binary.rs:57-69)
    q8 = rng.integers(0, 256, size=(8, 16), dtype=np.uint8)
    v8 = rng.integers(0, 256, size=(8, 16), dtype=np.uint8)
    out["small__q"] = q8
    out["small__v"] = v8
    out["small__popcnt64x2"] = np.array(
        [R.impl_xor_popcnt_sse_uint64(q8[i].ctypes.data, v8[i].ctypes.data, 2) for i in range(8)],
        dtype=np.uint32)
    out["small__popcnt32x2"] = np.array(
        [R.impl_xor_popcnt_sse_uint32(q8[i].ctypes.data, v8[i].ctypes.data, 2) for i in range(8)],
        dtype=np.uint32)
    out["case_names"] = np.array(case_names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pair_kernels.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(case_names)} cases, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
