"""Latency of the per-pair API calls (score_point / score_internal / small score_ids) — the
reference's own call granularity, kept for compatibility."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import quantization_amd as qa
n, dim = 200_000, 768
rng = np.random.default_rng(0)
data = rng.random((n, dim), dtype=np.float32)
for name, enc in (("u8", qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))),
                  ("bin", qa.EncodedVectorsBin.encode(data - 0.5, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)))):
    q = enc.encode_query(rng.random(dim, dtype=np.float32))
    for _ in range(100): enc.score_point(q, 5)
    t0 = time.perf_counter()
    for i in range(2000): enc.score_point(q, i)
    t1 = time.perf_counter()
    ids = rng.integers(0, n, 64).astype(np.uint32)
    for _ in range(50): enc.score_ids(q, ids)
    t2 = time.perf_counter()
    for _ in range(2000): enc.score_ids(q, ids)
    t3 = time.perf_counter()
    print(f"{name}: score_point {(t1 - t0) / 2000 * 1e6:.1f} us/call   score_ids(64 host ids) {(t3 - t2) / 2000 * 1e6:.1f} us/call", flush=True)
