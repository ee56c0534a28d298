"""Profiling driver: PQ encode INCLUDING k-means training (200k x 768, chunk 8: 96 chunks, 10 000-row sample,
up to 100 iterations), two calls -- put after `rocprofv3 --kernel-trace --stats ... --` (tools/kstats.sh)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim = 200_000, 768
data = torch.rand((n, dim), device=dev)
vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc = qa.EncodedVectorsPQ.encode(data, vp, 8, max_kmeans_threads=int(os.environ.get("WORKERS", 1)))
    torch.cuda.synchronize()
    print(f"encode incl. k-means: {(time.perf_counter() - t0) * 1e3:.1f} ms, iterations/empties {enc.kmeans_info}", flush=True)
    del enc
