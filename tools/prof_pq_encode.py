"""Profiling driver: PQ encode with given centroids (500k x 768, chunk 8), three calls."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim = 500_000, 768
data = torch.rand((n, dim), device=dev)
cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
for _ in range(3):
    enc = qa.EncodedVectorsPQ.encode(data, vp, 8, centroids=cen)
    torch.cuda.synchronize()
    del enc
