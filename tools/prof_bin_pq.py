"""Profiling driver (rocprofv3 --pmc): binary 50M x 1024 and PQ 10M x 768 m=96 scans, a few launches each."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quantization_amd as qa
D = qa.DistanceType
dev = torch.device("cuda", 0)
n, dim = 50_000_000, 1024
vp = qa.VectorParameters(dim, n, D.Dot, False)
rows = torch.randint(0, 256, (n, 128), device=dev, dtype=torch.uint8)
enc = qa.EncodedVectorsBin.from_storage(rows, vp)
del rows
q = enc.encode_query(torch.randn(dim, device=dev))
out = torch.empty(n, dtype=torch.float32, device=dev)
for _ in range(4):
    enc.score_all(q, out=out)
torch.cuda.synchronize()
del enc
n, dim = 10_000_000, 768
vp = qa.VectorParameters(dim, n, D.Dot, False)
rows = torch.randint(0, 256, (n, 96), device=dev, dtype=torch.uint8)
cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
enc = qa.EncodedVectorsPQ.from_storage(rows, vp, 8, cen)
del rows
q = enc.encode_query(torch.rand(dim, device=dev))
for _ in range(4):
    enc.score_all(q, out=out[:n])
torch.cuda.synchronize()
