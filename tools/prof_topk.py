"""Minimal driver to put after `rocprofv3 ... --`: single-query top-k calls on a 10M x 768 store."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim = 10_000_000, 768
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
q = enc.encode_query(torch.rand(dim, device=dev))
ids = torch.empty(30, dtype=torch.int32, device=dev)
sc = torch.empty(30, dtype=torch.float32, device=dev)
out = torch.empty(n, dtype=torch.float32, device=dev)
for _ in range(20):
    enc.topk(q, 30, out_ids=ids, out_scores=sc)
    enc.score_all(q, out=out)
torch.cuda.synchronize()
