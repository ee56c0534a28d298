"""Minimal driver to put after `rocprofv3 ... --`: a few topk_batch calls of one batch size on a 10M x 768 store."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim, nq = 10_000_000, 768, int(os.environ.get("NQ", 1024))
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
batch = enc.encode_query_batch(torch.rand((nq, dim), device=dev))
ids = torch.empty(nq * 30, dtype=torch.int32, device=dev)
sc = torch.empty(nq * 30, dtype=torch.float32, device=dev)
for _ in range(4):
    enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc)
torch.cuda.synchronize()
