"""Driver for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_small_topk.py`: the single-launch
top-k on small stores (100k and 1M rows x 768), device outputs, 200 calls each."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
for n in (100_000, 1_000_000):
    data = torch.rand((n, 768), device=dev)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(768, n, qa.DistanceType.Dot, False))
    q = enc.encode_query(torch.rand(768, device=dev))
    ids = torch.empty(30, dtype=torch.int32, device=dev); sc = torch.empty(30, device=dev)
    out = torch.empty(n, device=dev)
    for _ in range(200):
        enc.topk(q, 30, out_ids=ids, out_scores=sc)
    for _ in range(200):
        enc.score_all(q, out=out)
    torch.cuda.synchronize()
