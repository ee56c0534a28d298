"""Time topk_batch for a list of batch sizes on one store (developer A/B driver: run it under
QAMD_LIB_PATH=<other build> to compare two builds on the same box)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
nqs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,128,256,1024").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 768
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
torch.cuda.synchronize()
time.sleep(3)
out = []
for nq in nqs:
    batch = enc.encode_query_batch(torch.rand((nq, dim), device=dev))
    ids = torch.empty(nq * 30, dtype=torch.int32, device=dev)
    sc = torch.empty(nq * 30, dtype=torch.float32, device=dev)
    for _ in range(12):  # the first ~10 calls of a new shape run 5-20 % slower (clock ramp, pool warm-up)
        enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    out.append(f"{nq}q {np.median(ts):.3f}")
print(os.environ.get("QAMD_LIB_PATH", "tree build"), "|", "  ".join(out), flush=True)
