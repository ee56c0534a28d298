"""PQ single-query top-k at config 2's shape (10M x 768, m = 96): ms per topk(30) call, device outputs, and per score_all call.
QAMD_PQ_SKEW=0 selects the older scan kernel for comparison."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quantization_amd as qa
D = qa.DistanceType
dev = torch.device("cuda", 0)
for n, dim, chunk in ((10_000_000, 768, 8), (12_500_000, 1536, 8)):
    m = dim // chunk
    rows = torch.randint(0, 256, (n, m), device=dev, dtype=torch.uint8)
    cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
    enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, D.Dot, False), chunk, cen)
    del rows
    q = enc.encode_query(torch.rand(dim, device=dev))
    ids = torch.empty(30, dtype=torch.int32, device=dev)
    sc = torch.empty(30, dtype=torch.float32, device=dev)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn in (("topk(30)", lambda: enc.topk(q, 30, out_ids=ids, out_scores=sc)), ("score_all", lambda: enc.score_all(q, out=out))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        print(f"{n} x {dim} m={m} {name}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per call")
    del enc, out
