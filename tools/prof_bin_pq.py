"""Profiling driver for the binary and PQ kernels -- put after `rocprofv3 ... --` (profiles/collect_bin_pq.sh):
binary 50M x 1024: 4 x score_all (bin_scan_kernel), 4 x score_batch of 4 queries (bin_scan_multi_kernel),
3 x topk_batch(30) of 64 queries (bin_gemm_rs_kernel on the matrix cores); PQ 10M x 768, m = 96: 4 x score_all
(pq_scan_fast_kernel).  PART=bin|pq runs one half."""
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
part = os.environ.get("PART", "all")
if part in ("all", "bin"):
    n, dim = 50_000_000, 1024
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    rows = torch.randint(0, 256, (n, 128), device=dev, dtype=torch.uint8)
    enc = qa.EncodedVectorsBin.from_storage(rows, vp)
    del rows
    q = enc.encode_query(torch.randn(dim, device=dev))
    out = torch.empty(4 * n, dtype=torch.float32, device=dev)
    for _ in range(4):
        enc.score_all(q, out=out[:n])
    b4 = enc.encode_query_batch(torch.randn((4, dim), device=dev))
    for _ in range(4):
        enc.score_batch(b4, out=out)
    b64 = enc.encode_query_batch(torch.randn((64, dim), device=dev))
    ids = torch.empty(64 * 30, dtype=torch.int32, device=dev)
    sc = torch.empty(64 * 30, dtype=torch.float32, device=dev)
    for _ in range(3):
        enc.topk_batch(b64, 30, out_ids=ids, out_scores=sc)
    torch.cuda.synchronize()
    del enc, out
if part in ("all", "pq"):
    n, dim = 10_000_000, 768
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    rows = torch.randint(0, 256, (n, 96), device=dev, dtype=torch.uint8)
    cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
    enc = qa.EncodedVectorsPQ.from_storage(rows, vp, 8, cen)
    del rows
    q = enc.encode_query(torch.rand(dim, device=dev))
    out = torch.empty(n, dtype=torch.float32, device=dev)
    for _ in range(4):
        enc.score_all(q, out=out)
    torch.cuda.synchronize()
