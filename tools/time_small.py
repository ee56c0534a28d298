"""Latency of whole-store calls on SMALL stores (where per-call host overhead, not HBM, decides)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
for n in (10_000, 100_000, 1_000_000):
    dim = 768
    data = torch.rand((n, dim), device=dev)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
    q = enc.encode_query(torch.rand(dim, device=dev))
    out = torch.empty(n, device=dev)
    ids = torch.empty(30, dtype=torch.int32, device=dev); sc = torch.empty(30, device=dev)
    qh = np.random.default_rng(0).random(dim, dtype=np.float32)
    def t(f, reps=300):
        for _ in range(30): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    a = t(lambda: enc.score_all(q, out=out))
    b = t(lambda: enc.topk(q, 30, out_ids=ids, out_scores=sc))
    c = t(lambda: enc.topk(enc.encode_query(qh), 30))               # a fresh query object per search (the reference's shape)
    qr = enc.encode_query(qh)
    d = t(lambda: enc.topk(enc.encode_query(qh, reuse=qr), 30))   # recycled query object
    print(f"n={n}: score_all(dev) {a:.1f} us   topk(dev out) {b:.1f} us   encode_query(host)+topk(host out) {c:.1f} us"
          f"   same, query object reused {d:.1f} us", flush=True)

# the other two quantizers on a small store (single-launch path)
n, dim = 100_000, 768
rows = torch.randint(0, 256, (n, 96), device=dev, dtype=torch.uint8)
cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
penc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False), 8, cen)
pq = penc.encode_query(torch.rand(dim, device=dev))
ids = torch.empty(30, dtype=torch.int32, device=dev); sc = torch.empty(30, device=dev); out = torch.empty(n, device=dev)
print(f"pq m96 n={n}: score_all {t(lambda: penc.score_all(pq, out=out)):.1f} us   topk(dev out) "
      f"{t(lambda: penc.topk(pq, 30, out_ids=ids, out_scores=sc)):.1f} us", flush=True)
brows = torch.randint(0, 256, (n, 128), device=dev, dtype=torch.uint8)
benc = qa.EncodedVectorsBin.from_storage(brows, qa.VectorParameters(1024, n, qa.DistanceType.Dot, False))
bq = benc.encode_query(torch.randn(1024, device=dev))
print(f"binary 1024 n={n}: score_all {t(lambda: benc.score_all(bq, out=out)):.1f} us   topk(dev out) "
      f"{t(lambda: benc.topk(bq, 30, out_ids=ids, out_scores=sc)):.1f} us", flush=True)
