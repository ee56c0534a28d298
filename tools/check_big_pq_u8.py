"""One-off stress check: PQ 200M x m=32 and u8 250M x 64 dims (row counts near 2^28, byte offsets
far beyond 2^32): top-k against torch.topk on score_all, score_ids at the far end of the store."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)

def check(name, enc, q, n, largest):
    out = torch.empty(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter(); enc.score_all(q, out=out); torch.cuda.synchronize(); t1 = time.perf_counter()
    ids, sc = enc.topk(q, 40, largest=largest)
    best = torch.topk(out, 40, largest=largest).values.cpu().numpy()
    ok_scores = np.array_equal(sc, best)
    ok_ids = np.array_equal(out[torch.from_numpy(ids.astype(np.int64)).to(dev)].cpu().numpy(), sc)
    far = np.array([n - 1, n - 2, n // 2, 0], dtype=np.uint32)
    ok_far = np.array_equal(enc.score_ids(q, far), out[torch.from_numpy(far.astype(np.int64)).to(dev)].cpu().numpy())
    print(f"{name}: scan {1e3*(t1-t0):.2f} ms; topk scores {ok_scores}, ids consistent {ok_ids}, score_ids at the ends {ok_far}", flush=True)

n, dim, chunk = 200_000_000, 256, 8
vp = qa.VectorParameters(dim, n, qa.DistanceType.L2, False)
rows = torch.randint(0, 256, (n, dim // chunk), generator=g, device=dev, dtype=torch.uint8)
cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
enc = qa.EncodedVectorsPQ.from_storage(rows, vp, chunk, cen)
del rows
check(f"pq {n} x m={dim // chunk}", enc, enc.encode_query(torch.rand(dim, generator=g, device=dev)), n, False)
del enc
n, dim = 250_000_000, 64
data = torch.rand((n, dim), generator=g, device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
check(f"u8 {n} x {dim}", enc, enc.encode_query(torch.rand(dim, generator=g, device=dev)), n, True)
