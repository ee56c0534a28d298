"""One-off stress check: topk_batch on a 100M x 144 u8 store (row offsets far beyond 2^32 bytes) against the exact single-query top-k."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim = 100_000_000, 144
g = torch.Generator(device=dev); g.manual_seed(1)
data = torch.rand((n, dim), generator=g, device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.L2, False))
del data
for nq in (140, 33):
    q = torch.rand((nq, dim), generator=g, device=dev)
    b = enc.encode_query_batch(q)
    t0 = time.perf_counter()
    ids, sc = enc.topk_batch(b, 20, largest=False)
    t1 = time.perf_counter()
    bad = 0
    for qi in (0, 7, nq - 1):
        wi, ws = enc.topk(enc.encode_query(q[qi]), 20, largest=False)
        if not (np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32))): bad += 1
    print(f"n={n} dim={dim} nq={nq}: topk_batch {1e3*(t1-t0):.1f} ms, mismatching queries: {bad}", flush=True)
