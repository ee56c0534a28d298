"""One-off stress check of the batched u8 kernels on stores whose byte offsets go far beyond 2^32:
topk_batch (row-streaming: 16 queries, query-streaming: 1024 queries; ping-pong forced by the caller
with QAMD_GEMM_CFG=p) against the exact single-query top-k, on 60M x 768 (46 GB of codes) and
24M x 1536 (37 GB)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(11)
for n, dim in ((60_000_000, 768), (24_000_000, 1536)):
    vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
    # encode in 4M-row batches (the f32 data of the whole store would be 184 GB)
    def batches():
        gg = torch.Generator(device=dev); gg.manual_seed(3)
        for i in range(0, n, 4_000_000):
            yield torch.rand((min(4_000_000, n - i), dim), generator=gg, device=dev)
    enc = qa.EncodedVectorsU8.encode_stream(batches, vp)
    for nq in (16, 1024):
        queries = torch.rand((nq, dim), generator=g, device=dev)
        batch = enc.encode_query_batch(queries)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, sc = enc.topk_batch(batch, 30)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        qh = queries.cpu().numpy()
        ok = True
        far = 0
        for qi in (0, nq // 2, nq - 1):
            wi, ws = enc.topk(enc.encode_query(qh[qi]), 30)
            ok &= bool(np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32)))
            far = max(far, int(wi.max()))
        print(f"u8 {n} x {dim}, {nq} queries: topk_batch {1e3 * (t1 - t0):.1f} ms, equal to the single-query top-k: {ok} (largest id seen {far})",
              flush=True)
    del enc
