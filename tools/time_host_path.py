"""Host query in -> host results out on a small store, calling the C ABI directly through ctypes with
pre-bound arguments (what a Rust / C caller pays), next to the Python mirror's figure in time_small.py."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantization_amd as qa
from quantization_amd import _lib
dev = torch.device("cuda", 0)
L = qa.lib()
for n in (100_000, 1_000_000):
    dim = 768
    data = torch.rand((n, dim), device=dev)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
    qh = np.random.default_rng(0).random(dim, dtype=np.float32)
    ids = np.empty(30, np.uint32); sc = np.empty(30, np.float32)
    qobj = C.c_void_p()
    qp, ip, sp = C.c_void_p(qh.ctypes.data), C.c_void_p(ids.ctypes.data), C.c_void_p(sc.ctypes.data)
    def reuse():
        L.qamd_u8_encode_query(enc._h, qp, dim, 0, None, C.byref(qobj))
        L.qamd_u8_topk(enc._h, qobj, 30, 1, ip, sp, 0, None)
    def fresh():
        q = C.c_void_p()
        L.qamd_u8_encode_query(enc._h, qp, dim, 0, None, C.byref(q))
        L.qamd_u8_topk(enc._h, q, 30, 1, ip, sp, 0, None)
        L.qamd_u8_query_free(q)
    out = []
    for f in (reuse, fresh):
        for _ in range(100): f()
        t0 = time.perf_counter()
        for _ in range(2000): f()
        out.append((time.perf_counter() - t0) / 2000 * 1e6)
    print(f"n={n}: encode_query(host) + topk(30, host out): query object reused {out[0]:.1f} us, fresh per search {out[1]:.1f} us", flush=True)
