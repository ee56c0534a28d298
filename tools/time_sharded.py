"""Overhead of the single-process sharded handle (csrc/sharded.hip) on ONE GPU: the same store behind
1, 2, 8 logical shards against the plain handle.  With every shard on one device the scans serialise,
so what this shows is the cost of the fan-out to the worker threads, the peer-copy stand-in and the
device-side merge -- not a speed-up."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
for n in (1_000_000, 10_000_000):
    dim = 768
    data = torch.rand((n, dim), device=dev)
    vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
    one = qa.EncodedVectorsU8.encode(data, vp)
    qh = np.random.default_rng(0).random(dim, dtype=np.float32)
    def t(f, reps=100):
        for _ in range(10): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    q1 = one.encode_query(qh)
    out = torch.empty(n, device=dev)
    line = [f"n={n}: plain handle topk(30, host out) {t(lambda: one.topk(q1, 30)):.0f} us, score_all(dev) {t(lambda: one.score_all(q1, out=out)):.0f} us"]
    for G in (1, 2, 8):
        sh = qa.ShardedVectorsU8.encode(data, vp, [0] * G)
        qs = sh.encode_query(qh)
        line.append(f"G={G}: topk {t(lambda: sh.topk(qs, 30)):.0f} us, score_all(dev) {t(lambda: sh.score_all(qs, out=out)):.0f} us, "
                    f"encode_query {t(lambda: sh.encode_query(qh, reuse=qs)):.0f} us")
        del sh, qs
    print(" | ".join(line), flush=True)
    del one, data
