"""One-off experiment kept for reference: how returning a large allocation to the driver (empty_cache) slows the scans that follow (VRAM scrub)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim = 10_000_000, 768
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
if os.environ.get("EMPTY_CACHE", "1") == "1":
    torch.cuda.empty_cache()
queries = torch.rand((16, dim), device=dev)
q = enc.encode_query(queries[0])
outs = [torch.empty(n, device=dev) for _ in range(2)]

def run(name, body, reps=60):
    for i in range(5): body(i)
    torch.cuda.synchronize()
    ev = []
    for i in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pre = body.__defaults__[0] if body.__defaults__ else None
        if pre: pre(i)
        a.record(); enc.score_all(q, out=outs[body.__kwdefaults__["alt"] and i % 2 or 0]); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in ev)
    print(f"{name:50s} median {ts[len(ts)//2]:.4f} mean {sum(ts)/len(ts):.4f}")

def mk(pre, alt):
    def body(i, pre=pre, *, alt=alt):
        if pre: pre(i)
        enc.score_all(q, out=outs[i % 2 if alt else 0])
    return body

for rnd in range(2):
    run("plain, one out buffer", mk(None, False))
    run("both", mk(lambda i: enc.encode_query(queries[i % 16], reuse=q), True))
big = torch.empty(64 << 20, device=dev)  # 256 MiB block; carve the score buffers out of it
outs = [big[: n], big[16 << 20: (16 << 20) + n]]
run("out buffers carved from one 256 MiB allocation", mk(None, True))
