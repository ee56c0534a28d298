"""Timing of the burst calls (one launch for many random-access pairs): 64 lists x 32 ids with device lists and
outputs (enqueue-only: time per call over a back-to-back run), the same with host lists / host outputs (mapped
scratch, one synchronisation), and 1M random pairs (kernel time by events -> fraction of the HBM roofline).
u8 768 / binary 1024 / PQ m = 96 on 10M rows."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantization_amd as qa
D = qa.DistanceType
dev = torch.device("cuda", 0)
n = int(os.environ.get("ROWS", 10_000_000))
g = torch.Generator(device=dev); g.manual_seed(1)
rng = np.random.default_rng(0)


def bench(name, enc, dim, row_bytes, make_queries):
    nl, per = 64, 32
    offs = np.arange(0, nl * per + 1, per, dtype=np.uint32)
    ids = rng.integers(0, n, nl * per).astype(np.uint32)
    rows = rng.integers(0, n, nl).astype(np.uint32)
    t = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)
    d_offs, d_ids, d_rows = t(offs), t(ids), t(rows)
    out = torch.empty(nl * per, dtype=torch.float32, device=dev)
    batch = enc.encode_query_batch(make_queries(nl))

    def loop(f, reps=2000):
        for _ in range(50): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

    res = {"quantizer": name, "rows": n, "dim": dim}
    res["score_ids_batch 64x32, device lists + device out (us/call, back to back)"] = round(loop(lambda: enc.score_ids_batch(batch, d_offs, d_ids, out=out)), 2)
    res["score_internal_ids_batch 64x32, device (us/call)"] = round(loop(lambda: enc.score_internal_ids_batch(d_rows, d_offs, d_ids, out=out)), 2)
    res["score_ids_batch 64x32, host lists -> host scores (us/call)"] = round(loop(lambda: enc.score_ids_batch(batch, offs, ids), 500), 2)
    res["score_internal_ids 32 host ids -> host (us/call)"] = round(loop(lambda: enc.score_internal_ids(5, ids[:32]), 500), 2)
    # 1M random pairs: 1024 lists x 1024 ids
    nl2, per2 = 1024, 1024
    big_offs = t(np.arange(0, nl2 * per2 + 1, per2, dtype=np.uint32))
    big_ids = torch.randint(0, n, (nl2 * per2,), generator=g, device=dev, dtype=torch.int32)
    big_rows = torch.randint(0, n, (nl2,), generator=g, device=dev, dtype=torch.int32)
    big_out = torch.empty(nl2 * per2, dtype=torch.float32, device=dev)
    big_batch = enc.encode_query_batch(make_queries(nl2))
    for label, f in (("score_ids_batch", lambda: enc.score_ids_batch(big_batch, big_offs, big_ids, out=big_out)),
                     ("score_internal_ids_batch", lambda: enc.score_internal_ids_batch(big_rows, big_offs, big_ids, out=big_out))):
        for _ in range(5): f()
        evs = []
        for _ in range(10):  # 8 launches between one pair of events: the event records' own latency is not kernel time
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8): f()
            b.record(); evs.append((a, b))
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs])) / 8
        gbps = nl2 * per2 * (row_bytes + 8) / (ms * 1e-3) / 1e9  # row bytes + the id read + the score written
        res[f"{label} 1M random pairs"] = {"ms": round(ms, 4), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 3)}
    print(json.dumps(res), flush=True)


data = torch.rand((n, 768), generator=g, device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(768, n, D.Dot, False))
del data
bench("u8", enc, 768, 772, lambda q: torch.rand((q, 768), generator=g, device=dev))
del enc
rows = torch.randint(0, 256, (n, 128), generator=g, device=dev, dtype=torch.uint8)
enc = qa.EncodedVectorsBin.from_storage(rows, qa.VectorParameters(1024, n, D.Dot, False))
del rows
bench("binary", enc, 1024, 128, lambda q: torch.randn((q, 1024), generator=g, device=dev))
del enc
rows = torch.randint(0, 256, (n, 96), generator=g, device=dev, dtype=torch.uint8)
cen = rng.random((256, 768), dtype=np.float32)
enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(768, n, D.Dot, False), 8, cen)
del rows
bench("pq", enc, 768, 96, lambda q: torch.rand((q, 768), generator=g, device=dev))
