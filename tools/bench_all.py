#!/usr/bin/env python3
"""Per-kernel measurements on one MI355X for DESIGN.md (not the driver's bench line):
u8 dot/L2/L1 scans, binary scan, PQ scan, top-k, random-access ids, encoders.
Prints one JSON object per line."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import json
import time
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import quantization_amd as qa  # noqa: E402

dev = torch.device("cuda", 0)
D = qa.DistanceType


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def report(name, rows, bytes_per_row, med, mn, **kw):
    print(json.dumps({"kernel": name, "rows": rows, "bytes_per_row": bytes_per_row, "median_ms": round(med, 4),
                      "min_ms": round(mn, 4), "Gvec_per_s": round(rows / med / 1e6, 3),
                      "GBps": round(rows * bytes_per_row / med / 1e6, 1),
                      "frac_of_8TBps": round(rows * bytes_per_row / med / 1e6 / 8000, 4), **kw}), flush=True)


which = set(sys.argv[1:]) or {"u8", "bin", "pq", "topk", "ids", "encode", "batch"}

if "u8" in which or "topk" in which or "ids" in which:
    for dim, n in ((768, 10_000_000), (1536, 5_000_000), (128, 20_000_000), (1024, 8_000_000)):
        if dim != 768 and "u8" not in which:
            continue
        for dist in (D.Dot, D.L2, D.L1):
            if dim != 768 and dist != D.Dot:
                continue
            data = torch.rand((n, dim), device=dev)
            enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, False))
            del data
            q = enc.encode_query(torch.rand(dim, device=dev))
            out = torch.empty(n, dtype=torch.float32, device=dev)
            if "u8" in which:
                med, mn = timeit(lambda: enc.score_all(q, out=out))
                report(f"u8_scan {dist.name} dim{dim}", n, enc.scan_bytes_per_row(), med, mn)
            if dim == 768 and dist == D.Dot:
                if "topk" in which:
                    ids = torch.empty(30, dtype=torch.int32, device=dev)
                    sc = torch.empty(30, dtype=torch.float32, device=dev)
                    med, mn = timeit(lambda: enc.topk(q, 30, out_ids=ids, out_scores=sc), reps=10)
                    report("u8_topk30 (scan+select) dim768", n, enc.scan_bytes_per_row(), med, mn)
                    med, mn = timeit(lambda: qa.topk_scores(out, n, 30, out_ids=ids, out_scores=sc), reps=10)
                    report("topk30 select only", n, 4, med, mn)
                if "ids" in which:
                    rid = torch.randint(0, n, (1_000_000,), device=dev, dtype=torch.int32)
                    o2 = torch.empty(1_000_000, dtype=torch.float32, device=dev)
                    med, mn = timeit(lambda: enc.score_ids(q, rid, out=o2))
                    report("u8_score_ids random 1M dim768", 1_000_000, enc.scan_bytes_per_row(), med, mn)
                if "u8" in which:
                    enc.set_lane_mode(1)
                    med, mn = timeit(lambda: enc.score_all(q, out=out))
                    report("u8_scan avx2-lane-order mode dim768", n, enc.scan_bytes_per_row(), med, mn)
            del enc, out
            pass  # no empty_cache: returning tens of GB to the driver starts a VRAM scrub that slows (up to 4x) whatever runs next

if "bin" in which:
    for dim, n in ((1024, 50_000_000), (1536, 30_000_000), (128, 100_000_000)):
        vp = qa.VectorParameters(dim, n, D.Dot, False)
        nb = qa.EncodedVectorsBin.get_quantized_vector_size_from_params(vp)
        rows = torch.randint(0, 256, (n, nb), device=dev, dtype=torch.uint8)
        enc = qa.EncodedVectorsBin.from_storage(rows, vp)
        del rows
        q = enc.encode_query(torch.randn(dim, device=dev))
        out = torch.empty(n, dtype=torch.float32, device=dev)
        med, mn = timeit(lambda: enc.score_all(q, out=out))
        report(f"bin_scan dim{dim}", n, nb, med, mn, note="bytes_per_row excludes the 4 B score write")
        del enc, out
        pass  # no empty_cache: returning tens of GB to the driver starts a VRAM scrub that slows (up to 4x) whatever runs next

if "pq" in which:
    for dim, chunk, n in ((768, 8, 10_000_000), (768, 4, 5_000_000), (128, 8, 20_000_000)):
        vp = qa.VectorParameters(dim, n, D.Dot, False)
        m = qa.EncodedVectorsPQ.get_quantized_vector_size(vp, chunk)
        rows = torch.randint(0, 256, (n, m), device=dev, dtype=torch.uint8)
        cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
        enc = qa.EncodedVectorsPQ.from_storage(rows, vp, chunk, cen)
        del rows
        q = enc.encode_query(torch.rand(dim, device=dev))
        out = torch.empty(n, dtype=torch.float32, device=dev)
        med, mn = timeit(lambda: enc.score_all(q, out=out), reps=10)
        report(f"pq_scan dim{dim} m{m}", n, m, med, mn, lut_bytes=m * 1024)
        med, mn = timeit(lambda: enc.encode_query(torch.rand(dim, device=dev), reuse=q), reps=10)
        report(f"pq_encode_query (LUT build) dim{dim} m{m}", 1, m * 1024, med, mn)
        del enc, out
        pass  # no empty_cache: returning tens of GB to the driver starts a VRAM scrub that slows (up to 4x) whatever runs next

if "encode" in which:
    n, dim = 2_000_000, 768
    data = torch.rand((n, dim), device=dev)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    med, mn = timeit(lambda: qa.EncodedVectorsU8.encode(data, vp), reps=5, warm=1)
    report("u8_encode (minmax + quantize) dim768", n, dim * 4 * 2 + dim + 4, med, mn,
           note="bytes: two passes over f32 input + codes written; includes alloc of the store")
    med, mn = timeit(lambda: qa.EncodedVectorsBin.encode(data, vp), reps=5, warm=1)
    report("bin_encode dim768", n, dim * 4 + dim // 8, med, mn)
    cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
    n2 = 500_000
    vp2 = qa.VectorParameters(dim, n2, D.Dot, False)
    med, mn = timeit(lambda: qa.EncodedVectorsPQ.encode(data[:n2], vp2, 8, centroids=cen), reps=3, warm=1)
    report("pq_encode given centroids dim768 m96", n2, dim * 4 + 96, med, mn,
           note=f"{256 * dim * 3 * n2 / med / 1e9:.2f} TFLOP/s f32 VALU (sub, mul, add)")
    n3 = 200_000
    med, mn = timeit(lambda: qa.EncodedVectorsPQ.encode(data[:n3], qa.VectorParameters(dim, n3, D.Dot, False), 8),
                     reps=2, warm=1)
    report("pq_encode incl. k-means training dim768 m96", n3, dim * 4 + 96, med, mn)

if "batch" in which:
    for dim, n in ((768, 10_000_000), (1536, 12_500_000)):
        data = torch.rand((n, dim), device=dev)
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
        del data
        pass  # no empty_cache: returning tens of GB to the driver starts a VRAM scrub that slows (up to 4x) whatever runs next
        torch.cuda.synchronize()
        time.sleep(3)  # let the driver finish clearing what the encode allocated / freed
        for nq in [int(x) for x in os.environ.get("BATCH_NQ", "2,3,4,5,16,64,128,256,512,768,1024,2048").split(",")]:
            queries = torch.rand((nq, dim), device=dev)
            batch = enc.encode_query_batch(queries)
            ids = torch.empty(nq * 30, dtype=torch.int32, device=dev)
            sc = torch.empty(nq * 30, dtype=torch.float32, device=dev)
            med, mn = timeit(lambda: enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc), reps=9, warm=12)
            ops = 2.0 * nq * n * enc.metadata["actual_dim"]
            print(json.dumps({"kernel": f"u8_topk_batch k30 dim{dim}", "rows": n, "queries": nq,
                              "median_ms": round(med, 3), "min_ms": round(mn, 3),
                              "query_rows_per_s": round(nq * n / med / 1e6, 2), "unit": "G (query,row) pairs/s",
                              "int8_TOPs": round(ops / med / 1e9, 1),
                              "frac_of_5000_TOPs_dense_i8_peak": round(ops / med / 1e9 / 5000, 4),
                              "frac_of_3650_TOPs_measured_mfma_issue_ceiling": round(ops / med / 1e9 / 3650, 4),
                              "store_GBps_if_read_once": round(n * enc.scan_bytes_per_row() / med / 1e6, 1),
                              "speedup_vs_single_query_loop_at_1.12ms": round(nq * 1.12 * (n / 1e7) * (dim / 768) / med, 1)}),
                  flush=True)
        del enc
        pass  # no empty_cache: returning tens of GB to the driver starts a VRAM scrub that slows (up to 4x) whatever runs next

if "binbatch" in which:
    # several queries per row read: where does the binary scan stop being HBM-bound?
    for dim, n in ((1024, 50_000_000), (2048, 20_000_000)):
        vp = qa.VectorParameters(dim, n, D.Dot, False)
        nb = qa.EncodedVectorsBin.get_quantized_vector_size_from_params(vp)
        rows = torch.randint(0, 256, (n, nb), device=dev, dtype=torch.uint8)
        enc = qa.EncodedVectorsBin.from_storage(rows, vp)
        del rows
        for nq in (1, 2, 4, 8, 16):
            batch = enc.encode_query_batch(torch.randn((nq, dim), device=dev))
            out = torch.empty(nq * n, dtype=torch.float32, device=dev)
            med, mn = timeit(lambda: enc.score_batch(batch, out=out), reps=10)
            print(json.dumps({"kernel": f"bin_score_batch dim{dim}", "rows": n, "queries": nq, "median_ms": round(med, 4),
                              "min_ms": round(mn, 4), "G_pairs_per_s": round(nq * n / med / 1e6, 2),
                              "hbm_GBps_algorithmic": round(n * (nb * ((nq + 7) // 8 if nq > 1 else 1) + 4 * nq) / med / 1e6, 1),
                              "vs_single_query_loop": round(nq * 1.0, 2)}), flush=True)
            del out
        batch = enc.encode_query_batch(torch.randn((64, dim), device=dev))
        ids = torch.empty(64 * 30, dtype=torch.int32, device=dev)
        sc = torch.empty(64 * 30, dtype=torch.float32, device=dev)
        med, mn = timeit(lambda: enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc), reps=5, warm=2)
        q1 = enc.encode_query(torch.randn(dim, device=dev))
        i1 = torch.empty(30, dtype=torch.int32, device=dev)
        s1 = torch.empty(30, dtype=torch.float32, device=dev)
        med1, _ = timeit(lambda: enc.topk(q1, 30, out_ids=i1, out_scores=s1), reps=10)
        print(json.dumps({"kernel": f"bin_topk_batch k30 dim{dim}", "rows": n, "queries": 64, "median_ms": round(med, 3),
                          "ms_per_query": round(med / 64, 4), "single_query_topk_ms": round(med1, 4)}), flush=True)
        del enc

if "pqbatch" in which:
    for dim, chunk, n in ((768, 8, 10_000_000), (1536, 8, 12_500_000)):
        vp = qa.VectorParameters(dim, n, D.Dot, False)
        m = qa.EncodedVectorsPQ.get_quantized_vector_size(vp, chunk)
        rows = torch.randint(0, 256, (n, m), device=dev, dtype=torch.uint8)
        cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
        enc = qa.EncodedVectorsPQ.from_storage(rows, vp, chunk, cen)
        del rows
        q1 = enc.encode_query(torch.rand(dim, device=dev))
        i1 = torch.empty(30, dtype=torch.int32, device=dev)
        s1 = torch.empty(30, dtype=torch.float32, device=dev)
        med1, _ = timeit(lambda: enc.topk(q1, 30, out_ids=i1, out_scores=s1), reps=10)
        for nq in (16, 64, 256):
            queries = torch.rand((nq, dim), device=dev)
            med_e, _ = timeit(lambda: enc.encode_query_batch(queries), reps=5)
            batch = enc.encode_query_batch(queries)
            ids = torch.empty(nq * 30, dtype=torch.int32, device=dev)
            sc = torch.empty(nq * 30, dtype=torch.float32, device=dev)
            med, mn = timeit(lambda: enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc), reps=5, warm=2)
            print(json.dumps({"kernel": f"pq_topk_batch k30 dim{dim} m{m}", "rows": n, "queries": nq,
                              "median_ms": round(med, 3), "ms_per_query": round(med / nq, 4),
                              "single_query_topk_ms": round(med1, 4), "encode_query_batch_ms": round(med_e, 4),
                              "G_pairs_per_s": round(nq * n / med / 1e6, 2)}), flush=True)
        del enc
