#!/usr/bin/env python3
"""Headline benchmark: scored vectors/s of the u8 scalar-quantized dot scan, 10M x 768 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one query against the whole (sharded) store: encode_query -> score_all over the
rank's shard -> the exchange of per-shard results.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line (metric, value, roofline, cpu_baseline, ...).

Workload (BASELINE.json configs[1]): f32 i.i.d. uniform [0,1) vectors (demos/benches/encode.rs:17-22),
fixed seed, scalar-u8 encoded on the GPU by the library itself; queries from the same
distribution.  Weak scaling: every rank holds --rows-per-gpu rows (default 10M).

roofline.achieved = (actual_dim + 4) algorithmic bytes per row (772 B at dim 768)
x rows one launch scores / that kernel's mean duration, measured here with HIP events on the
stream the scan is launched on (torch's current stream, handed to the C ABI).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows-per-gpu", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--distance", choices=["dot", "l2"], default="dot")
    ap.add_argument("--quantizer", choices=["u8", "binary", "pq"], default="u8",
                    help="u8 = the headline metric (BASELINE configs[1]); binary / pq run configs[3] / [2] "
                         "through the same harness (e.g. --quantizer binary --dim 1024 --rows-per-gpu 6250000)")
    ap.add_argument("--pq-chunk", type=int, default=8)
    ap.add_argument("--exchange", choices=["scores", "topk", "none"], default="scores",
                    help="per-query result exchange across ranks (N>1): gather of per-shard scores "
                         "to rank 0 (overlapped with the next scan), per-shard top-k + all-gather, or none")
    ap.add_argument("--gather-root", default="rotate",
                    help="--exchange scores: 'rotate' (step i gathers to rank i %% N: consecutive gathers use "
                         "disjoint inbound xGMI links) or a rank number (every step to that rank)")
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--queries", type=int, default=16)
    ap.add_argument("--batch-queries", type=int, default=0,
                    help="opt-in second workload (BASELINE config 4 shape): per step, top-k of this many queries "
                         "at once over the shard on the matrix cores (u8 only), then one all-gather of "
                         "world*Q*k pairs and a merge; the default 0 runs the headline single-query scan")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; "
                    "gloo only to rehearse the multi-rank code path, e.g. several ranks on one GPU)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only: put every rank on this one device (needs --backend gloo)")
    return ap.parse_args()


def cpu_baseline(qa, enc, data_sample, queries, gpu_scores_sample, dist_id):
    """Times the reference's caller loop (encode_query once, score_point for every row,
    demos/src/ann_benchmark.rs:247-252) on the host: the oracle's loop driving the REFERENCE's
    own compiled impl_score_dot_avx (oracle/_ref) when present ("reference"), else the
    oracle's restatement ("port").  Also checks the GPU scores of those rows bit-for-bit."""
    import threading

    import numpy as np

    from oracle import qoracle as qo

    md = enc.metadata
    vp = md["vector_parameters"]
    S = data_sample.shape[0]
    # Encode the sample with the store's own (alpha, offset): byte-identical to its first S rows.
    sub = qa.EncodedVectorsU8.encode(data_sample, qa.VectorParameters(vp.dim, S, vp.distance_type, vp.invert),
                                     alpha_offset=(float(md["alpha"]), float(md["offset"])))
    rows = sub.storage_bytes()
    meta = qo.Meta(md["actual_dim"], float(md["alpha"]), float(md["offset"]), float(md["multiplier"]),
                   vp.dim, S, dist_id, int(vp.invert))
    kind = "reference" if qo.ref() is not None else "port"
    use_ref = kind == "reference"
    codes, qoff = qo.u8_encode_query(meta, queries[0])
    want = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2, use_ref=use_ref)
    parity = bool(np.array_equal(want.view(np.uint32), gpu_scores_sample.view(np.uint32)))

    def one_core():
        t0 = time.perf_counter()
        c, o = qo.u8_encode_query(meta, queries[0])
        qo.u8_score_all(meta, rows, c, o, order=qo.ORDER_AVX2, use_ref=use_ref)
        return time.perf_counter() - t0

    one_core()  # warm-up (criterion-like: warm-up + >= 10 samples, median)
    samples = []
    t_budget = time.perf_counter()
    while len(samples) < 10 or (time.perf_counter() - t_budget < 8.0 and len(samples) < 40):
        samples.append(one_core())
    t1 = float(np.median(samples))

    cores = os.cpu_count() or 1
    nthreads = max(1, min(cores, 64))

    def all_cores():
        c, o = qo.u8_encode_query(meta, queries[0])
        bounds = [(S * i) // nthreads for i in range(nthreads + 1)]
        ts = [threading.Thread(target=qo.u8_score_all, args=(meta, rows, c, o),
                               kwargs=dict(order=qo.ORDER_AVX2, use_ref=use_ref, begin=bounds[i], end=bounds[i + 1]))
              for i in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return time.perf_counter() - t0

    all_cores()
    tn = float(np.median([all_cores() for _ in range(10)]))
    cpu_model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {
        "value": S / t1, "unit": "vectors/s", "cores": 1, "kind": kind,
        "sample": f"first {S} rows of the same store, 1 query, median of {len(samples)} passes of the "
                  f"reference loop (encode_query + score_point per row, "
                  f"{'compiled reference impl_score_dot_avx' if use_ref else 'oracle restatement'}); "
                  f"host: {cpu_model}, {cores} logical cores",
        "all_cores": {"value": S / tn, "cores": nthreads},
        "gpu_matches_cpu_bits": parity,
    }


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import quantization_amd as qa
    from quantization_amd import _lib
    from quantization_amd.sharded import ScoreGather, ShardedTopK

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    dev_index = local_rank if args.all_ranks_on_device is None else args.all_ranks_on_device
    torch.cuda.set_device(dev_index)
    L = _lib.lib()
    if L.qamd_set_device(dev_index) != 0:
        raise SystemExit(L.qamd_last_error().decode())
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.backend)

    n, dim = args.rows_per_gpu, args.dim
    dtype = qa.DistanceType.Dot if args.distance == "dot" else qa.DistanceType.L2
    total_rows = n * world

    # ---- synthetic store, generated and encoded on the GPU (never timed) -------------------
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    qgen = torch.Generator(device=dev)
    qgen.manual_seed(43)
    sample_rows = min(args.cpu_sample_rows, n)
    data_sample = None
    if args.quantizer == "u8":
        data = torch.rand((n, dim), generator=gen, device=dev, dtype=torch.float32)
        # one global (alpha, offset): data is U[0,1) on every rank, so use the analytic interval
        # [0, 1) -> alpha = 1/127, offset = 0 rather than a cross-rank min/max reduction.
        alpha_offset = (float(np.float32(1.0) / np.float32(127.0)), 0.0) if world > 1 else None
        vp = qa.VectorParameters(dim, n, dtype, False)
        enc = qa.EncodedVectorsU8.encode(data, vp, alpha_offset=alpha_offset)
        queries = torch.rand((args.queries, dim), generator=qgen, device=dev, dtype=torch.float32)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            data_sample = data[:sample_rows].cpu().numpy()
        del data  # stays in torch's caching allocator on purpose: returning the 30.7 GB block to the
        # driver (empty_cache) measured 3.5 % SLOWER scans afterwards on the same box (1.171 vs 1.131 ms,
        # tools/exp_ctx.py), whichever buffers the scores were then written to.
        bytes_per_row = enc.scan_bytes_per_row()
        kernel_name = "u8_scan_kernel"
    elif args.quantizer == "binary":
        # +-1 style data (demos/benches/binary.rs:10-21): random bit rows are exactly that, packed
        vp = qa.VectorParameters(dim, n, dtype, False)
        nb = qa.EncodedVectorsBin.get_quantized_vector_size_from_params(vp)
        rows = torch.randint(0, 256, (n, nb), generator=gen, device=dev, dtype=torch.uint8)
        enc = qa.EncodedVectorsBin.from_storage(rows, vp)
        del rows
        queries = torch.randn((args.queries, dim), generator=qgen, device=dev, dtype=torch.float32)
        bytes_per_row = nb
        kernel_name = "bin_scan_kernel"
    else:
        vp = qa.VectorParameters(dim, n, dtype, False)
        m = qa.EncodedVectorsPQ.get_quantized_vector_size(vp, args.pq_chunk)
        rows = torch.randint(0, 256, (n, m), generator=gen, device=dev, dtype=torch.uint8)
        cen = np.random.default_rng(7).random((256, dim), dtype=np.float32)
        enc = qa.EncodedVectorsPQ.from_storage(rows, vp, args.pq_chunk, cen)
        del rows
        queries = torch.rand((args.queries, dim), generator=qgen, device=dev, dtype=torch.float32)
        bytes_per_row = m
        kernel_name = "pq_scan_fast_kernel"
    args.no_cpu_baseline = args.no_cpu_baseline or args.quantizer != "u8"

    if args.batch_queries > 0:
        if args.quantizer != "u8":
            raise SystemExit("--batch-queries is the scalar-u8 multi-query path")
        from quantization_amd.sharded import ShardedTopKBatch
        Q, k = args.batch_queries, args.k
        bq = torch.rand((Q, dim), generator=qgen, device=dev, dtype=torch.float32)
        batch = enc.encode_query_batch(bq)
        xchg = ShardedTopKBatch(dist, torch, Q, k, dev, rank, world, total_rows)
        ids, sc = xchg.buffers()

        def bstep():
            enc.topk_batch(batch, k, largest=True, out_ids=ids, out_scores=sc)
            return xchg.exchange(largest=True)

        for _ in range(max(1, args.warmup)):
            bstep()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            bstep()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64,
                         device=dev if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            ad = enc.metadata["actual_dim"]
            ops = 2.0 * Q * n * ad  # per GPU and step
            per_gpu_tops = ops * args.steps / elapsed / 1e12
            print(json.dumps({
                "metric": f"(query, vector) pairs scored/sec, {Q} queries x {n}x{dim} u8 dot per GPU, top-{k} each",
                "value": float(Q) * total_rows * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
                "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 x u8 -> i32 (MFMA int8)",
                "data": "synthetic",
                "config": {"workload": f"{Q} queries x {n} x {dim} scalar-u8 rows per GPU, per step: topk_batch over the "
                                       f"shard + all-gather of world*Q*k pairs + per-query merge (host)",
                           "rows_per_gpu": n, "dim": dim, "queries": Q, "k": k, "total_rows": total_rows},
                "roofline": {"bound": "mfma", "achieved": per_gpu_tops, "peak": 5000.0, "unit": "TFLOP/s",
                             "frac": per_gpu_tops / 5000.0, "traffic": None,
                             "note": "int8 op/s per GPU over the whole step (sample pass, filter GEMM, scatter, "
                                     "sort, exchange); algorithmic ops = 2 * actual_dim per (query, row) pair"},
            }), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    qobj = enc.encode_query(queries[0])
    gather = topk = None
    if args.exchange == "scores":
        gather = ScoreGather(dist, torch, n, dev, rank, world,
                             dst=None if args.gather_root == "rotate" else int(args.gather_root))
    elif args.exchange == "topk":
        topk = ShardedTopK(dist, torch, args.k, dev, rank, world, total_rows)
    scores_plain = torch.empty(n, dtype=torch.float32, device=dev) if gather is None else None

    ev_pairs = []

    def step(i, timed):
        q = queries[i % args.queries]
        enc.encode_query(q, reuse=qobj)
        out = gather.slot(i) if gather is not None else scores_plain
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        enc.score_all(qobj, out=out)
        if timed:
            e1.record()
            ev_pairs.append((e0, e1))
        if gather is not None:
            gather.submit(i)
        elif topk is not None:
            ids, sc = topk.buffers()  # per-shard selection over the fresh scores, then k pairs/rank
            qa.topk_scores(out, n, args.k, largest=True, out_ids=ids, out_scores=sc)
            topk.exchange(largest=True)

    for i in range(args.warmup):
        step(i, False)
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kern_all = [a.elapsed_time(b) for a, b in ev_pairs]
    kern_ms = float(np.mean(kern_all)) if kern_all else float("nan")
    if os.environ.get("QAMD_BENCH_TRACE") and rank == 0:  # developer: per-step kernel times
        with open(os.environ["QAMD_BENCH_TRACE"], "w") as f:
            f.write("\n".join(f"{x:.4f}" for x in kern_all))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rows * args.steps / elapsed
        achieved = bytes_per_row * n / (kern_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                    "kernel": kernel_name, "kernel_ms": kern_ms,
                    "algorithmic_bytes_per_row": bytes_per_row, "rows_per_launch": n}
        # measured traffic from the committed PMC profile of this exact workload, if present
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_u8_scan.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if (args.quantizer == "u8" and j.get("rows_per_launch") == n
                        and j.get("algorithmic_read_bytes_per_launch") == n * bytes_per_row):
                    roofline["traffic"] = j.get("traffic_bytes_per_launch")
                    roofline["traffic_source"] = "profiles/r01_pmc_u8_scan.json (rocprofv3 --pmc, FETCH_SIZE x2 per guide)"
            except Exception:
                pass
        # on-box streaming-read ceiling with the same load shape (16 B/lane, nt)
        try:
            probe = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
            probe.zero_()
            scratch = torch.empty(1 << 16, dtype=torch.uint8, device=dev)
            s = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                L.qamd_stream_read(probe.data_ptr(), probe.numel(), scratch.data_ptr(), s)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                L.qamd_stream_read(probe.data_ptr(), probe.numel(), scratch.data_ptr(), s)
            b.record()
            torch.cuda.synchronize()
            roofline["stream_read_ceiling_GBps"] = probe.numel() * 10 / (a.elapsed_time(b) * 1e-3) / 1e9
            del probe
        except Exception as e:  # pragma: no cover
            roofline["stream_read_ceiling_GBps"] = None
            roofline["stream_read_error"] = str(e)

        result = {
            "metric": "scored vectors/sec, 10Mx768 u8 dot"
            if (n == 10_000_000 and dim == 768 and args.distance == "dot" and args.quantizer == "u8")
            else f"scored vectors/sec, {n}x{dim} {args.quantizer} {args.distance}",
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"u8": "u8", "binary": "u1 (xor+popcount, i32)", "pq": "f32 (LUT adds)"}[args.quantizer],
            "data": "synthetic",
            "config": {"workload": (f"{n} x {dim} f32 U[0,1) per GPU -> scalar u8 ({args.distance}); "
                                    if args.quantizer == "u8" else
                                    f"{n} x {dim} {args.quantizer} rows per GPU ({bytes_per_row} B/row, {args.distance}); ") +
                                   f"per step: encode_query + score_all over the shard"
                                   + (f" + {args.exchange} exchange" if world > 1 and args.exchange != "none" else ""),
                       "quantizer": args.quantizer, "rows_per_gpu": n, "dim": dim, "distance": args.distance,
                       "exchange": args.exchange,
                       "gather_root": args.gather_root if (world > 1 and args.exchange == "scores") else None,
                       "total_rows": total_rows, "queries": args.queries},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                torch.cuda.synchronize()
                enc.encode_query(queries[0], reuse=qobj)
                full = enc.score_all(qobj, out=torch.empty(n, dtype=torch.float32, device=dev))
                torch.cuda.synchronize()
                gpu_sample = full[:sample_rows].cpu().numpy()
                result["cpu_baseline"] = cpu_baseline(qa, enc, data_sample, queries.cpu().numpy(), gpu_sample,
                                                      0 if args.distance == "dot" else 2)
            except Exception as e:
                result["cpu_baseline"] = {"value": None, "unit": "vectors/s", "cores": 0, "kind": "port",
                                          "sample": f"failed: {e}"}
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
