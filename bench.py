#!/usr/bin/env python3
"""Headline benchmark: scored vectors/s of the u8 scalar-quantized dot scan over a 10M x 768 store.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Launch.  `python bench.py --gpus N` with N > 1 and no rank environment starts its own ranks: the
parent, before it touches torch.cuda or HIP, runs `python -m torch.distributed.run --nproc-per-node N
... bench.py <same flags>` as a CHILD process (never exec), relays rank 0's JSON line and exits with
the child's code; under torch.distributed.run (RANK/WORLD_SIZE set) it is a rank.  `--single-process`
instead drives all N GPUs from this one process through the row-sharded C ABI (`qamd_u8_sharded_*`,
what INTEGRATION.md tells a single-process caller like Qdrant to hold) and prints the same line.

One "step" = one query against the whole store: encode_query -> score_all over the rank's shard ->
the exchange of per-shard results.  Inputs are resident in HBM before the timed region.  Rank 0
prints ONE JSON line (metric, value, roofline, cpu_baseline, ...).

Workload (BASELINE.json configs[1]): f32 i.i.d. uniform [0,1) vectors (demos/benches/encode.rs:17-22),
fixed seed, scalar-u8 encoded on the GPU by the library itself; queries from the same distribution.

Scaling.  N = 1: the whole store (--rows, default 10M) on one GPU.  N > 1 defaults to STRONG
scaling — the store is FIXED at --rows and row-sharded, rank g holding rows [g*R/N, (g+1)*R/N)
(1.25M rows = 965 MB per GPU at N = 8) — because that is what "10M x 768 ... >= 6x shard-parallel
speedup at 8 GPUs" measures.  `--scaling weak` keeps --rows per GPU instead.

roofline.achieved = (actual_dim + 4) algorithmic bytes per row (772 B at dim 768)
x rows one launch scores / that kernel's mean duration, measured here with HIP events on the
stream the scan is launched on (torch's current stream, handed to the C ABI).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E vendor peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
LDS_B32_PEAK_GBPS = 75000.0  # ds_read_b32, every CU streaming at ~2.4 GHz (MI355X_MICROARCH.md, LDS section)
MFMA_INT8_PEAK_TOPS = 5000.0  # dense int8 (the guide's ~5 POP/s; vendor figures with sparsity are not used)
MFMA_FP4_PEAK_TOPS = 10000.0  # dense fp4 through the f8f6f4 MFMAs (2 x the fp8 rate): binary batches of 129+ queries run there


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", "--rows-per-gpu", dest="rows", type=int, default=10_000_000,
                    help="rows of the store: the WHOLE store under strong scaling (N>1 default), rows per GPU "
                         "under --scaling weak; at N=1 the two are the same thing")
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="auto = strong for N>1 (fixed store, row-sharded: the >=6x-at-8-GPUs target), n/a at N=1")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--distance", choices=["dot", "l2"], default="dot")
    ap.add_argument("--quantizer", choices=["u8", "binary", "pq"], default="u8",
                    help="u8 = the headline metric (BASELINE configs[1]); binary / pq run configs[3] / [2] "
                         "through the same harness (e.g. --quantizer binary --dim 1024 --rows 50000000)")
    ap.add_argument("--pq-chunk", type=int, default=8)
    ap.add_argument("--exchange", choices=["auto", "scores", "topk", "none"], default="auto",
                    help="per-query result exchange across ranks (N>1): gather of per-shard scores "
                         "(overlapped with the next scan), per-shard top-k + all-gather + device merge, or none.  "
                         "auto = scores for u8 / pq (the north_star's 'RCCL gather of per-shard scores'), topk for "
                         "binary at N>1: a 128 B/row scan is faster than the 4 B/row score gather over xGMI "
                         "(SURVEY 8e: 25 MB per rank and query against a 0.125 ms shard scan at 50M x 1024)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive --gpus N devices from THIS process through the row-sharded C ABI "
                         "(qamd_*_sharded_*: worker threads per GPU, peer copies over xGMI, device-side top-k merge) "
                         "instead of one rank per GPU over RCCL")
    ap.add_argument("--devices", default=None,
                    help="--single-process: comma-separated device list, repeats allowed (logical shards, e.g. 0,0 "
                         "to rehearse two shards on a one-GPU box); default 0..N-1")
    ap.add_argument("--gather-root", default="rotate",
                    help="--exchange scores: 'rotate' (step i gathers to rank i %% N: consecutive gathers use "
                         "disjoint inbound xGMI links) or a rank number (every step to that rank)")
    ap.add_argument("--gather-group", type=int, default=0,
                    help="--exchange scores: queries whose per-shard scores travel in ONE collective (0 = auto: 1 at "
                         "N=1, 4 at N>1, where a shard scan is ~0.15 ms and a collective's launch latency would "
                         "otherwise be paid per query)")
    ap.add_argument("--encode-ahead", type=int, default=-1, choices=[-1, 0, 1],
                    help="1: encode_query of step i+1 runs on a side stream while the scan of step i runs (two query "
                         "objects); 0 / -1 (default): strictly in line, which measured faster")
    ap.add_argument("--time-every", type=int, default=0,
                    help="HIP-event-time the scan kernel every this many steps (0 = auto: every step at N=1, every "
                         "8th at N>1: an event pair costs ~10 us of launch gap, 7 %% of a 1.25M-row shard scan)")
    ap.add_argument("--k", type=int, default=30)
    ap.add_argument("--cpu-sample-rows", type=int, default=0,
                    help="rows of the store the CPU baseline scans (0 = all of them, capped at 10M)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--queries", type=int, default=16)
    ap.add_argument("--batch-queries", type=int, default=0,
                    help="opt-in second workload (BASELINE config 4 shape): per step, top-k of this many queries "
                         "at once over the shard on the matrix cores (u8 only), then one all-gather of "
                         "world*Q*k pairs and a device-side merge; the default 0 runs the headline single-query scan")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; "
                    "gloo only to rehearse the multi-rank code path, e.g. several ranks on one GPU)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None,
                    help="rehearsal only: put every rank on this one device (needs --backend gloo)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even at world size 1 (exercises the RCCL code path on one GPU)")
    return ap.parse_args()


def host_cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU
    box hands each job a share of a large host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(quantizer, enc, query, gpu_scores, dist_id, sample_rows, pq_chunk=None):
    """Times the reference's caller loop (encode_query once, score_point for every row,
    demos/src/ann_benchmark.rs:247-252) on the host over the store's OWN encoded rows, through the oracle's loop:
      u8      the REFERENCE's compiled impl_score_dot_avx (oracle/_ref; "reference"), else the restatement ("port")
      binary  the REFERENCE's compiled impl_xor_popcnt_sse_uint128 ("reference"), else the restatement
      pq      the restatement of score_point_sse (encoded_vectors_pq.rs:405-440; Rust intrinsics, not compilable
              here: "port")
    One core is what the reference does (it is single-threaded); the all-cores figure splits the rows over persistent
    worker threads.  Also checks every GPU score of those rows bit for bit."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from oracle import qoracle as qo

    S = sample_rows
    rows = enc.storage_rows(0, S)  # reference-format rows, straight from the store being benchmarked
    have_ref = qo.ref() is not None
    if quantizer == "u8":
        md = enc.metadata
        vp = md["vector_parameters"]
        meta = qo.Meta(md["actual_dim"], float(md["alpha"]), float(md["offset"]), float(md["multiplier"]),
                       vp.dim, S, dist_id, int(vp.invert))
        kind, what = ("reference", "compiled reference impl_score_dot_avx") if have_ref else ("port", "oracle restatement")

        def prepare():
            return qo.u8_encode_query(meta, query)

        def score(q, begin, end):
            return qo.u8_score_all(meta, rows, q[0], q[1], order=qo.ORDER_AVX2, use_ref=have_ref, begin=begin, end=end)
    elif quantizer == "binary":
        vp = enc.vector_parameters
        kind, what = (("reference", "compiled reference impl_xor_popcnt_sse_uint128") if have_ref else
                      ("port", "oracle restatement of xor_popcnt"))

        def prepare():
            return qo.bin_encode(query[None, :], qo.STORE_U128)[0]

        def score(q, begin, end):
            return qo.bin_score_all(rows, q, vp.dim, dist_id, bool(vp.invert), qo.STORE_U128, use_ref=have_ref,
                                    begin=begin, end=end)
    else:
        vp = enc.vector_parameters
        cen = enc.centroids
        kind, what = "port", "oracle restatement of score_point_sse (the reference's is Rust std::arch, not compilable here)"

        def prepare():
            return qo.pq_encode_query(query, pq_chunk, cen, dist_id, bool(vp.invert))

        def score(q, begin, end):
            return qo.pq_score_all(rows, q, order=qo.ORDER_SSE, begin=begin, end=end)

    want = score(prepare(), 0, S)
    parity = bool(np.array_equal(want.view(np.uint32), gpu_scores[:S].view(np.uint32)))

    def one_core():
        t0 = time.perf_counter()
        score(prepare(), 0, S)
        return time.perf_counter() - t0

    one_core()  # warm-up (criterion-like: warm-up + >= 10 samples, median)
    samples = []
    t_budget = time.perf_counter()
    while len(samples) < 10 or (time.perf_counter() - t_budget < 6.0 and len(samples) < 30):
        samples.append(one_core())
        if time.perf_counter() - t_budget > 20.0 and len(samples) >= 3:
            break
    t1 = float(np.median(samples))

    cores = os.cpu_count() or 1
    nthreads = max(1, min(usable_cores(), 64))
    bounds = [(S * i) // nthreads for i in range(nthreads + 1)]
    pool = ThreadPoolExecutor(max_workers=nthreads)  # persistent workers: no thread start inside a pass

    def all_cores():
        q = prepare()
        t0 = time.perf_counter()
        futs = [pool.submit(score, q, bounds[i], bounds[i + 1]) for i in range(nthreads)]
        for f in futs:
            f.result()
        return time.perf_counter() - t0

    all_cores()
    all_cores()
    tn = float(np.median([all_cores() for _ in range(10)]))
    pool.shutdown()
    return {
        "value": S / t1, "unit": "vectors/s", "cores": 1, "kind": kind,
        "sample": f"the first {S} rows of the benchmarked store itself, 1 query, median of {len(samples)} passes of "
                  f"the reference loop (encode_query + score_point per row, {what}); "
                  f"host: {host_cpu_model()}, {cores} logical cores, {usable_cores()} usable by this job",
        "all_cores": {"value": S / tn, "cores": nthreads,
                      "note": "same rows split over persistent worker threads, median of 10 passes"},
        "gpu_matches_cpu_bits": parity,
    }


def pmc_traffic(key, n, read_bytes_per_launch_unit):
    """HBM bytes per scan from a committed rocprofv3 --pmc profile of THIS workload and kernel source:
    profiles/r*_pmc_<key>.json (key: u8_scan, bin_scan, pq_scan_m<m>, u8_batch<Q>, bin_batch<Q>) names the commit it was
    taken at and a hash of the kernel source (profiles/summarize.py kernel_source_hash); a profile of other code, or of
    another row count, is not quoted."""
    import glob

    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from summarize import kernel_source_hash  # the one definition of "the source that was profiled"
        src_hash = kernel_source_hash(ROOT, key)
    except Exception:
        return None, None
    finally:
        sys.path.pop(0)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{key}*.json")), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        if (j.get("rows_per_launch") == n and j.get("algorithmic_read_bytes_per_launch") == read_bytes_per_launch_unit
                and j.get("kernel_source_sha256_16") == src_hash):
            return j.get("traffic_bytes_per_launch"), (f"{os.path.relpath(path, ROOT)} (rocprofv3 --pmc, FETCH_SIZE x2 "
                                                       f"per the guide; commit {j.get('commit', '?')})")
    return None, None


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args) -> int:
    """--gpus N > 1 without a rank environment: start the N ranks as a child torch.distributed.run
    (nothing in this process has touched the GPU yet, and it never will), relay its output, return its
    exit code."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    port = os.environ.get("MASTER_PORT") or str(free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    for line in child.stdout:  # rank 0's JSON line (and anything else the ranks print) goes straight through
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def resolve_exchange(args, world: int) -> tuple[str, str]:
    """(exchange, why) for this run: `auto` picks the mode that can meet the scaling target."""
    if args.exchange != "auto":
        return args.exchange, "given on the command line"
    if args.quantizer == "binary" and world > 1:
        return "topk", ("auto: binary rows are 128 B at dim 1024, the per-shard score gather (4 B/row over xGMI) "
                        "costs more than the scan itself (SURVEY 8e) - per-shard top-k + all-gather of world*k pairs")
    return "scores", "auto: the north_star's gather of per-shard scores (asynchronous, grouped, rotating root)"


def run_single_process(args):
    """All --gpus N devices driven from this one process through qamd_u8_sharded_* (csrc/sharded.hip):
    per step encode_query + score_all into a device buffer on devices[0] (exchange scores: each shard's
    scores peer-copied over xGMI into their slice) or + topk (per-shard top-k, G*k pairs peer-copied,
    one merge kernel on devices[0]).  Same JSON line as the rank form."""
    import numpy as np
    import torch

    import quantization_amd as qa
    from quantization_amd import _lib

    if args.quantizer != "u8" or args.batch_queries:
        raise SystemExit("--single-process runs the headline u8 single-query workload")
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit("--devices must name exactly --gpus devices")
    L = _lib.lib()
    G, dim, n_total = args.gpus, args.dim, args.rows
    scaling = "strong" if (args.scaling in ("auto", "strong")) else "weak"
    if scaling == "weak":
        n_total = args.rows * G
    dist_t = qa.DistanceType.Dot if args.distance == "dot" else qa.DistanceType.L2
    exchange, why = resolve_exchange(args, G)
    if exchange == "none":
        raise SystemExit("--single-process: the sharded handle always delivers (scores or top-k); use scores or topk")
    root = devices[0]
    torch.cuda.set_device(root)
    dev = torch.device("cuda", root)
    vp = qa.VectorParameters(dim, n_total, dist_t, False)
    alpha_offset = (float(np.float32(1.0) / np.float32(127.0)), 0.0)  # U[0,1): the analytic interval, as the rank form
    # The store is encoded piece by piece on devices[0] (the 30.7 GB of f32 never exist as one buffer), exported
    # as reference-format rows and adopted by the sharded handle with from_rows -- what a caller holding a store
    # encoded elsewhere (Qdrant's mmap) does; every shard peer-copies its own row range.  Never timed.
    stride = qa.EncodedVectorsU8.get_quantized_vector_size(vp)
    rows = torch.empty((n_total, stride), dtype=torch.uint8, device=dev)
    piece = 2_500_000
    for r0 in range(0, n_total, piece):
        nr = min(piece, n_total - r0)
        gen = torch.Generator(device=dev)
        gen.manual_seed(42 + r0 // piece)
        part = torch.rand((nr, dim), generator=gen, device=dev, dtype=torch.float32)
        enc = qa.EncodedVectorsU8.encode(part, qa.VectorParameters(dim, nr, dist_t, False), alpha_offset=alpha_offset)
        enc.storage_bytes(out=rows[r0:r0 + nr])
        meta = enc.metadata
        del part, enc
    meta["vector_parameters"] = vp
    sh = qa.ShardedVectorsU8.from_storage(rows, meta, devices)
    del rows
    qgen = torch.Generator(device=dev)
    qgen.manual_seed(43)
    queries = torch.rand((args.queries, dim), generator=qgen, device=dev, dtype=torch.float32)
    torch.cuda.synchronize()

    f_encode, f_score, f_topk = L.qamd_u8_sharded_encode_query, L.qamd_u8_sharded_score_all, L.qamd_u8_sharded_topk
    hq = C.c_void_p()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = torch.empty(max(n_total, 1), dtype=torch.float32, device=dev)
    ids = torch.empty(args.k, dtype=torch.int32, device=dev)
    sc = torch.empty(args.k, dtype=torch.float32, device=dev)
    q_ptrs = [C.c_void_p(queries[i].data_ptr()) for i in range(args.queries)]

    def step(i):
        st = f_encode(sh._h, q_ptrs[i % args.queries], dim, _lib.MEM_DEVICE, stream, C.byref(hq))
        if exchange == "scores":
            st |= f_score(sh._h, hq, C.c_void_p(out.data_ptr()), _lib.MEM_DEVICE, stream)
        else:
            st |= f_topk(sh._h, hq, args.k, 1, C.c_void_p(ids.data_ptr()), C.c_void_p(sc.data_ptr()), _lib.MEM_DEVICE, stream)
        if st:
            raise RuntimeError(L.qamd_last_error().decode())

    for i in range(20 + args.warmup):
        step(i)
    for d in set(devices):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)  # every sharded call is synchronous: its workers have finished when it returns
    for d in set(devices):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0

    # the dominant kernel, timed on shard 0's own handle with HIP events on the stream it is launched on
    view, _base = sh.shard(0)
    n0 = view.count
    with torch.cuda.device(devices[0]):
        q0 = view.encode_query(queries[0])
        o0 = torch.empty(max(n0, 1), dtype=torch.float32, device=dev)
        for _ in range(10):
            view.score_all(q0, out=o0)
        evs = []
        for _ in range(20):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            view.score_all(q0, out=o0)
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
    kern_all = [a.elapsed_time(b) for a, b in evs]
    kern_ms = float(np.mean(kern_all))
    bytes_per_row = view.scan_bytes_per_row()
    achieved = bytes_per_row * n0 / (kern_ms * 1e-3) / 1e9
    names = [torch.cuda.get_device_name(d) for d in devices]
    headline = (n_total == 10_000_000 and dim == 768 and args.distance == "dot")
    print(json.dumps({
        "metric": "scored vectors/sec, 10Mx768 u8 dot" if headline else f"scored vectors/sec, {n_total}x{dim} u8 {args.distance}",
        "value": n_total * args.steps / elapsed, "unit": "vectors/s", "n_gpus": G, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak" if (G == 1 or scaling == "weak") else "strong", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{n_total} x {dim} store row-sharded over {G} GPUs behind ONE handle "
                               f"(qamd_u8_sharded_*, {n0} rows on shard 0); per step: encode_query + "
                               + ("score_all into a device buffer on devices[0] (peer copies of 4 B/row)"
                                  if exchange == "scores" else f"topk({args.k}) (per-shard top-k, device-side merge)"),
                   "launch": "single-process", "devices": devices, "device_names": names, "quantizer": "u8",
                   "rows_per_gpu": n0, "dim": dim, "distance": args.distance, "exchange": exchange,
                   "exchange_reason": why, "total_rows": n_total, "queries": args.queries,
                   "shard_lanes": int(os.environ.get("QAMD_SHARD_LANES", "3"))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": "u8_scan_kernel", "kernel_ms": kern_ms,
                     "kernel_ms_min": float(np.min(kern_all)), "kernel_ms_median": float(np.median(kern_all)),
                     "kernel_ms_mean": kern_ms, "algorithmic_bytes_per_row": bytes_per_row, "rows_per_launch": n0,
                     "note": "shard 0's scan timed on its own handle after the timed region (the sharded calls launch "
                             "on the library's worker streams)"},
    }), flush=True)


def main():
    args = parse_args()
    if args.single_process:
        return run_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    import quantization_amd as qa
    from quantization_amd import _lib
    from quantization_amd.sharded import ScoreGather, ShardedTopK, max_shard_rows, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world  # the launcher's world size wins
    args.exchange, exchange_reason = resolve_exchange(args, world)
    dev_index = local_rank if args.all_ranks_on_device is None else args.all_ranks_on_device
    torch.cuda.set_device(dev_index)
    qa.set_device(dev_index)
    L = _lib.lib()
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {} if "RANK" in os.environ else {"rank": 0, "world_size": 1}
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend=args.backend, **kw)

    # What the process group really is, for the record: backend, the world size torch.distributed reports,
    # the RCCL version, and which device every rank sits on (all-gathered through the group itself).
    dist_info = {"launch": "torch.distributed.run ranks" if "RANK" in os.environ else "single rank",
                 "dist_backend": None, "world_size_seen": 1, "rccl_version": None,
                 "rank_devices": [f"cuda:{dev_index} {torch.cuda.get_device_name(dev_index)}"]}
    if use_dist:
        mine = {"rank": rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index),
                "uuid": str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", "")), "pid": os.getpid()}
        everyone = [None] * dist.get_world_size()
        dist.all_gather_object(everyone, mine)
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        dist_info.update({"dist_backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                          "rccl_version": rccl if args.backend == "nccl" else None,
                          "rank_devices": [f"rank {e['rank']}: cuda:{e['device']} {e['name']} {e['uuid']} pid {e['pid']}"
                                           for e in everyone]})

    scaling = args.scaling if args.scaling != "auto" else "strong"
    if world == 1:
        total_rows, row0, n = args.rows, 0, args.rows
    elif scaling == "strong":
        total_rows = args.rows
        row0, row1 = shard_range(total_rows, rank, world)
        n = row1 - row0
    else:
        total_rows, row0, n = args.rows * world, args.rows * rank, args.rows
    n_max = n if world == 1 else (max_shard_rows(total_rows, world) if scaling == "strong" else args.rows)
    dim = args.dim
    dtype = qa.DistanceType.Dot if args.distance == "dot" else qa.DistanceType.L2

    # ---- synthetic store, generated and encoded on the GPU (never timed) -------------------
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    qgen = torch.Generator(device=dev)
    qgen.manual_seed(43)
    if args.quantizer == "u8":
        data = torch.rand((n, dim), generator=gen, device=dev, dtype=torch.float32)
        vp = qa.VectorParameters(dim, n, dtype, False)
        if use_dist:
            # ONE global (alpha, offset) as the reference's encode finds it (encoded_vectors_u8.rs:57-71): every rank's
            # min / max over its own rows, folded by an all-reduce over the process group, then every rank quantises
            # its shard (quantization_amd/sharded.py encode_u8; tests/test_sharded_encode_gloo.py: shard bytes and
            # metadata equal the single-handle encode of the concatenated data)
            from quantization_amd.sharded import encode_u8 as distributed_encode_u8
            enc, _ = distributed_encode_u8(dist, torch, data, qa.VectorParameters(dim, total_rows, dtype, False))
            dist_info["encode"] = "distributed: per-rank find_min_max + all-reduce of the interval over the process group"
        else:
            enc = qa.EncodedVectorsU8.encode(data, vp)
        queries = torch.rand((args.queries, dim), generator=qgen, device=dev, dtype=torch.float32)
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
        kernel_name, pq_launches = enc.scan_kernel()  # the library says which scan kernel this store takes
        pq_skew = kernel_name.startswith("pq_scan_skew_kernel")
    scaling_field = "weak" if (world == 1 or scaling == "weak") else "strong"
    # (at N = 1 there is nothing to scale; the contract's field keeps its default)

    host_enqueue = [0.0]

    def timed_region(body, steps):
        """barrier + synchronize, `steps` calls of body(i), synchronize + barrier; max over ranks."""
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            body(i)
        host_enqueue[0] = (time.perf_counter() - t0) / max(steps, 1) * 1e3  # host time to ENQUEUE a step
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def batch_roofline(tops, mfma_peak, ops_per_step, read_bytes_per_step, step_s):
        """A batch step has two floors: the store's rows leave HBM once, and the pairs cost 2 * actual_dim matrix-core ops each.
        The larger floor is the bound the step is priced against; the other side's figures ride along."""
        t_hbm, t_mfma = read_bytes_per_step / (HBM_PEAK_GBPS * 1e9), ops_per_step / (mfma_peak * 1e12)
        gbps = read_bytes_per_step / step_s / 1e9
        both = {"mfma_achieved_TOPs": tops, "mfma_peak_TOPs": mfma_peak, "mfma_frac": tops / mfma_peak,
                "hbm_achieved_GBps": gbps, "hbm_frac": gbps / HBM_PEAK_GBPS, "floor_ms": {"hbm": t_hbm * 1e3, "mfma": t_mfma * 1e3}}
        if t_hbm >= t_mfma:
            return {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, **both}
        return {"bound": "mfma", "achieved": tops, "peak": mfma_peak, "unit": "TFLOP/s", "frac": tops / mfma_peak, **both}

    if args.batch_queries > 0:
        # (u8 and binary: the matrix-core multi-query kernels; pq: one table fills the LDS, a batch is the per-query pipelines enqueued
        # back to back - BASELINE config 4's PQ leg, 1024 queries per step)
        if not 1 <= args.k <= 1024:
            raise SystemExit("--batch-queries: --k must be 1 .. 1024 (the library's limit for one top-k call)")
        from quantization_amd.sharded import ShardedTopKBatch
        Q, k = args.batch_queries, args.k
        bq = torch.rand((Q, dim), generator=qgen, device=dev, dtype=torch.float32)
        batch = enc.encode_query_batch(bq)
        xchg = ShardedTopKBatch(dist, torch, Q, k, dev, rank, world, total_rows)
        if scaling_field == "weak" and world > 1:
            xchg.bases = [args.rows * r for r in range(world)]
        ids, sc = xchg.buffers()

        def bstep(_i):
            enc.topk_batch(batch, k, largest=True, out_ids=ids, out_scores=sc)
            return xchg.exchange(largest=True)

        # untimed set-up passes for about a quarter second of GPU work (the clock ramps with busy time, see the scan path below)
        # (every rank runs the SAME number of passes: a step holds a collective)
        t_pre = time.perf_counter()
        for i in range(2):
            bstep(i)
        torch.cuda.synchronize()
        n_pre = int(float(os.environ.get("QAMD_BENCH_PREWARM_S", "0.25")) / max((time.perf_counter() - t_pre) / 2, 1e-5))
        if use_dist:
            t_n = torch.tensor([n_pre], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t_n, op=dist.ReduceOp.MAX)
            n_pre = int(t_n.item())
        for i in range(min(n_pre, 2000)):
            bstep(i)
        torch.cuda.synchronize()
        for i in range(max(1, args.warmup)):
            bstep(i)
        elapsed = timed_region(bstep, args.steps)
        if rank == 0:
            if args.quantizer == "pq":
                m = bytes_per_row
                kernel_name, launches = enc.scan_kernel()
                cpu = None
                if world == 1 and not args.no_cpu_baseline:
                    try:
                        q0 = enc.encode_query(bq[0])
                        full = enc.score_all(q0, out=torch.empty(n, dtype=torch.float32, device=dev)).cpu().numpy()
                        S = min(n, args.cpu_sample_rows or n, 10_000_000)
                        cpu = cpu_baseline("pq", enc, bq[0].cpu().numpy(), full, 0 if args.distance == "dot" else 2, S, args.pq_chunk)
                        cpu["unit"] = "pairs/s"
                        cpu["sample"] = (f"ONE query of the batch (the CPU path scores a batch as {Q} such passes: pairs/s is the same "
                                         f"figure); ") + cpu["sample"]
                        enc.topk_batch(batch, k, largest=True, out_ids=ids, out_scores=sc)
                        got_ids = ids[:k].cpu().numpy().view(np.uint32)
                        got_sc = sc[:k].cpu().numpy()
                        order = np.lexsort((np.arange(n), -full))[:k]
                        cpu["batch_topk_of_that_query_matches"] = bool(np.array_equal(got_sc.view(np.uint32), full[order].view(np.uint32))
                                                                       and np.array_equal(np.sort(full[got_ids]), np.sort(full[order])))
                    except Exception as e:
                        cpu = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
                step_s = elapsed / args.steps
                gbps = float(Q) * n * m / step_s / 1e9  # every query streams the store's code bytes once
                one, source = pmc_traffic(f"pq_scan_m{m}", n, n * m)  # the single scan's profile: the same kernel, Q times
                print(json.dumps({
                    "metric": f"(query, vector) pairs scored/sec, {Q} queries x {total_rows}x{dim} pq dot, top-{k} each",
                    "value": float(Q) * total_rows * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
                    "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": step_s * 1e3,
                    "higher_is_better": True, "scaling": scaling_field, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": f"{Q} queries x {total_rows} x {dim} PQ rows (chunk {args.pq_chunk}: {m} code bytes; {n} rows on "
                                           f"rank 0), per step: topk_batch over the shard (a table fills the LDS: the per-query "
                                           f"pipelines back to back) + all-gather of world*Q*k pairs + per-query merge on the GPU",
                               "rows_per_gpu": n, "dim": dim, "queries": Q, "k": k, "total_rows": total_rows, **dist_info},
                    "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                                 "traffic": None if one is None else one * Q,
                                 "traffic_source": None if one is None else f"{Q} x the single scan's: {source}",
                                 "kernel": kernel_name, "launches_per_query": launches, "ms_per_query": step_s * 1e3 / Q,
                                 "algorithmic_read_bytes_per_step": Q * n * m,
                                 "note": "m code bytes per (query, row) pair from HBM: no two queries' tables fit the LDS together, so "
                                         "every query streams the store; achieved is over the WHOLE step (tables, sample passes, "
                                         "filter scans, merges, exchange); traffic = the committed single-scan PMC profile x queries"},
                    "cpu_baseline": cpu,
                }), flush=True)
                if use_dist:
                    dist.barrier()
                    dist.destroy_process_group()
                return
            is_bin = args.quantizer == "binary"
            ad = dim if is_bin else enc.metadata["actual_dim"]  # binary: one 0/1 operand byte per bit on the matrix cores
            ops = 2.0 * Q * n * ad  # per GPU and step
            # binary batches of 3 and of 5+ queries on rows of 4 / 6 / 8 / 12 128-bit words take the fp4 matrix-core kernels (csrc/bin.hip:
            # bin_gemm_rs4_kernel while the batch's nibble image fits in LDS, bin_gemm_qs4_kernel beyond)
            bin_fp4 = is_bin and (Q == 3 or Q >= 5) and (ad + 127) // 128 in (4, 6, 8, 12)
            mfma_peak = MFMA_FP4_PEAK_TOPS if bin_fp4 else MFMA_INT8_PEAK_TOPS
            per_gpu_tops = ops * args.steps / elapsed / 1e12
            row_bytes = nb if is_bin else bytes_per_row
            traffic, source = pmc_traffic(f"{'bin' if is_bin else 'u8'}_batch{Q}_{dim}", n, n * row_bytes)
            cpu = None
            if world == 1 and not args.no_cpu_baseline:
                # the reference has no batched form: a batch of Q queries is Q passes of its caller loop, timed here for
                # one query of the batch over a bounded sample of the store's own rows; the batch's top-k for that query
                # is checked against the top-k of the CPU's scores when the sample is the whole store
                try:
                    dist_id = 0 if args.distance == "dot" else 2
                    q0 = enc.encode_query(bq[0])
                    full = enc.score_all(q0, out=torch.empty(n, dtype=torch.float32, device=dev)).cpu().numpy()
                    S = min(n, args.cpu_sample_rows or n, 10_000_000)
                    cpu = cpu_baseline(args.quantizer, enc, bq[0].cpu().numpy(), full, dist_id, S, args.pq_chunk)
                    cpu["unit"] = "pairs/s"
                    cpu["sample"] = (f"ONE query of the batch (the CPU path scores a batch as {Q} such passes: pairs/s is "
                                     f"the same figure); ") + cpu["sample"]
                    enc.topk_batch(batch, k, largest=True, out_ids=ids, out_scores=sc)
                    got_ids = ids[:k].cpu().numpy().view(np.uint32)
                    got_sc = sc[:k].cpu().numpy()
                    order = np.lexsort((np.arange(n), -full))[:k]
                    cpu["batch_topk_of_that_query_matches"] = bool(np.array_equal(got_sc.view(np.uint32), full[order].view(np.uint32))
                                                                   and np.array_equal(np.sort(full[got_ids]), np.sort(full[order])))
                except Exception as e:
                    cpu = {"value": None, "unit": "pairs/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
            print(json.dumps({
                "metric": f"(query, vector) pairs scored/sec, {Q} queries x {total_rows}x{dim} {args.quantizer} dot, top-{k} each",
                "value": float(Q) * total_rows * args.steps / elapsed, "unit": "pairs/s", "n_gpus": world,
                "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": scaling_field, "vs_baseline": None,
                "dtype": ("1-bit x 1-bit -> exact count (bits as E2M1 0/1 nibbles, MFMA fp4 -> f32)" if bin_fp4 else
                          "1-bit x 1-bit -> i32 (bits expanded to 0/1 bytes, MFMA int8)") if is_bin else "u8 x u8 -> i32 (MFMA int8)",
                "data": "synthetic",
                "config": {"workload": f"{Q} queries x {total_rows} x {dim} {'binary' if is_bin else 'scalar-u8'} rows ({n} on rank 0), per step: "
                                       f"topk_batch over the shard + all-gather of world*Q*k pairs + per-query merge "
                                       f"on the GPU",
                           "rows_per_gpu": n, "dim": dim, "queries": Q, "k": k, "total_rows": total_rows, **dist_info},
                "roofline": batch_roofline(per_gpu_tops, mfma_peak, ops, n * row_bytes, elapsed / args.steps) | {
                             "traffic": traffic, "traffic_source": source,
                             "algorithmic_read_bytes_per_step": n * row_bytes,
                             "note": ("fp4" if bin_fp4 else "int8") + " op/s per GPU over the whole step (sample pass, filter GEMM, scatter, "
                                     "sort, exchange); algorithmic ops = 2 * actual_dim per (query, row) pair; traffic = HBM bytes "
                                     "of the step's dominant (filter) kernel: the store's rows leave HBM once per batch"},
                "cpu_baseline": cpu,
            }), flush=True)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    multi = world > 1 or args.force_dist
    group = args.gather_group or (4 if multi else 1)
    encode_ahead = args.encode_ahead == 1  # measured at 1.25M rows/GPU: 0.151 ms per step against 0.147 in line
    # (the side stream's kernel and the two cross-stream waits cost more than the 4 us encode they hide)
    time_every = args.time_every or (8 if multi else 1)
    qobjs = [enc.encode_query(queries[0]), enc.encode_query(queries[0])]
    gather = topk = None
    if args.exchange == "scores":
        gather = ScoreGather(dist, torch, n_max, dev, rank, world,
                             dst=None if args.gather_root == "rotate" else int(args.gather_root),
                             always_collective=args.force_dist, group_steps=group)
    elif args.exchange == "topk":
        topk = ShardedTopK(dist, torch, args.k, dev, rank, world, total_rows)
        if scaling_field == "weak" and world > 1:
            topk.bases = [args.rows * r for r in range(world)]
    scores_plain = torch.empty(max(n_max, 1), dtype=torch.float32, device=dev) if gather is None else None

    # The hot step goes to the C ABI directly with pre-bound arguments: at 8 GPUs a shard scan is
    # ~0.15 ms, so the tens of microseconds the Python mirror spends per call (buffer checks,
    # torch stream lookup) would show up in the strong-scaling number.
    pfx = {"u8": "u8", "binary": "bin", "pq": "pq"}[args.quantizer]
    f_encode = getattr(L, f"qamd_{pfx}_encode_query")
    f_score = getattr(L, f"qamd_{pfx}_score_all")
    h_store = enc._h
    h_query = [C.c_void_p(q._h.value) for q in qobjs]
    main_stream = torch.cuda.current_stream()
    stream = C.c_void_p(main_stream.cuda_stream)
    side = torch.cuda.Stream() if encode_ahead else None  # non-blocking side stream for the next query's encode
    side_ptr = C.c_void_p(side.cuda_stream) if side is not None else None
    scan_done = [torch.cuda.Event(), torch.cuda.Event()]  # the scan that last READ query object j
    q_ptrs = [C.c_void_p(queries[i].data_ptr()) for i in range(args.queries)]
    qdim = int(dim)
    ev_pairs = []

    def encode(i, on):
        """encode_query of step i into query object i % 2 (the library orders a consumer on another
        stream after it: ReadyEvent in csrc/common.hpp)."""
        return f_encode(h_store, q_ptrs[i % args.queries], qdim, _lib.MEM_DEVICE, on, C.byref(h_query[i % 2]))

    def step(i, timed=True, first=False, last=False):
        st = 0
        if encode_ahead:
            if first:
                st |= encode(i, side_ptr)
            if not last:  # step i+1's query: its object was last read by the scan of step i-1
                side.wait_event(scan_done[(i + 1) % 2])
                st |= encode(i + 1, side_ptr)
        else:
            st |= encode(i, stream)
        out = gather.slot(i) if gather is not None else scores_plain
        timed = timed and (i % time_every == 0)
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        st |= f_score(h_store, h_query[i % 2], C.c_void_p(out.data_ptr()), _lib.MEM_DEVICE, stream)
        if timed:
            e1.record()
            ev_pairs.append((e0, e1))
        if encode_ahead:
            scan_done[i % 2].record(main_stream)
        if st:
            raise RuntimeError(L.qamd_last_error().decode())
        if gather is not None:
            gather.submit(i)
        elif topk is not None:
            ids, sc = topk.buffers()  # per-shard selection over the fresh scores, then k pairs/rank
            qa.topk_scores(out, n, args.k, largest=True, out_ids=ids, out_scores=sc)
            topk.exchange(largest=True)

    # set-up, like the encode above: untimed passes so that the W warm-up steps and the K timed steps start from a ramped
    # clock and warm allocator pools.  The ramp takes TIME, not launches: a 0.2 ms PQ scan was still getting faster 40
    # passes in (0.228 -> 0.195 ms per launch, profiles/r04_pq.txt), so the passes go on for about a quarter of a second
    # of GPU work whatever a pass costs; they use the plain score buffer and take part in no exchange
    prewarm = torch.empty(max(n_max, 1), dtype=torch.float32, device=dev)

    def prewarm_passes(count):
        t0 = time.perf_counter()
        for i in range(count):
            encode(i, stream)
            f_score(h_store, h_query[i % 2], C.c_void_p(prewarm.data_ptr()), _lib.MEM_DEVICE, stream)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    per_pass = prewarm_passes(20) / 20
    prewarm_passes(min(5000, int(float(os.environ.get("QAMD_BENCH_PREWARM_S", "0.25")) / max(per_pass, 1e-6))))
    del prewarm
    for i in range(args.warmup):
        step(i, False, first=(i == 0), last=(i == args.warmup - 1))
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()

    def body(i):
        step(i, True, first=(i == 0), last=(i == args.steps - 1))
        if i == args.steps - 1 and gather is not None:
            gather.drain()

    elapsed = timed_region(body, args.steps)
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
                    "kernel_ms_min": float(np.min(kern_all)), "kernel_ms_median": float(np.median(kern_all)),
                    "kernel_ms_mean": kern_ms,
                    "algorithmic_bytes_per_row": bytes_per_row, "rows_per_launch": n,
                    # `achieved` counts what SURVEY 8(d) counts: the row bytes READ.  The scan also writes one f32 score per
                    # row; with it the kernel moves this much (3 % more for 128-byte binary rows, 0.5 % for u8 at dim 768):
                    "score_bytes_written_per_row": 4,
                    "achieved_incl_score_writes": (bytes_per_row + 4) * n / (kern_ms * 1e-3) / 1e9}
        if args.quantizer == "pq":
            # The PQ scan reads only m bytes per row from HBM; the LDS gather rate rides along.
            lds = 4.0 * bytes_per_row * n / (kern_ms * 1e-3) / 1e9
            if pq_skew:
                # pq_scan_skew_kernel (m = 32 / 64 / 96 / 128): transposed LUT + quads skewed in time, no bank conflicts
                # (SQ_LDS_BANK_CONFLICT = 0): the nominal bound, HBM reads of m bytes per row, is the roofline again
                roofline.update({"lds_gather_GBps": lds, "lds_gather_frac_of_conflict_free_peak": lds / LDS_B32_PEAK_GBPS,
                                 "note": "m code bytes per row from HBM; per chunk and row one ds_read_u8 (code), one "
                                         "ds_read_b32 (table entry) and two vector-ALU operations, all conflict-free.  The "
                                         "kernel runs 3 % above its own stream (loads, LDS ring, 4-byte score stores); the "
                                         "stores - 4 % of the bytes - cost a quarter of that stream's time "
                                         "(profiles/r04_pq_skew_probe.txt)"})
            else:
                roofline.update({"bound": "lds", "achieved": lds, "peak": LDS_B32_PEAK_GBPS, "frac": lds / LDS_B32_PEAK_GBPS,
                                 "hbm_achieved_GBps": achieved, "hbm_frac": achieved / HBM_PEAK_GBPS,
                                 "note": "achieved = 4 B x m LUT gathers per row (ds_read_b32) against the conflict-free "
                                         "LDS rate of all CUs; random 8-bit codes give ~3.5-way bank conflicts, i.e. a "
                                         "practical ceiling near 0.29 of that peak"})
        pmc_key = {"u8": "u8_scan", "binary": "bin_scan", "pq": f"pq_scan_m{bytes_per_row}"}[args.quantizer]
        traffic, source = pmc_traffic(pmc_key, n, n * bytes_per_row)
        if traffic is not None:
            roofline["traffic"] = traffic
            roofline["traffic_source"] = source
        # on-box streaming-read ceiling with the same load shape (16 B/lane, nt): a measurement helper built
        # beside the tools (tools/probe), not an entry point of the product library
        try:
            P = C.CDLL(_lib.PROBE_PATH)
            P.qamd_probe_stream_read.restype = C.c_int
            P.qamd_probe_stream_read.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
            probe = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
            probe.zero_()
            scratch = torch.empty(1 << 16, dtype=torch.uint8, device=dev)
            s = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                if P.qamd_probe_stream_read(probe.data_ptr(), probe.numel(), scratch.data_ptr(), s):
                    raise RuntimeError("probe launch failed")
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                P.qamd_probe_stream_read(probe.data_ptr(), probe.numel(), scratch.data_ptr(), s)
            b.record()
            torch.cuda.synchronize()
            roofline["stream_read_ceiling_GBps"] = probe.numel() * 10 / (a.elapsed_time(b) * 1e-3) / 1e9
            del probe
        except Exception as e:  # pragma: no cover
            roofline["stream_read_ceiling_GBps"] = None
            roofline["stream_read_error"] = str(e)

        headline = (total_rows == 10_000_000 and dim == 768 and args.distance == "dot" and args.quantizer == "u8")
        shard_txt = (f"{total_rows} x {dim} store" if world == 1 else
                     f"{total_rows} x {dim} store row-sharded over {world} GPUs ({n} rows on rank 0, {scaling_field} scaling)")
        result = {
            "metric": "scored vectors/sec, 10Mx768 u8 dot" if headline
            else f"scored vectors/sec, {total_rows}x{dim} {args.quantizer} {args.distance}",
            "value": value, "unit": "vectors/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "host_enqueue_ms_per_step": host_enqueue[0],
            "higher_is_better": True, "scaling": scaling_field, "vs_baseline": None,
            "dtype": {"u8": "u8", "binary": "u1 (xor+popcount, i32)", "pq": "f32 (LUT adds)"}[args.quantizer],
            "data": "synthetic",
            "config": {"workload": (f"{shard_txt}: f32 U[0,1) -> scalar u8 ({args.distance}); "
                                    if args.quantizer == "u8" else
                                    f"{shard_txt}: {args.quantizer} rows ({bytes_per_row} B/row, {args.distance}); ") +
                                   f"per step: encode_query + score_all over the shard"
                                   + (f" + {args.exchange} exchange" if world > 1 and args.exchange != "none" else ""),
                       "quantizer": args.quantizer, "rows_per_gpu": n, "dim": dim, "distance": args.distance,
                       "exchange": args.exchange, "exchange_reason": exchange_reason,
                       "gather_group": group if gather is not None else None, **dist_info,
                       "encode_ahead": encode_ahead, "kernel_timed_every": time_every,
                       "gather_root": args.gather_root if (world > 1 and args.exchange == "scores") else None,
                       "total_rows": total_rows, "queries": args.queries},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                torch.cuda.synchronize()
                enc.encode_query(queries[0], reuse=qobjs[0])
                full = enc.score_all(qobjs[0], out=torch.empty(n, dtype=torch.float32, device=dev))
                torch.cuda.synchronize()
                sample_rows = min(n, args.cpu_sample_rows or n, 10_000_000)
                result["cpu_baseline"] = cpu_baseline(args.quantizer, enc, queries[0].cpu().numpy(), full.cpu().numpy(),
                                                      0 if args.distance == "dot" else 2, sample_rows, args.pq_chunk)
            except Exception as e:
                result["cpu_baseline"] = {"value": None, "unit": "vectors/s", "cores": 0, "kind": "port",
                                          "sample": f"failed: {e}"}
        print(json.dumps(result), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
