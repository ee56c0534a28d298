"""bench.py end to end on the GPU box: the single-GPU line (all three quantizers) and the
multi-rank code path rehearsed with two ranks on ONE GPU over gloo (RCCL needs one GPU per rank;
the driver runs the real N=2,4,8 on an 8-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("extra", [[], ["--quantizer", "binary", "--dim", "1024"], ["--quantizer", "pq"],
                                   ["--quantizer", "pq", "--pq-chunk", "4"], ["--distance", "l2"]])
def test_bench_single_gpu_line(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--rows-per-gpu", "300000",
           "--cpu-sample-rows", "50000"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    j = _last_json(res.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 5 and j["value"] > 0
    rf = j["roofline"]
    # PQ: m = 96 takes the conflict-free scan; m = 192 (--pq-chunk 4) the same kernel per LUT slice of the planar scan
    # image (round 3: pq_scan_fast_kernel, bound by LDS bank conflicts): HBM is the roofline of both
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1.2
    assert rf["kernel_ms_min"] <= rf["kernel_ms_median"] and rf["kernel_ms_mean"] == rf["kernel_ms"]
    if "pq" in extra:
        assert rf["kernel"] == ("pq_scan_skew_kernel<SLICED>" if "--pq-chunk" in extra else "pq_scan_skew_kernel")
        assert 0 < rf["lds_gather_frac_of_conflict_free_peak"] < 1.0
    # every quantizer's line carries the CPU baseline: the reference's compiled kernels for u8 and binary, the oracle's
    # restatement of score_point_sse for PQ; all GPU scores of the sample equal the CPU's bit for bit
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] in ("reference", "port")
    assert cb["gpu_matches_cpu_bits"] is True
    assert cb["all_cores"]["value"] > 0
    if "pq" in extra:
        assert cb["kind"] == "port"


def test_bench_rccl_code_path_on_one_gpu():
    """torch.distributed over nccl (= RCCL) initialised at world size 1: process group, barrier,
    all_reduce and a one-rank asynchronous score gather all go through RCCL on the single GPU."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--rows", "300000",
           "--no-cpu-baseline", "--force-dist", "--backend", "nccl"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    assert j["n_gpus"] == 1 and j["value"] > 0


@pytest.mark.parametrize("exchange,scaling", [("scores", "strong"), ("topk", "strong"), ("scores", "weak")])
def test_bench_two_ranks_one_gpu_gloo_rehearsal(exchange, scaling):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--rows", "200001", "--backend", "gloo", "--all-ranks-on-device", "0",
           "--exchange", exchange] + (["--scaling", "weak"] if scaling == "weak" else [])
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    # the default for N > 1 is STRONG scaling: the store stays at --rows and is row-sharded
    assert j["n_gpus"] == 2 and j["scaling"] == scaling
    assert j["config"]["total_rows"] == (200001 if scaling == "strong" else 400002)
    assert j["config"]["rows_per_gpu"] == (100000 if scaling == "strong" else 200001)
    assert "cpu_baseline" not in j


def test_bench_four_ranks_strong_scaling_grouped_rotating_gather():
    """Four ranks on one GPU over gloo: the strong-scaling default (store fixed, 4 ragged shards), gathers in
    groups of 4 queries with a rotating root, a step count that leaves a partly filled last group."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
           "127.0.0.1", "--master-port", "29535", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "10",
           "--warmup", "3", "--rows", "400003", "--backend", "gloo", "--all-ranks-on-device", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    assert j["n_gpus"] == 4 and j["scaling"] == "strong" and j["config"]["total_rows"] == 400003
    assert j["config"]["rows_per_gpu"] == 100000 and j["config"]["gather_group"] == 4
    assert j["value"] > 0 and j["roofline"]["rows_per_launch"] == 100000


@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_batched_topk_mode(ranks):
    """The opt-in config-4 shape: many queries per step on the matrix cores, sharded rows, one
    all-gather of world*Q*k pairs (two ranks rehearsed on one GPU over gloo)."""
    base = [os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--rows-per-gpu",
            "1200000", "--dim", "192", "--batch-queries", "200", "--k", "10"]
    if ranks == 1:
        cmd = [sys.executable] + base
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29534"] + base + ["--backend", "gloo", "--all-ranks-on-device", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    assert j["n_gpus"] == ranks and j["unit"] == "pairs/s" and j["value"] > 0
    rf = j["roofline"]  # priced against the larger of its two floors (rows out of HBM once; 2 * dim matrix-core ops per pair)
    assert rf["bound"] == ("hbm" if rf["floor_ms"]["hbm"] >= rf["floor_ms"]["mfma"] else "mfma") and 0 < rf["frac"] < 1
    assert 0 < rf["mfma_frac"] < 1 and 0 < rf["hbm_frac"] < 1
    assert j["config"]["total_rows"] == 1200000  # strong scaling: the store is fixed, the ranks split it
    if ranks == 1:  # one query of the batch through the reference's per-query loop on the CPU; its top-k checked
        cb = j["cpu_baseline"]
        assert cb["value"] > 0 and cb["unit"] == "pairs/s" and cb["gpu_matches_cpu_bits"] is True
        assert cb["batch_topk_of_that_query_matches"] is True


def test_bench_pq_batched_mode():
    """--quantizer pq --batch-queries: BASELINE config 4's PQ leg in small - the per-query pipelines back to back, priced against the code
    bytes every query streams; one query of the batch checked against the oracle's caller loop."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--quantizer", "pq", "--steps", "1", "--warmup", "1", "--rows", "300000", "--dim",
           "192", "--batch-queries", "8", "--k", "10"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, QAMD_BENCH_PREWARM_S="0.02"))
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    assert j["unit"] == "pairs/s" and j["value"] > 0 and j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    assert j["roofline"]["algorithmic_read_bytes_per_step"] == 8 * 300000 * 24
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["gpu_matches_cpu_bits"] is True and cb["batch_topk_of_that_query_matches"] is True


def test_bench_starts_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no rank environment: the parent (which never touches the GPU) starts the
    two ranks as a child torch.distributed.run, relays rank 0's line and exits with the child's code.  The line
    says what the process group was: backend, the world size torch.distributed saw, one device entry per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--rows",
           "200001", "--backend", "gloo", "--all-ranks-on-device", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    cfg = j["config"]
    assert cfg["dist_backend"] == "gloo" and cfg["world_size_seen"] == 2 and len(cfg["rank_devices"]) == 2
    assert cfg["exchange"] == "scores" and "auto" in cfg["exchange_reason"]
    # a failing rank must fail the launcher too
    bad = subprocess.run(cmd + ["--batch-queries", "8", "--k", "5000"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert bad.returncode != 0


def test_bench_binary_multi_rank_defaults_to_the_topk_exchange():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rows",
           "400000", "--quantizer", "binary", "--dim", "1024", "--backend", "gloo", "--all-ranks-on-device", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    assert j["config"]["exchange"] == "topk" and "score gather" in j["config"]["exchange_reason"]


@pytest.mark.parametrize("exchange", ["scores", "topk"])
def test_bench_single_process_sharded_handle(exchange):
    """--single-process: one process drives the shards through qamd_u8_sharded_* (two logical shards here)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--single-process", "--gpus", "2", "--devices", "0,0",
           "--steps", "5", "--warmup", "2", "--rows", "300001", "--exchange", exchange]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    j = _last_json(res.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert j["config"]["launch"] == "single-process" and j["config"]["devices"] == [0, 0]
    assert j["config"]["rows_per_gpu"] == 150000 and j["config"]["exchange"] == exchange
    assert 0 < j["roofline"]["frac"] < 1.2
