"""The C-ABI boundary driven from plain C99 (tests/c_abi/c_abi_smoke.c): the program is compiled
with gcc -std=c99 -pedantic against include/quantization_amd.h and linked against the product .so.
Without a GPU we only prove it links and starts; on the GPU box it runs encode (one-shot and
streaming) -> encode_query -> score_all -> topk -> the sharded forms, and everything it prints is
compared with what the ORACLE computes from the same LCG inputs."""
import os
import subprocess

import numpy as np
import pytest

from quantization_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.c")
COUNT, DIM, K = 3000, 72, 30


def build_program(tmp_path, src=SRC, std="c99", extra=()) -> str:
    exe = str(tmp_path / os.path.splitext(os.path.basename(src))[0])
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", f"-std={std}", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    src, "-L", libdir, "-lquantization_amd", f"-Wl,-rpath,{libdir}", *extra, "-o", exe], check=True)
    return exe


def lcg_inputs():
    state = np.uint32(12345)
    vals = np.empty(COUNT * DIM + DIM, dtype=np.float32)
    s = int(state)
    for i in range(vals.size):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[i] = np.float32(((s >> 8) & 0xFFFF) / 65536.0)
    return vals[: COUNT * DIM].reshape(COUNT, DIM), vals[COUNT * DIM:]


def fold31(values) -> int:
    acc = 0
    for v in values:
        acc = (acc * 31 + int(v)) & 0xFFFFFFFF
    return acc


def test_c_program_compiles_and_links(tmp_path):
    exe = build_program(tmp_path)
    out = subprocess.run([exe, "--link-only"], capture_output=True, text=True, check=True).stdout
    assert out.startswith("version quantization_amd")


@pytest.mark.gpu
def test_c_program_matches_oracle(tmp_path, qo):
    exe = build_program(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    assert lines[-1] == "OK"
    data, query = lcg_inputs()
    rows, meta = qo.u8_encode(data, qo.DOT, False)
    codes, qoff = qo.u8_encode_query(meta, query)
    want = qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2)
    got = {ln.split()[0]: ln.split()[1:] for ln in lines if ln.split()[0] in ("meta", "rows", "scores")}
    f32bits = lambda x: int(np.float32(x).view(np.uint32))
    assert [int(x, 16) for x in got["meta"][:3]] == [f32bits(meta.alpha), f32bits(meta.offset), f32bits(meta.multiplier)]
    assert int(got["meta"][3]) == meta.actual_dim == 80
    assert int(got["rows"][0], 16) == fold31(rows.reshape(-1))
    assert int(got["scores"][0], 16) == fold31(want.view(np.uint32))
    top = [(int(ln.split()[1]), int(ln.split()[2], 16)) for ln in lines if ln.startswith("top ")]
    order = np.lexsort((np.arange(COUNT), -want.astype(np.float64)))[:K]
    assert [t[0] for t in top] == order.tolist()
    assert [t[1] for t in top] == want[order].view(np.uint32).tolist()


THREADS_SRC = os.path.join(ROOT, "tests", "c_abi", "sharded_threads.c")


def test_threaded_c_program_compiles_and_links(tmp_path):
    build_program(tmp_path, THREADS_SRC, std="gnu99", extra=("-lpthread",))


@pytest.mark.gpu
def test_sharded_handle_lets_search_threads_overlap(tmp_path):
    """Six pthreads searching one 4-shard handle (tests/c_abi/sharded_threads.c): every answer equals
    the plain handle's, and the calls INTERLEAVE -- there is no per-handle lock: the wall time of the
    concurrent run is under half the serial one (logical shards on one GPU, latency-bound searches)."""
    exe = build_program(tmp_path, THREADS_SRC, std="gnu99", extra=("-lpthread",))
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    f = res.stdout.split()
    got = {f[i]: float(f[i + 1]) for i in range(0, len(f) - 1, 2)}
    assert got["mismatches"] == 0 and got["failures"] == 0, res.stdout
    print(res.stdout)
    assert got["concurrent_us"] < 0.5 * got["serial_us"], res.stdout
