"""In-kernel timeline of u8_gemm_kernel (developer tool): phase durations in shader cycles.

Usage: python tools/gemm_timeline.py [NQ] [N] [DIM]
"""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quantization_amd as qa
from quantization_amd import _lib

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 768
dev = torch.device("cuda", 0)
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
batch = enc.encode_query_batch(torch.rand((nq, dim), device=dev))
ids = torch.empty(nq * 30, dtype=torch.int32, device=dev)
sc = torch.empty(nq * 30, dtype=torch.float32, device=dev)
L = _lib.lib()
L.qamd_dev_gemm_stamps.argtypes = [C.c_void_p]
WAVES = int(os.environ.get("WAVES", 8))
for _ in range(2):
    enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc)
torch.cuda.synchronize()
if os.environ.get("DBG"):  # ablations need a library built with `make EXTRA=-DQAMD_GEMM_ABLATION`
    L.qamd_dev_gemm_debug.argtypes = [C.c_uint]
    L.qamd_dev_gemm_debug(int(os.environ["DBG"]))
stamps = torch.zeros(4096 * WAVES * 16, dtype=torch.int64, device=dev)
assert L.qamd_dev_gemm_stamps(C.c_void_p(stamps.data_ptr())) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
enc.topk_batch(batch, 30, out_ids=ids, out_scores=sc)
e1.record()
torch.cuda.synchronize()
L.qamd_dev_gemm_stamps(None)
print(f"topk_batch with stamps: {e0.elapsed_time(e1):.2f} ms")
st = stamps.cpu().numpy().reshape(4096, WAVES, 16)
if os.environ.get("QAMD_GEMM_CFG", "p")[0] in "qgs":  # g: the queries-in-registers form (BLK_ROWS=64, CHUNK_Q=32); s: resident queries
    # query-streaming kernel: per wave, cycles summed over its row blocks
    n_wg, blk_rows = 256, int(os.environ.get("BLK_ROWS", 128))  # BLK_ROWS=96: rows of 1153-1536 bytes
    blk = st[:n_wg]
    ok = (blk[:, :, 15] > 0).all(axis=1)
    blk = blk[ok].astype(np.float64)
    tot = blk[:, :, :6].sum(axis=2).mean()
    names = ["barrier 1 (waiting for the other waves)", "row pieces landing + LDS writes + barrier 2", "accumulator set-up",
             "K loop", "epilogue", "row request + loop overhead"]
    print("workgroups with stamps:", blk.shape[0], " cycles per wave (mean):", int(tot))
    for i, nm in enumerate(names):
        x = blk[:, :, i]
        print(f"  {nm:46s} {x.mean():12.0f}  {100 * x.mean() / tot:5.1f} %   (p10 {np.percentile(x, 10):10.0f}, p90 {np.percentile(x, 90):10.0f})")
    n_blocks = (n + blk_rows - 1) // blk_rows / n_wg
    chunk_q = int(os.environ.get("CHUNK_Q", 64))
    mf = n_blocks * ((nq + chunk_q - 1) // chunk_q) / WAVES * ((enc.metadata["actual_dim"] + 127) // 128) * (blk_rows // 4) * chunk_q / 64
    print(f"  MFMAs per wave {mf:.0f}: K loop cycles per MFMA {blk[:, :, 3].mean() / mf:.1f} (two waves share a SIMD: 64 nominal)")
    if blk[:, :, 7].mean() > 0:
        print(f"  kernel {blk[:, :, 7].mean() / 100:.1f} us per wave (10 ns ticks), shader clock while it ran "
              f"{blk[:, :, 6].mean() / blk[:, :, 7].mean() * 100:.0f} MHz")
    for i in (0, 3, 4):
        print("  per wave index,", names[i][:24], ":", " ".join(f"{v/1e3:8.0f}k" for v in blk[:, :, i].mean(axis=0)))
    sys.exit(0)
elif os.environ.get("QAMD_GEMM_CFG", "p")[0] == "p":
    # ping-pong kernel: persistent workgroups; slot 0 entry, 1 prologue done, 2..9 MFMAs of tile
    # 0..7 issued, 10 end of the last stamped epilogue, 12 exit
    blk = st[:256]
    ok = (blk[:, :, 15] > 0).all(axis=1)
    blk = blk[ok]
    print("workgroups with stamps:", blk.shape[0])
    def dd(a, b):
        x = (blk[:, :, b] - blk[:, :, a]).astype(np.float64)
        return f"{x.mean():9.0f} (p10 {np.percentile(x, 10):7.0f}, p90 {np.percentile(x, 90):7.0f})"
    n_slabs = (enc.metadata["actual_dim"] + 63) // 64
    print(f"K-tiles per tile: {n_slabs}; ideal MFMA cycles per tile: {n_slabs * 1024}")
    print("entry -> prologue done   ", dd(0, 1))
    prev = 1
    for tix in range(4):
        print(f"tile {tix}: set-up          ", dd(prev, 2 + 3 * tix))
        print(f"tile {tix}: K loop          ", dd(2 + 3 * tix, 3 + 3 * tix))
        if tix < 3:  # slot 13 holds HW_ID
            print(f"tile {tix}: epilogue        ", dd(3 + 3 * tix, 4 + 3 * tix))
        prev = 4 + 3 * tix
    print("whole workgroup          ", dd(0, 15))
    hw = blk[:, :, 13]
    simd = (hw >> 4) & 3
    print("SIMD of waves 0..7, first workgroups:", [list(map(int, simd[i])) for i in range(4)])
    same = sum(int(simd[i, w] == simd[i, w + 4]) for i in range(blk.shape[0]) for w in range(4))
    print(f"wave w and w+4 on the same SIMD: {same} of {blk.shape[0] * 4}")
    flagged = float(blk[:, :, 14].sum())
    groups = nq / 4.0 * n  # (query group, row) pairs of the filter pass
    print(f"groups sent to the exact epilogue: {flagged:.0f} of {groups:.3g} = {flagged / groups:.2e}")
    sys.exit(0)
# the LAST launch that wrote stamps is the filter pass over the full store (the sample pass ran
# first and was overwritten for the blocks both have).  Use blocks 1024.. (steady state).
blk = st[1024:4096]
ok = blk[:, :, 12] > 0
print("blocks with stamps:", int(ok.all(axis=1).sum()))
def d(a, b):
    x = (blk[:, :, b] - blk[:, :, a])[ok]
    return f"{x.mean():9.0f} (p10 {np.percentile(x, 10):7.0f}, p90 {np.percentile(x, 90):7.0f})"
print("entry -> first slab staged   ", d(0, 1))
n_slabs = (enc.metadata["actual_dim"] + 127) // 128
prev = 1
for s in range(min(n_slabs, 8)):
    print(f"slab {s}: MFMAs issued         ", d(prev, 2 + s))
    prev = 2 + s
print("last slab -> loop end        ", d(prev, 10))
print("loop end -> epilogue consts  ", d(10, 11))
print("epilogue                     ", d(11, 12))
print("whole workgroup              ", d(0, 12))
# workgroup-level: how long is a CU busy with one tile, and start-to-start spacing per CU unknown;
# report the spread of wave end times inside a workgroup
end = blk[:, :, 12]
print("wave end skew inside a workgroup (max-min):", float((end.max(axis=1) - end.min(axis=1)).mean()))
