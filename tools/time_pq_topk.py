"""PQ single-query top-k and score_all, ms per call with device outputs: config 2's shape (10M x 768, m = 96), config 4's PQ
leg (12.5M x 1536, m = 192: two LUT slices of the planar scan image), m = 48 (dim 768 at chunk 16: two rows per ring row),
m = 16 (two rows per 32-chunk ring row), m = 80 / 112 (ring rows padded to 96 / 128 chunks with zero table columns) the reference bench's m = 512, and m = 120 / 100 (dim 960 / 800 at chunk 8: rows padded to 128 / 112 bytes, zero table columns past m).  With
QAMD_LIB_PATH=tools/lib/libquantization_amd_dev.so, QAMD_PQ_SKEW=0 selects the older scan kernel for comparison."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:
    print(__doc__)
    _sys.exit(0)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quantization_amd as qa
D = qa.DistanceType
dev = torch.device("cuda", 0)
shapes = ((10_000_000, 768, 8), (12_500_000, 1536, 8), (20_000_000, 768, 16), (12_000_000, 640, 8), (8_000_000, 896, 8),
          (20_000_000, 128, 8), (2_000_000, 1024, 2), (8_000_000, 960, 8), (10_000_000, 800, 8))
only = [int(a) for a in _sys.argv[1:] if a.isdigit()]  # optional: the m values to run
for n, dim, chunk in shapes:
    if only and dim // chunk not in only:
        continue
    m = dim // chunk
    rows = torch.randint(0, 256, (n, m), device=dev, dtype=torch.uint8)
    cen = np.random.default_rng(0).random((256, dim), dtype=np.float32)
    enc = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(dim, n, D.Dot, False), chunk, cen)
    del rows
    q = enc.encode_query(torch.rand(dim, device=dev))
    ids = torch.empty(30, dtype=torch.int32, device=dev)
    sc = torch.empty(30, dtype=torch.float32, device=dev)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    for name, fn in (("topk(30)", lambda: enc.topk(q, 30, out_ids=ids, out_scores=sc)), ("score_all", lambda: enc.score_all(q, out=out))):
        t_warm = time.perf_counter()  # the clock ramps with busy TIME, not launches: a quarter of a second of calls first
        while time.perf_counter() - t_warm < 0.25:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        print(f"{n} x {dim} m={m} {enc.scan_kernel()} {name}: {ms:.4f} ms per call = {n * m / ms / 1e9:.2f} TB/s of code bytes = "
              f"{n * m / ms / 1e9 / 8:.3f} of 8 TB/s", flush=True)
    del enc, out
