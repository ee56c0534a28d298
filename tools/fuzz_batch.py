#!/usr/bin/env python3
"""Developer fuzz (GPU box): random shapes through score_batch / topk_batch against the single-query
path (itself pinned to the oracle by tests/test_gpu_u8.py).  Exercises the kernel selection
boundaries of csrc/u8_batch.hip (query counts around 4/5, 32, 64, 128, 256/257, 384/385, 703/704, 959/960, 2048; row lengths
around 128, 1152, 1536, 2304, 4608; stores around the 32768-row fused threshold and ragged tails).
    python tools/fuzz_batch.py [cases] [seed]      (QAMD_GEMM_CFG=r|q|p forces one kernel)"""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import quantization_amd as qa  # noqa: E402

D = qa.DistanceType
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
NQ = [2, 4, 5, 31, 32, 33, 64, 65, 127, 128, 129, 200, 256, 257, 300, 384, 385, 511, 512, 640, 703, 704, 705, 768, 959, 960, 1000, 1024, 1100, 2047, 2048,
      2049, 2500]
DIMS = [16, 17, 64, 100, 128, 129, 144, 256, 300, 384, 512, 640, 700, 768, 896, 1000, 1024, 1040, 1152, 1153, 1168, 1300, 1536, 1537, 1552, 2000, 2304,
        2320, 4608, 4700]
t0 = time.time()
bad = 0
for case in range(cases):
    nq = int(rng.choice(NQ))
    dim = int(rng.choice(DIMS))
    budget = 6e8  # elements of f32 data per case
    n_max = int(min(400_000, budget / dim))
    n = int(rng.choice([rng.integers(100, 5000), rng.integers(32_000, 34_000), rng.integers(34_000, max(34_001, n_max))]))
    if nq * n > 6e8:
        n = max(300, int(6e8 // nq))
    dist, invert = [(D.Dot, False), (D.L2, False), (D.Dot, True), (D.L2, True)][int(rng.integers(0, 4))]
    largest = bool(rng.integers(0, 2))
    k = int(rng.choice([1, 10, 30, 64, 100, 300]))
    g = torch.Generator(device="cuda")
    g.manual_seed(int(rng.integers(0, 1 << 30)))
    data = torch.rand((n, dim), generator=g, device=dev) - (0.3 if rng.integers(0, 2) else 0.0)
    queries = torch.rand((nq, dim), generator=g, device=dev) - (0.5 if rng.integers(0, 2) else 0.0)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    del data
    batch = enc.encode_query_batch(queries)
    ids, sc = enc.topk_batch(batch, k, largest=largest)
    picks = sorted(set(int(x) for x in rng.integers(0, nq, 6)) | {0, nq - 1})
    scores = enc.score_batch(batch) if nq * n <= 2e8 else None
    qh = queries.cpu().numpy()
    ok = True
    qobj = None
    for qi in picks:
        qobj = enc.encode_query(qh[qi], reuse=qobj)
        wi, ws = enc.topk(qobj, k, largest=largest)
        if not (np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32))):
            ok = False
            print(f"  MISMATCH topk query {qi}")
        if scores is not None:
            one = enc.score_all(qobj)
            if not np.array_equal(scores[qi].view(np.uint32), one.view(np.uint32)):
                ok = False
                print(f"  MISMATCH scores query {qi}: {int((scores[qi].view(np.uint32) != one.view(np.uint32)).sum())} of {n}")
    bad += not ok
    print(f"case {case:3d}: n={n:7d} dim={dim:5d} nq={nq:5d} k={k:4d} {dist} invert={invert} largest={largest}  {'ok' if ok else 'FAILED'}",
          flush=True)
    del enc, batch, scores
print(f"{cases} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
