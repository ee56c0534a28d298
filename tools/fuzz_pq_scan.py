#!/usr/bin/env python3
"""Developer fuzz (GPU box): random PQ shapes through the whole-store scan (pq_scan_skew_kernel for m % 32 == 0, whole rows and
rows of several LUT slices; pq_scan_fast_kernel otherwise) against the id-list kernel on every row - a different kernel, the same
sums in the same order, so the bits must agree - and the top-k against the sorted scores.    python tools/fuzz_pq_scan.py [cases] [seed]"""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import quantization_amd as qa  # noqa: E402

D = qa.DistanceType
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
bad = 0
for case in range(cases):
    m = int(rng.choice([32, 64, 96, 128, 160, 192, 224, 256, 288, 384, 512, 16, 48, 80, 112, 100, 144, 176, 20, 36, 52, 72, 88, 120, 124, 65, 33]))
    chunk = int(rng.choice([1, 2, 4, 8]))
    # (from ~0.5M / 1M / 2M rows on - by waves per workgroup and rows per ring row - a wave takes runs of four blocks and the
    # plain score output leaves as 256-byte stores: the last choice)
    n = int(rng.choice([rng.integers(4096, 4200), rng.integers(4200, 70_000), rng.integers(70_000, 600_000), rng.integers(520_000, 2_300_000)]))
    if n > 600_000:
        chunk = 1
    dim = m * chunk - (int(rng.integers(0, chunk)) if m % 16 else 0)
    dist = [D.Dot, D.L2, D.L1][int(rng.integers(0, 3))]
    invert, largest = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    k = int(rng.choice([1, 10, 30, 64, 100]))
    cen = (rng.random((256, dim), dtype=np.float32) - 0.5).astype(np.float32)
    cen[rng.integers(0, 256, size=16)] = 0.0
    g = torch.Generator(device="cuda")
    g.manual_seed(int(rng.integers(0, 1 << 30)))
    vp = qa.VectorParameters(dim, n, dist, invert)
    mm = qa.EncodedVectorsPQ.get_quantized_vector_size(vp, chunk)
    rows = torch.randint(0, 256, (n, mm), generator=g, device=dev, dtype=torch.uint8)
    enc = qa.EncodedVectorsPQ.from_storage(rows, vp, chunk, cen)
    del rows
    q = enc.encode_query((rng.random(dim, dtype=np.float32) - 0.5).astype(np.float32))
    scores = np.asarray(enc.score_all(q))
    by_ids = np.asarray(enc.score_ids(q, np.arange(n, dtype=np.uint32)))
    ok = np.array_equal(scores.view(np.uint32), by_ids.view(np.uint32))
    ids, sc = enc.topk(q, k, largest=largest)
    order = np.lexsort((np.arange(n), -scores if largest else scores))[:k]
    ok_topk = np.array_equal(np.asarray(sc).view(np.uint32), scores[order].view(np.uint32)) and \
        np.array_equal(np.sort(scores[np.asarray(ids)]), np.sort(scores[order]))
    bad += not (ok and ok_topk)
    print(f"case {case:3d}: n={n:7d} m={mm:4d} chunk={chunk} k={k:3d} {dist} invert={invert} largest={largest}  "
          f"{'ok' if ok and ok_topk else 'FAILED scan=%s topk=%s' % (ok, ok_topk)}", flush=True)
    del enc
print(f"{cases} cases, {bad} failed")
sys.exit(1 if bad else 0)
