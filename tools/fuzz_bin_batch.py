#!/usr/bin/env python3
"""Developer fuzz (GPU box): random shapes through the binary topk_batch (matrix-core path from 16 queries and
32768 rows on) against the single-query top-k.    python tools/fuzz_bin_batch.py [cases] [seed]"""
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
bad = 0
for case in range(cases):
    nq = int(rng.choice([5, 11, 12, 13, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 128, 129, 200, 256, 257, 288, 289, 300, 576, 577, 700, 1000, 1152, 1153]))
    dim = int(rng.choice([64, 65, 100, 128, 129, 256, 384, 512, 500, 768, 700, 1000, 1024, 1536, 1500, 2048, 2304, 2320, 4096, 4992, 5000, 8192]))
    n = int(rng.choice([rng.integers(32_768, 40_000), rng.integers(40_000, 300_000), rng.integers(1000, 32_768)]))
    dist = [D.Dot, D.L2, D.L1][int(rng.integers(0, 3))]
    invert, largest = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    k = int(rng.choice([1, 10, 30, 64, 100]))
    store = int(rng.integers(0, 2))
    g = torch.Generator(device="cuda")
    g.manual_seed(int(rng.integers(0, 1 << 30)))
    data = torch.randn((n, dim), generator=g, device=dev)
    queries = torch.randn((nq, dim), generator=g, device=dev)
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, dist, invert))
    del data
    ids, sc = enc.topk_batch(enc.encode_query_batch(queries), k, largest=largest)
    qh = queries.cpu().numpy()
    ok = True
    for qi in sorted(set(int(x) for x in rng.integers(0, nq, 5)) | {0, nq - 1}):
        wi, ws = enc.topk(enc.encode_query(qh[qi]), k, largest=largest)
        if not (np.array_equal(ids[qi], wi) and np.array_equal(sc[qi].view(np.uint32), ws.view(np.uint32))):
            ok = False
            print(f"  MISMATCH query {qi}")
    bad += not ok
    print(f"case {case:3d}: n={n:7d} dim={dim:5d} nq={nq:4d} k={k:3d} {dist} invert={invert} largest={largest}  {'ok' if ok else 'FAILED'}", flush=True)
    del enc
print(f"{cases} cases, {bad} failed")
sys.exit(1 if bad else 0)
