#!/usr/bin/env python3
"""Developer fuzz (GPU box): random bursts of random-access pairs (score_ids_batch, score_internal_ids,
score_internal_ids_batch) on random stores of the three quantizers, host and device lists, ragged / empty lists,
against the single-list calls of the same handle (score_ids per query; score_internal per pair for a sample),
which tests/test_gpu_bursts.py pins to the oracle.
    python tools/fuzz_bursts.py [cases] [seed]"""
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
t0 = time.time()
bad = 0


def bits(a):
    a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
    return a.view(np.uint32)


for case in range(cases):
    kind = ["u8", "bin", "pq"][int(rng.integers(0, 3))]
    n = int(rng.choice([rng.integers(1, 200), rng.integers(200, 5000), rng.integers(5000, 120_000)]))
    dist = [D.Dot, D.L1, D.L2][int(rng.integers(0, 3))]
    invert = bool(rng.integers(0, 2))
    if kind == "u8":
        dim = int(rng.choice([1, 16, 17, 65, 100, 128, 300, 768, 1000, 1536, 2064]))
        data = rng.random((n, dim), dtype=np.float32) - np.float32(0.3)
        enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, dist, invert))
    elif kind == "bin":
        dim = int(rng.choice([1, 8, 20, 33, 64, 65, 128, 129, 387, 1024, 2065]))
        data = rng.random((n, dim), dtype=np.float32) - np.float32(0.5)
        enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, dist, invert))
    else:
        dim, cs = [(96, 1), (100, 7), (768, 8), (768, 4), (130, 2), (60, 4), (1536, 8)][int(rng.integers(0, 7))]
        data = rng.random((n, dim), dtype=np.float32)
        cen = rng.random((256, dim), dtype=np.float32)
        enc = qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(dim, n, dist, invert), cs, centroids=cen)
    n_lists = int(rng.choice([1, 2, 7, 64, 300]))
    lens = rng.integers(0, int(rng.choice([2, 40, 700])), n_lists)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    n_ids = int(offs[-1])
    ids = rng.integers(0, n, n_ids).astype(np.uint32)
    rows = rng.integers(0, n, n_lists).astype(np.uint32)
    queries = rng.random((n_lists, dim), dtype=np.float32) - np.float32(0.4)
    batch = enc.encode_query_batch(queries)
    on_device = bool(rng.integers(0, 2))
    if on_device:
        t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(dev)
        out = torch.empty(max(n_ids, 1), dtype=torch.float32, device=dev)[:n_ids]
        got_q = bits(enc.score_ids_batch(batch, t(offs), t(ids), out=out)).copy()
        got_i = bits(enc.score_internal_ids_batch(t(rows), t(offs), t(ids), out=out)).copy()
    else:
        got_q = bits(enc.score_ids_batch(batch, offs, ids)).copy()
        got_i = bits(enc.score_internal_ids_batch(rows, offs, ids)).copy()
    ok = True
    qobj = None
    for l in rng.permutation(n_lists)[:12]:
        a, b = int(offs[l]), int(offs[l + 1])
        if a == b:
            continue
        qobj = enc.encode_query(queries[l], reuse=qobj)
        if not np.array_equal(got_q[a:b], bits(enc.score_ids(qobj, ids[a:b]))):
            ok = False
            print(f"  MISMATCH score_ids_batch list {l}")
        if not np.array_equal(got_i[a:b], bits(enc.score_internal_ids(int(rows[l]), ids[a:b]))):
            ok = False
            print(f"  MISMATCH score_internal_ids_batch list {l}")
        j = int(rng.integers(a, b))
        if got_i[j] != np.float32(enc.score_internal(int(rows[l]), int(ids[j]))).view(np.uint32):
            ok = False
            print(f"  MISMATCH score_internal pair {j}")
        if got_q[j] != np.float32(enc.score_point(qobj, int(ids[j]))).view(np.uint32):
            ok = False
            print(f"  MISMATCH score_point pair {j}")
    bad += not ok
    print(f"case {case:3d}: {kind} n={n:7d} dim={dim:5d} {dist} invert={invert} lists={n_lists} ids={n_ids} "
          f"{'device' if on_device else 'host'}  {'ok' if ok else 'FAILED'}", flush=True)
    del enc, batch
print(f"{cases} cases, {bad} failed, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
