"""PCIe-inclusive rates (host buffers across the C ABI) for DESIGN.md — never the bench `value`."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import quantization_amd as qa
n, dim = 2_000_000, 768
rng = np.random.default_rng(0)
data = rng.random((n, dim), dtype=np.float32)
vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
t0 = time.perf_counter(); enc = qa.EncodedVectorsU8.encode(data, vp); t1 = time.perf_counter()
print(json.dumps({"what": "u8 encode from HOST f32 (pageable), 2M x 768", "seconds": round(t1 - t0, 3),
                  "input_GBps": round(data.nbytes / (t1 - t0) / 1e9, 2), "rows_per_s": round(n / (t1 - t0))}))
q = rng.random(dim, dtype=np.float32)
qo = enc.encode_query(q)
for _ in range(3): enc.score_all(qo)
t0 = time.perf_counter()
for _ in range(10): s = enc.score_all(enc.encode_query(q))
t1 = time.perf_counter()
print(json.dumps({"what": "encode_query(host) + score_all -> HOST scores, 2M x 768", "ms_per_query": round((t1 - t0) * 100, 3),
                  "rows_per_s": round(n * 10 / (t1 - t0)), "d2h_GBps_of_scores": round(n * 4 * 10 / (t1 - t0) / 1e9, 2)}))
t0 = time.perf_counter()
for _ in range(10): ids, sc = enc.topk(enc.encode_query(q), 30)
t1 = time.perf_counter()
print(json.dumps({"what": "encode_query(host) + topk(30) -> HOST ids/scores, 2M x 768", "ms_per_query": round((t1 - t0) * 100, 3),
                  "rows_per_s": round(n * 10 / (t1 - t0))}))
t0 = time.perf_counter(); rows = enc.storage_bytes(); t1 = time.perf_counter()
print(json.dumps({"what": "export_rows to HOST (reference row format)", "GBps": round(rows.nbytes / (t1 - t0) / 1e9, 2)}))
t0 = time.perf_counter(); e2 = qa.EncodedVectorsU8.from_storage(rows, enc.metadata); t1 = time.perf_counter()
print(json.dumps({"what": "from_storage from HOST rows", "GBps": round(rows.nbytes / (t1 - t0) / 1e9, 2)}))
