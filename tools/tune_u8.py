#!/usr/bin/env python3
"""Runs the developer sweep of tune.hip on a 10M x 768 store (GPU box)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import quantization_amd as qa  # noqa: E402
from quantization_amd import _lib  # noqa: E402

n, dim = int(os.environ.get("ROWS", 10_000_000)), 768
L = _lib.lib()
dev = torch.device("cuda", 0)
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
q = enc.encode_query(torch.rand(dim, device=dev))
out = torch.zeros(n + 1024, dtype=torch.float32, device=dev)
out.view(torch.int32)[n + 1] = 0x3A000000  # filter experiment pivot key (lets ~nothing through)
ref = torch.empty(n, dtype=torch.float32, device=dev)
enc.score_all(q, out=ref)
codes, offs = C.c_void_p(), C.c_void_p()
L.qamd_dev_u8_ptrs(enc._h, C.byref(codes), C.byref(offs))
L.qamd_dev_u8_query_ptr.restype = C.c_void_p
qp = C.c_void_p(L.qamd_dev_u8_query_ptr(q._h))
rep = C.create_string_buffer(1 << 16)
torch.cuda.synchronize()
st = L.qamd_dev_u8_sweep(codes, offs, qp, C.c_float(float(enc.metadata["multiplier"])), C.c_uint32(n),
                         C.c_void_p(out.data_ptr()), int(os.environ.get("ROUNDS", 5)), rep, C.c_size_t(1 << 16))
print("status", st, L.qamd_last_error())
print(rep.value.decode())
torch.cuda.synchronize()
print("last variant output equals shipped kernel:", bool(torch.equal(out[:n], ref)))
