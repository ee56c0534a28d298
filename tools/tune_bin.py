#!/usr/bin/env python3
"""Developer sweep of the binary scan (50M x 1024 bits) — see tune.hip."""
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

n, dim = int(os.environ.get("ROWS", 50_000_000)), 1024
L = _lib.lib()
dev = torch.device("cuda", 0)
vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
rows = torch.randint(0, 256, (n, 128), device=dev, dtype=torch.uint8)
enc = qa.EncodedVectorsBin.from_storage(rows, vp)
del rows
q = enc.encode_query(torch.randn(dim, device=dev))
out = torch.empty(n, dtype=torch.float32, device=dev)
ref = torch.empty(n, dtype=torch.float32, device=dev)
enc.score_all(q, out=ref)
L.qamd_dev_bin_rows.restype = C.c_void_p
L.qamd_dev_bin_query_ptr.restype = C.c_void_p
rep = C.create_string_buffer(1 << 16)
torch.cuda.synchronize()
st = L.qamd_dev_bin_sweep(C.c_void_p(L.qamd_dev_bin_rows(enc._h)), C.c_void_p(L.qamd_dev_bin_query_ptr(q._h)),
                          C.c_float(dim), C.c_uint32(n), C.c_void_p(out.data_ptr()),
                          int(os.environ.get("ROUNDS", 5)), rep, C.c_size_t(1 << 16))
print("status", st)
print(rep.value.decode())
torch.cuda.synchronize()
print("last variant output equals shipped kernel:", bool(torch.equal(out, ref)))
