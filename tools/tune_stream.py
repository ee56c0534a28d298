#!/usr/bin/env python3
"""Developer sweep (dev library): pure-load kernels with the access patterns considered for the
row-streaming MFMA kernel, on a 10M x 768 u8 store.  QAMD_LIB_PATH must point at
libquantization_amd_dev.so (make -C quantization_amd/csrc dev)."""
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

n, dim = int(os.environ.get("ROWS", 10_000_000)), int(os.environ.get("DIM", 768))
L = _lib.lib()
dev = torch.device("cuda", 0)
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
sink = torch.zeros(16, dtype=torch.int32, device=dev)
codes, offs = C.c_void_p(), C.c_void_p()
L.qamd_dev_u8_ptrs(enc._h, C.byref(codes), C.byref(offs))
rep = C.create_string_buffer(1 << 16)
torch.cuda.synchronize()
st = L.qamd_dev_stream_sweep(codes, C.c_uint32(n), C.c_uint32(dim), C.c_void_p(sink.data_ptr()),
                             int(os.environ.get("ROUNDS", 5)), rep, C.c_size_t(1 << 16))
print("status", st, L.qamd_last_error())
print(rep.value.decode())
