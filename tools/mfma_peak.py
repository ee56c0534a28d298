#!/usr/bin/env python3
"""Developer measurement (dev library): the int8 MFMA issue-rate ceiling of this box - back-to-back
v_mfma_i32_32x32x32_i8 on register operands, nothing else.  QAMD_LIB_PATH must point at
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

from quantization_amd import _lib  # noqa: E402

L = _lib.lib()
sink = torch.zeros(16, dtype=torch.int32, device="cuda")
rep = C.create_string_buffer(1 << 14)
st = L.qamd_dev_mfma_peak(C.c_void_p(sink.data_ptr()), rep, C.c_size_t(1 << 14))
print("status", st, L.qamd_last_error())
print(rep.value.decode())
