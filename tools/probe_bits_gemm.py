#!/usr/bin/env python3
"""Developer probe (dev library): binary rows (1024 bits) expanded to 0/1 bytes in registers and fed to int8
MFMAs against an LDS-resident query tile - the K loop a many-queries binary path would have."""
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

n = int(os.environ.get("ROWS", 50_000_000))
L = _lib.lib()
rows = torch.randint(0, 256, (n, 128), device="cuda", dtype=torch.uint8)
sink = torch.zeros(16, dtype=torch.int32, device="cuda")
rep = C.create_string_buffer(1 << 14)
st = L.qamd_dev_bits_gemm_probe(C.c_void_p(rows.data_ptr()), C.c_uint32(n), C.c_void_p(sink.data_ptr()), rep, C.c_size_t(1 << 14))
print("status", st, L.qamd_last_error())
print(rep.value.decode())
