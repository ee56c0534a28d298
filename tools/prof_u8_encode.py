"""Profiling driver: u8 encode (min/max pass + quantize pass) of a device-resident 2M x 768 f32 store,
five calls -- put after `rocprofv3 --kernel-trace --stats ... --` (tools/kstats.sh)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim = 2_000_000, int(os.environ.get("DIM", 768))
data = torch.rand((n, dim), device=dev)
vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
for _ in range(5):
    enc = qa.EncodedVectorsU8.encode(data, vp)
    torch.cuda.synchronize()
    del enc
