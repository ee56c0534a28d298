import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import quantization_amd as qa
n, dim = 2_000_000, 768
data = np.random.default_rng(0).random((n, dim), dtype=np.float32)
torch.cuda.init(); torch.zeros(1, device="cuda")
def t(f, label):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(f"{label}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True); return r
d = t(lambda: torch.empty((n, dim), device="cuda"), "torch.empty 6.1 GB (hipMalloc)")
t(lambda: d.copy_(torch.from_numpy(data)), "H2D copy pageable 6.1 GB, first touch")
t(lambda: d.copy_(torch.from_numpy(data)), "H2D copy pageable 6.1 GB, second")
vp = qa.VectorParameters(dim, n, qa.DistanceType.Dot, False)
e = t(lambda: qa.EncodedVectorsU8.encode(d, vp), "encode from device")
e = t(lambda: qa.EncodedVectorsU8.encode(data, vp), "encode from host (1)")
e = t(lambda: qa.EncodedVectorsU8.encode(data, vp), "encode from host (2)")
