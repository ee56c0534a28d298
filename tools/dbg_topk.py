import os, sys
sys.path.insert(0, "/root/repo")
import torch
import quantization_amd as qa
dev = torch.device("cuda", 0)
n, dim = 10_000_000, 768
data = torch.rand((n, dim), device=dev)
enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, qa.DistanceType.Dot, False))
del data
for i in range(4):
    q = enc.encode_query(torch.rand(dim, device=dev))
    enc.topk(q, 30)
    enc.topk(q, 30, largest=False)
