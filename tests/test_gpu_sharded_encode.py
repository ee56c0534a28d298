"""Distributed encode of the rank-per-GPU route ON the GPU: ranks rehearsed on one device over gloo (RCCL needs one GPU
per rank; the collectives are backend-independent and covered on CPU by tests/test_sharded_encode_gloo.py), the per-rank
operations through the C ABI (quantization_amd.sharded.LibraryOps: qamd_u8_find_min_max, qamd_u8_find_quantile_interval,
qamd_pq_find_centroids, the encoders).  Every rank's shard must equal the rows of the single-handle encode of the
concatenated data - and, where the oracle is deterministic, the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    import quantization_amd as qa
    import quantization_amd.sharded as sh
    from oracle import qoracle as qo

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    qa.set_device(0)
    D = qa.DistanceType
    rng = np.random.default_rng(3)

    def bits(x):
        return np.float32(x).view(np.uint32)

    # scalar u8: plain interval, quantile with count <= 100 000 (the reference's deterministic case), quantile with
    # count > 100 000 (the strided 100 000-row sample gathered to rank 0)
    for n, dim, dist_t, invert, quantile in ((20_011, 48, D.Dot, False, None), (5_003, 65, D.L2, True, 0.97),
                                             (130_001, 16, D.Dot, False, 0.99)):
        data = (rng.random((n, dim), dtype=np.float32) - np.float32(0.3)).astype(np.float32)
        b, e = sh.shard_range(n, rank, world)
        vp = qa.VectorParameters(dim, n, dist_t, invert)
        local = torch.from_numpy(data[b:e]).cuda()
        enc, (alpha, offset) = sh.encode_u8(dist, torch, local, vp, quantile)
        full = qa.EncodedVectorsU8.encode(torch.from_numpy(data).cuda(), vp, quantile)
        md, fmd = enc.metadata, full.metadata
        for key in ("alpha", "offset", "multiplier"):
            assert bits(md[key]) == bits(fmd[key]), f"{key} differs from the single-handle encode (n={n})"
        assert md["actual_dim"] == fmd["actual_dim"] and md["vector_parameters"].count == e - b
        assert np.array_equal(enc.storage_bytes(), full.storage_rows(b, e - b)), f"u8 shard rows differ (n={n}, rank {rank})"
        if n <= 100_000:
            rows, meta = qo.u8_encode(data, int(dist_t), invert, quantile)
            assert np.array_equal(enc.storage_bytes(), rows[b:e]) and bits(alpha) == bits(meta.alpha) and bits(offset) == bits(meta.offset)
        # the shard answers queries like its slice of the whole store
        q = rng.random(dim, dtype=np.float32)
        assert np.array_equal(enc.score_all(enc.encode_query(q)).view(np.uint32),
                              full.score_all(full.encode_query(q))[b:e].view(np.uint32))

    # PQ: the 10 000-row k-means sample gathered to rank 0, trained there, centroids broadcast
    n, dim, chunk = 24_001, 32, 4
    data = rng.random((n, dim), dtype=np.float32)
    b, e = sh.shard_range(n, rank, world)
    vp = qa.VectorParameters(dim, n, D.L2, False)
    enc, cen = sh.encode_pq(dist, torch, torch.from_numpy(data[b:e]).cuda(), vp, chunk, 2)
    full = qa.EncodedVectorsPQ.encode(torch.from_numpy(data).cuda(), vp, chunk, max_kmeans_threads=2)
    assert np.array_equal(np.asarray(cen).view(np.uint32), full.centroids.view(np.uint32)), "centroids differ from the single-handle encode"
    assert np.array_equal(enc.storage_bytes(), full.storage_rows(b, e - b)), "PQ shard codes differ"
    want_cen = qo.find_centroids(data, chunk, max_threads=2)[0]
    if full.kmeans_info()[1] == 0:  # no empty cluster was re-seeded: the oracle's restatement of kmeans.rs gives the same bits
        assert np.array_equal(np.asarray(cen).view(np.uint32), want_cen.view(np.uint32))

    # binary: no global statistic
    n, dim = 3_001, 200
    data = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    b, e = sh.shard_range(n, rank, world)
    enc = sh.encode_binary(dist, torch, torch.from_numpy(data[b:e]).cuda(), qa.VectorParameters(dim, n, D.Dot, False))
    assert np.array_equal(enc.storage_bytes(), qo.bin_encode(data)[b:e])

    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_encode_on_the_gpu_equals_the_single_handle_encode(tmp_path, world):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
