"""Distributed encode of the rank-per-GPU route on CPU: world sizes 2 and 3, gloo.  quantization_amd/sharded.py's
encode_u8 / encode_pq / encode_binary agree on the reference's global statistics (one (alpha, offset),
encoded_vectors_u8.rs:57-71; one set of centroids, encoded_vectors_pq.rs:278-342) with collectives and then encode each
rank's rows; here the per-rank operations are the oracle's (tests may use it; on the GPU they are the C ABI's,
tests/test_gpu_bench.py) and every rank's row bytes and metadata must equal the single encode of the concatenated data."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleOps:
    """quantization_amd.sharded.LibraryOps with the oracle behind every operation (CPU)."""

    def __init__(self, qo):
        self.qo = qo

    def find_min_max(self, rows):
        return self.qo.find_min_max(np.asarray(rows))

    def find_quantile_interval(self, rows, quantile):
        return self.qo.find_quantile_interval(np.asarray(rows), quantile)

    def find_centroids(self, rows, chunk_size, max_kmeans_threads):
        rows = np.asarray(rows)
        if rows.shape[0] <= 256:
            return self.qo.pq_centroids_small(rows)
        return self.qo.find_centroids(rows, chunk_size, max_threads=max_kmeans_threads)[0]

    def encode_u8(self, rows, vp, alpha_offset):
        rows = np.asarray(rows, dtype=np.float32).reshape(vp.count, vp.dim)
        if alpha_offset is None:
            return self.qo.u8_encode_empty(vp.dim, int(vp.distance_type), vp.invert)
        return self.qo.u8_encode_with(rows, int(vp.distance_type), vp.invert, *alpha_offset)

    def encode_pq(self, rows, vp, chunk_size, centroids):
        return self.qo.pq_encode(np.asarray(rows, dtype=np.float32).reshape(vp.count, vp.dim), chunk_size, centroids)

    def encode_binary(self, rows, vp):
        return self.qo.bin_encode(np.asarray(rows, dtype=np.float32).reshape(vp.count, vp.dim))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import quantization_amd.sharded as sh
    from oracle import qoracle as qo
    from quantization_amd.encoded_vectors import DistanceType, VectorParameters

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ops = OracleOps(qo)
    rng = np.random.default_rng(11)

    def meta_tuple(m):
        return (m.actual_dim, np.float32(m.alpha).view(np.uint32), np.float32(m.offset).view(np.uint32),
                np.float32(m.multiplier).view(np.uint32))

    # --- scalar u8: min/max interval and quantile interval (count <= 100 000: the deterministic case of the reference)
    for n, dim, dist_t, invert, quantile, sample_cap in ((1003, 24, DistanceType.Dot, False, None, None),
                                                         (1003, 24, DistanceType.L2, True, 0.98, None),
                                                         (130, 7, DistanceType.L1, False, 0.9, None),
                                                         (5, 9, DistanceType.Dot, True, None, None),
                                                         (2, 16, DistanceType.Dot, False, 0.99, None),   # a rank with no rows (world 3)
                                                         (3000, 8, DistanceType.Dot, False, 0.95, 1000)):  # the strided sample rule
        data = (rng.random((n, dim), dtype=np.float32) - np.float32(0.25)).astype(np.float32)
        if n == 1003:
            data[17, 3] = np.float32(-0.0)
            data[600, 1] = np.float32("nan")  # NaN never wins a compare (quantile.rs:9-16)
        b, e = sh.shard_range(n, rank, world)
        vp = VectorParameters(dim, n, dist_t, invert)
        if sample_cap:  # exercise "count > QUANTILE_SAMPLE_SIZE" at a size the oracle sorts in no time
            sh.QUANTILE_SAMPLE_SIZE = sample_cap
            picks = (np.arange(sample_cap, dtype=np.uint64) * np.uint64(n)) // np.uint64(sample_cap)
            want_iv = qo.find_quantile_interval(data[picks.astype(np.int64)], quantile)
            mn, mx = want_iv if want_iv is not None else qo.find_min_max(data)
            a, o = qo.alpha_offset(mn, mx)
            g_rows, g_meta = qo.u8_encode_with(data, int(dist_t), invert, a, o)
        else:
            sh.QUANTILE_SAMPLE_SIZE = 100_000
            g_rows, g_meta = qo.u8_encode(data, int(dist_t), invert, quantile)
        (rows, meta), (alpha, offset) = sh.encode_u8(dist, torch, data[b:e], vp, quantile, ops=ops)
        assert np.array_equal(rows, g_rows[b:e]), f"u8 shard rows differ (n={n}, rank {rank}/{world})"
        assert meta_tuple(meta) == meta_tuple(g_meta), f"u8 metadata differs (n={n}, rank {rank}/{world})"
        assert np.float32(alpha).view(np.uint32) == np.float32(g_meta.alpha).view(np.uint32)
        assert np.float32(offset).view(np.uint32) == np.float32(g_meta.offset).view(np.uint32)
    # empty store (encoded_vectors_u8.rs:43-54)
    (rows, meta), _ = sh.encode_u8(dist, torch, np.zeros((0, 12), dtype=np.float32), VectorParameters(12, 0, DistanceType.Dot, False),
                                   ops=ops)
    assert rows.shape[0] == 0 and meta.alpha == 0 and meta.multiplier == 0

    # --- PQ: the k-means sample gathered to rank 0, centroids broadcast (count > 256 and count <= 256)
    for n, dim, chunk in ((700, 6, 2), (200, 10, 3)):
        data = rng.random((n, dim), dtype=np.float32)
        b, e = sh.shard_range(n, rank, world)
        sh.KMEANS_SAMPLE_SIZE = 300 if n == 700 else 10_000  # "count > KMEANS_SAMPLE_SIZE": rows floor(k * count / S)
        want_cen = (qo.pq_centroids_small(data) if n <= 256 else
                    qo.find_centroids(data, chunk, sample_rows=qo.pq_sample_rows(n, sh.KMEANS_SAMPLE_SIZE), max_threads=2)[0])
        rows, cen = sh.encode_pq(dist, torch, data[b:e], VectorParameters(dim, n, DistanceType.L2, False), chunk, 2, ops=ops)
        assert np.array_equal(np.asarray(cen).view(np.uint32), want_cen.view(np.uint32)), f"centroids differ (rank {rank}/{world})"
        assert np.array_equal(rows, qo.pq_encode(data, chunk, want_cen)[b:e]), "PQ shard codes differ"

    # --- binary: no global statistic
    data = np.where(rng.random((77, 130)) < 0.5, -1.0, 1.0).astype(np.float32)
    b, e = sh.shard_range(77, rank, world)
    rows = sh.encode_binary(dist, torch, data[b:e], VectorParameters(130, 77, DistanceType.Dot, False), ops=ops)
    assert np.array_equal(rows, qo.bin_encode(data)[b:e])

    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_encode_equals_the_single_encode(tmp_path, world):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp

    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_sample_rows_of_shard_tile_the_sample():
    from quantization_amd.sharded import sample_rows_of_shard, shard_range

    for count in (0, 1, 5, 127, 9999, 10000, 10001, 123457):
        for S in (300, 10000):
            ss = min(S, count)
            full = (np.arange(ss, dtype=np.uint64) * np.uint64(count)) // np.uint64(max(ss, 1))
            for world in (1, 2, 3, 8):
                got = []
                for r in range(world):
                    b, e = shard_range(count, r, world)
                    loc = sample_rows_of_shard(count, S, b, e)
                    assert ((loc >= 0) & (loc < e - b)).all()
                    got.append(loc + b)
                assert np.array_equal(np.concatenate(got).astype(np.uint64), full)


def test_min_max_keys_order_like_f32():
    from quantization_amd.sharded import _f32_key, _key_f32

    vals = [np.float32(x) for x in (-3.4e38, -1.5, -1e-45, -0.0, 0.0, 1e-45, 1.0, 3.4e38)]
    keys = [_f32_key(v) for v in vals]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    for v, k in zip(vals, keys):
        assert _key_f32(k).view(np.uint32) == v.view(np.uint32)
