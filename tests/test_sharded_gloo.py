"""N>1 path on CPU: world_size 2, gloo.  The sharding/exchange logic of
quantization_amd/sharded.py is backend-independent; here the per-shard scorer is the oracle
(tests may use it), on the GPU it is the HIP scan (bench.py)."""
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


def _worker(rank, world, port, n, dim, tmp):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import qoracle as qo
    from quantization_amd.sharded import (ScoreGather, ShardedTopK, ShardedTopKBatch, assemble_global_scores,
                                          max_shard_rows, shard_range)

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)
    data = rng.random((n, dim), dtype=np.float32)
    queries = rng.random((3, dim), dtype=np.float32)
    alpha, offset = np.float32(1.0) / np.float32(127.0), np.float32(0.0)
    b, e = shard_range(n, rank, world)
    rows, meta = qo.u8_encode_with(data[b:e], qo.DOT, False, alpha, offset)  # this rank's shard only
    pad = max_shard_rows(n, world)
    gather = ScoreGather(dist, torch, pad, "cpu", rank, world, dst=0)
    rotating = ScoreGather(dist, torch, pad, "cpu", rank, world, dst=None)  # root = step % world
    grouped = ScoreGather(dist, torch, pad, "cpu", rank, world, dst=None, group_steps=2)  # 2 queries per collective
    topk = ShardedTopK(dist, torch, 10, "cpu", rank, world, n)
    g_rows, g_meta = qo.u8_encode_with(data, qo.DOT, False, alpha, offset)
    results = []
    for step, q in enumerate(queries):
        codes, qoff = qo.u8_encode_query(meta, q)
        local = qo.u8_score_all(meta, rows, codes, qoff)
        slot = gather.slot(step)
        slot[: e - b] = torch.from_numpy(local)
        gather.submit(step)
        # per-shard top-k (numpy stand-in for the device selection), then the exchange
        order = np.lexsort((np.arange(local.size), -local))[:10]
        ids, sc = topk.buffers()
        ids[:] = torch.from_numpy(order.astype(np.int32))
        sc[:] = torch.from_numpy(local[order])
        merged_ids, merged_sc = topk.exchange(largest=True)
        rslot = rotating.slot(step)
        rslot[: e - b] = torch.from_numpy(local)
        rotating.submit(step)
        gslot = grouped.slot(step)
        gslot[: e - b] = torch.from_numpy(local)
        grouped.submit(step)  # sends after steps 1 (full group) — step 2's half-filled group goes at drain()
        got = gather.collect(step)
        want = qo.u8_score_all(g_meta, g_rows, codes, qoff)
        if rank == 0:
            flat = assemble_global_scores(got, n, world).numpy()
            assert np.array_equal(flat.view(np.uint32), want.view(np.uint32)), "gathered scores differ"
        got_r = rotating.collect(step)
        assert (got_r is not None) == (rank == step % world), "rotating root"
        if got_r is not None:
            flat = assemble_global_scores(got_r, n, world).numpy()
            assert np.array_equal(flat.view(np.uint32), want.view(np.uint32)), "rotating-root gathered scores differ"
        worder = np.lexsort((np.arange(n), -want))[:10]
        assert np.array_equal(merged_ids, worder.astype(np.uint32)), "merged top-k ids differ"
        assert np.array_equal(merged_sc, want[worder])
        results.append(want)
    grouped.drain()
    for step, want in enumerate(results):  # group 0 = steps 0, 1 -> rank 0; group 1 = step 2 -> rank 1
        assert grouped.root(step) == (step // 2) % world
        got_g = grouped.collect(step)
        assert (got_g is not None) == (rank == grouped.root(step))
        if got_g is not None:
            flat = assemble_global_scores(got_g, n, world).numpy()
            assert np.array_equal(flat.view(np.uint32), want.view(np.uint32)), "grouped gather differs"
    # batched exchange: all three queries at once, k = 7 (some shards hold fewer than k rows when n is tiny)
    kb = 7
    batch = ShardedTopKBatch(dist, torch, len(queries), kb, "cpu", rank, world, n)
    bids, bsc = batch.buffers()
    wants = []
    for qi, q in enumerate(queries):
        codes, qoff = qo.u8_encode_query(meta, q)
        local = qo.u8_score_all(meta, rows, codes, qoff)
        order = np.lexsort((np.arange(local.size), -local))[:kb]
        ids_q = np.full(kb, -1, dtype=np.int32)  # 0xFFFFFFFF padding
        sc_q = np.full(kb, -np.inf, dtype=np.float32)
        ids_q[: order.size] = order
        sc_q[: order.size] = local[order]
        bids[qi * kb:(qi + 1) * kb] = torch.from_numpy(ids_q)
        bsc[qi * kb:(qi + 1) * kb] = torch.from_numpy(sc_q)
        wants.append(qo.u8_score_all(g_meta, g_rows, codes, qoff))
    mids, msc = batch.exchange(largest=True)
    for qi, want in enumerate(wants):
        worder = np.lexsort((np.arange(n), -want))[:kb]
        assert np.array_equal(mids[qi][: worder.size], worder.astype(np.uint32)), "batched merged ids differ"
        assert np.array_equal(msc[qi][: worder.size], want[worder])
    gather.drain()
    rotating.drain()
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


@pytest.mark.parametrize("n", [1001, 64])
def test_world2_gather_and_topk(tmp_path, n):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, 48, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_shard_ranges_tile_the_store():
    from quantization_amd.sharded import max_shard_rows, shard_range

    for count in (0, 1, 7, 8, 10_000_000, 50_000_001):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(count, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == count
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max_shard_rows(count, world) - min(e - b for b, e in edges) <= 1


def test_merge_topk_ties_and_padding():
    from quantization_amd.sharded import merge_topk

    ids = np.array([[0, 2, 0xFFFFFFFF], [1, 0, 3]], dtype=np.uint32)
    sc = np.array([[5.0, 4.0, -np.inf], [5.0, 4.0, 1.0]], dtype=np.float32)
    out_ids, out_sc = merge_topk(ids, sc, [0, 100], 4, True)
    assert out_ids.tolist() == [0, 101, 2, 100] and out_sc.tolist() == [5.0, 5.0, 4.0, 4.0]
    out_ids, out_sc = merge_topk(ids[:, :1], sc[:, :1], [0, 100], 4, True)
    assert out_ids.tolist() == [0, 101, 0xFFFFFFFF, 0xFFFFFFFF]
