"""Row-sharded scan across the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference has no multi-device path: rows are independent, metadata is global
and tiny, the query is replicated.  Rank g owns the contiguous row range
[g*N/G, (g+1)*N/G); global row id = shard base + local id.  There is exactly one exchange per
query, after the local scan:

  * `gather_scores`: per-shard f32 scores -> rank `dst` (4 B/row over xGMI).  Double-buffered
    and asynchronous so the gather of query i overlaps the scan of query i+1.
  * `topk`: per-shard top-k, then an all-gather of G*k (id, score) pairs and a merge — what a
    caller that wants neighbours should use (the score gather moves 4 B/row over ~150 GB/s
    links while the scan reads 128-772 B/row at ~6 TB/s, so for binary rows it costs more than
    the scan itself).  On the GPU the merge is one kernel of the C ABI (`qamd_topk_merge`, the
    same one the single-process sharded handles use): the result stays in HBM, nothing is copied
    to the host per query.  The numpy merge below serves the CPU (gloo) tests only.

The single-PROCESS form of all this — one handle, one worker thread per GPU, peer copies instead
of collectives — is `sharded_store.py` over `qamd_*_sharded_*`.

The scoring itself is any object with the EncodedVectors API of this package; this module only
does the index arithmetic and the collectives, so the CPU tests drive it with a stand-in scorer.
"""
from __future__ import annotations

import numpy as np


def shard_range(count: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [begin, end) of `count` rows for `rank` of `world`."""
    return (rank * count) // world, ((rank + 1) * count) // world


def max_shard_rows(count: int, world: int) -> int:
    return max(shard_range(count, r, world)[1] - shard_range(count, r, world)[0] for r in range(world))


def _needs_host_staging(dist, tensor) -> bool:
    """gloo cannot gather CUDA tensors; stage through host memory then (rehearsal runs of the
    multi-rank path on a single GPU).  With nccl (= RCCL) tensors stay in HBM."""
    return getattr(tensor, "is_cuda", False) and dist.get_backend() == "gloo"


class ScoreGather:
    """Double-buffered asynchronous gather of per-shard score vectors to one rank per group of steps.

    slot(step) is the tensor of `rows_padded` f32 the scan of that step writes; submit(step) starts
    the gather once the group's last step has been submitted (group_steps queries travel in ONE
    collective: at 8 GPUs a 1.25M-row shard scan takes ~0.15 ms, the launch latency of a collective is
    tens of microseconds, so paying it per query would cost ~20 % of the step) and returns; the
    caller goes straight on to the next scan.  collect(step) waits (stream-ordered) and, on that
    step's root, returns the [world, rows_padded] tensor holding every shard's scores of that step
    (None elsewhere).

    dst = an int: every group gathers to that rank (its 7 inbound xGMI links carry everything).
    dst = None ("rotate"): group j gathers to rank j % world, so consecutive gathers use disjoint
    inbound links and overlap each other as well as the scans; each query's scores land on one GPU,
    round-robin, which is also how their post-processing would be balanced.
    """

    def __init__(self, dist, torch, rows_padded: int, device, rank: int, world: int, dst=0, group=None,
                 always_collective: bool = False, group_steps: int = 1):
        self.dist, self.torch = dist, torch
        self.rank, self.world, self.dst, self.group = rank, world, dst, group
        self.rows_padded = rows_padded
        self.B = max(1, int(group_steps))
        # always_collective: run the collective even at world size 1 (a one-rank gather is legal;
        # it lets a single-GPU box exercise the RCCL call path)
        self.single = world == 1 and not always_collective
        self.local = [torch.empty((self.B, rows_padded), dtype=torch.float32, device=device) for _ in range(2)]
        self.gathered = None
        if not self.single and (dst is None or rank == dst):
            self.gathered = [torch.empty((world, self.B, rows_padded), dtype=torch.float32, device=device)
                             for _ in range(2)]
        self.work = [None, None]
        self.pending = [0, 0]  # steps written into buffer s and not yet sent
        self.pending_step = [0, 0]

    def root(self, step: int) -> int:
        return (step // self.B) % self.world if self.dst is None else self.dst

    def _buf(self, step: int) -> int:
        return (step // self.B) % 2

    def slot(self, step: int):
        """Buffer the scan of `step` must write into (after the previous use has drained)."""
        s = self._buf(step)
        if step % self.B == 0 and self.work[s] is not None:
            self.work[s].wait()
            self.work[s] = None
        return self.local[s][step % self.B]

    def _send(self, s: int, dst: int) -> None:
        self.pending[s] = 0
        if self.single:  # single shard: the local scores ARE the global scores
            return
        local = self.local[s].view(-1)
        if _needs_host_staging(self.dist, local):
            host = local.cpu()
            hlist = [self.torch.empty_like(host) for _ in range(self.world)] if self.rank == dst else None
            self.dist.gather(host, gather_list=hlist, dst=dst, group=self.group)
            if self.rank == dst:
                self.gathered[s].copy_(self.torch.stack(hlist).view(self.world, self.B, self.rows_padded))
            return
        glist = list(self.gathered[s].view(self.world, -1).unbind(0)) if self.rank == dst else None
        self.work[s] = self.dist.gather(local, gather_list=glist, dst=dst, group=self.group, async_op=True)

    def submit(self, step: int) -> None:
        s = self._buf(step)
        self.pending[s] += 1
        self.pending_step[s] = step
        if step % self.B == self.B - 1:
            self._send(s, self.root(step))

    def flush(self, step: int) -> None:
        """Send a partly filled group (the last steps of a run)."""
        s = self._buf(step)
        if self.pending[s]:
            self._send(s, self.root(step))

    def collect(self, step: int):
        s = self._buf(step)
        if self.pending[s]:
            self._send(s, self.root(step))
        if self.work[s] is not None:
            self.work[s].wait()
            self.work[s] = None
        if self.single:
            return self.local[s][step % self.B].unsqueeze(0)
        return self.gathered[s][:, step % self.B] if self.rank == self.root(step) else None

    def drain(self) -> None:
        for s in range(2):
            if self.pending[s]:  # a partly filled group; every rank holds the same one, so this stays collective
                self._send(s, self.root(self.pending_step[s]))
            if self.work[s] is not None:
                self.work[s].wait()
                self.work[s] = None


def assemble_global_scores(gathered, count: int, world: int):
    """[world, rows_padded] shard-major scores -> flat [count] in global row order."""
    parts = []
    for r in range(world):
        b, e = shard_range(count, r, world)
        parts.append(gathered[r, : e - b])
    if hasattr(gathered, "numpy") and not isinstance(gathered, np.ndarray):
        import torch
        return torch.cat(parts)
    return np.concatenate(parts)


def merge_topk(ids_per_rank, scores_per_rank, bases, k: int, largest: bool):
    """Merge per-shard top-k lists (local ids + shard base -> global ids) into the global top-k,
    best first, ties to the lower global id — the same order a single-GPU topk returns.
    Inputs are numpy arrays [world, k]; ids 0xFFFFFFFF mark padding entries."""
    ids = np.asarray(ids_per_rank, dtype=np.uint32).astype(np.int64)
    sc = np.asarray(scores_per_rank, dtype=np.float32)
    valid = ids != 0xFFFFFFFF
    gids = ids + np.asarray(bases, dtype=np.int64)[:, None]
    gids, sc = gids[valid], sc[valid]
    order = np.lexsort((gids, -sc if largest else sc))[:k]
    out_ids = np.full(k, 0xFFFFFFFF, dtype=np.uint32)
    out_sc = np.full(k, -np.inf if largest else np.inf, dtype=np.float32)
    out_ids[: order.size] = gids[order].astype(np.uint32)
    out_sc[: order.size] = sc[order]
    return out_ids, out_sc


def _device_merge(torch, all_pairs, n_queries: int, k: int, bases, largest: bool, out):
    """[world][2][n_queries][k] gathered i32 bit patterns (ids plane, scores plane) in HBM ->
    out [2][n_queries][k] on the same device, through the C ABI's merge kernel."""
    import ctypes as C

    from . import _lib
    from .encoded_vectors import check, stream_ptr

    world = all_pairs.shape[0]
    per = n_queries * k
    arr = (C.c_uint64 * world)(*[int(b) for b in bases])
    base = all_pairs.data_ptr()
    check(_lib.lib().qamd_topk_merge(C.c_void_p(base), C.c_void_p(base + 4 * per), 2 * per, arr, world, n_queries, k,
                                     int(bool(largest)), C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 4 * per),
                                     _lib.MEM_DEVICE, stream_ptr(None)))
    return out[0], out[1].view(torch.float32)


class ShardedTopK:
    """Per-shard device top-k + all-gather of world*k pairs + merge (on the GPU: one kernel, the
    merged (ids, scores) stay in HBM as tensors; on CPU tensors: numpy)."""

    def __init__(self, dist, torch, k: int, device, rank: int, world: int, count: int, group=None):
        self.dist, self.torch, self.k = dist, torch, k
        self.rank, self.world, self.group = rank, world, group
        self.bases = [shard_range(count, r, world)[0] for r in range(world)]
        # one packed buffer: k ids (as i32 bit patterns) then k scores (as f32 bit patterns)
        self.pack = torch.empty(2 * k, dtype=torch.int32, device=device)
        self.all = torch.empty((world, 2 * k), dtype=torch.int32, device=device)
        self.merged = torch.empty((2, 1, k), dtype=torch.int32, device=device)

    def buffers(self):
        """(ids, scores) device views the local topk writes into."""
        return self.pack[: self.k], self.pack[self.k:].view(self.torch.float32)

    def exchange(self, largest: bool = True):
        if self.world == 1:
            self.all[0].copy_(self.pack)
        elif _needs_host_staging(self.dist, self.pack):
            hp = self.pack.cpu()
            ha = self.torch.empty((self.world, 2 * self.k), dtype=self.torch.int32)
            self.dist.all_gather_into_tensor(ha.view(-1), hp, group=self.group)
            self.all.copy_(ha)
        else:
            self.dist.all_gather_into_tensor(self.all.view(-1), self.pack, group=self.group)
        if self.all.is_cuda:  # merged on the device; callers that want host values copy them out
            ids, sc = _device_merge(self.torch, self.all.view(self.world, 2, 1, self.k), 1, self.k, self.bases, largest,
                                    self.merged)
            return ids[0], sc[0]
        host = self.all.numpy()
        ids = host[:, : self.k].view(np.uint32)
        sc = host[:, self.k:].view(np.float32)
        return merge_topk(ids, sc, self.bases, self.k, largest)


class ShardedTopKBatch:
    """Batched form of ShardedTopK (BASELINE config 4: many queries, top-k each, rows sharded):
    every rank runs `topk_batch` over its shard into `buffers()`, one all-gather moves
    world * n_queries * k (id, score) pairs, and each query's lists are merged with the
    single-GPU tie rule."""

    def __init__(self, dist, torch, n_queries: int, k: int, device, rank: int, world: int, count: int, group=None):
        self.dist, self.torch, self.k, self.nq = dist, torch, k, n_queries
        self.rank, self.world, self.group = rank, world, group
        self.bases = [shard_range(count, r, world)[0] for r in range(world)]
        # [2][n_queries][k] int32 bit patterns: plane 0 ids, plane 1 scores
        self.pack = torch.empty((2, n_queries, k), dtype=torch.int32, device=device)
        self.all = torch.empty((world, 2, n_queries, k), dtype=torch.int32, device=device)
        self.merged = torch.empty((2, n_queries, k), dtype=torch.int32, device=device)

    def buffers(self):
        """(ids [n_queries*k], scores [n_queries*k]) device views for the local topk_batch."""
        return self.pack[0].view(-1), self.pack[1].view(-1).view(self.torch.float32)

    def exchange(self, largest: bool = True):
        if self.world == 1:
            self.all[0].copy_(self.pack)
        elif _needs_host_staging(self.dist, self.pack):
            hp = self.pack.cpu()
            ha = self.torch.empty(self.all.shape, dtype=self.torch.int32)
            self.dist.all_gather_into_tensor(ha.view(-1), hp.view(-1), group=self.group)
            self.all.copy_(ha)
        else:
            self.dist.all_gather_into_tensor(self.all.view(-1), self.pack.view(-1), group=self.group)
        if self.all.is_cuda:
            return _device_merge(self.torch, self.all, self.nq, self.k, self.bases, largest, self.merged)
        host = self.all.numpy()
        ids = host[:, 0].view(np.uint32)   # [world, nq, k]
        sc = host[:, 1].view(np.float32)
        # all queries at once: [nq, world*k] candidates per query, one lexsort along the rows
        valid = (ids != 0xFFFFFFFF).transpose(1, 0, 2).reshape(self.nq, -1)
        gids = (ids.astype(np.int64) + np.asarray(self.bases, dtype=np.int64)[:, None, None])
        gids = gids.transpose(1, 0, 2).reshape(self.nq, -1)
        s2 = sc.transpose(1, 0, 2).reshape(self.nq, -1)
        key = np.where(valid, -s2 if largest else s2, np.inf).astype(np.float64)  # padding sorts last
        gkey = np.where(valid, gids, np.iinfo(np.int64).max)
        order = np.lexsort((gkey, key), axis=1)[:, : self.k]
        out_ids = np.take_along_axis(gids, order, axis=1).astype(np.uint32)
        out_sc = np.take_along_axis(s2, order, axis=1).astype(np.float32)
        ok = np.take_along_axis(valid, order, axis=1)
        out_ids[~ok] = 0xFFFFFFFF
        out_sc[~ok] = -np.inf if largest else np.inf
        return out_ids, out_sc
