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

`encode_u8` / `encode_pq` / `encode_binary` build such a sharded store from data that is ALREADY spread over the ranks
(rank g holds rows [g*N/G, (g+1)*N/G) of the f32 input): the reference's `encode` finds its global statistics over all
data before any row is quantised - ONE (alpha, offset) (encoded_vectors_u8.rs:57-71), ONE set of centroids
(encoded_vectors_pq.rs:278-342) - so the ranks agree on them first (two tiny collectives, or a gather of the <= 100 000
/ <= 10 000 sampled rows to rank 0 and a broadcast) and then every rank encodes its own rows.  Each rank's row bytes and
metadata equal the single-handle encode of the concatenated data.

The scoring and the local encoding are any object with the EncodedVectors API of this package; this module only does
the index arithmetic and the collectives, so the CPU tests drive it with stand-in (oracle-backed) operations.
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


# ----------------------------------------------------------------------------------------------- distributed encode
QUANTILE_SAMPLE_SIZE = 100_000  # quantile.rs:3
KMEANS_SAMPLE_SIZE = 10_000     # encoded_vectors_pq.rs:22


def sample_rows_of_shard(count: int, sample_size: int, begin: int, end: int) -> np.ndarray:
    """LOCAL indices (ascending) of the sampled rows that fall into shard [begin, end): the sample of `count` rows is the
    rows floor(k * count / S), k < S = min(sample_size, count) - every row when count <= sample_size - which is the rule of
    the single-handle encoders (include/quantization_amd.h: qamd_u8_find_quantile_interval, qamd_pq_find_centroids; the
    reference draws a random Permutor sample there)."""
    S = min(sample_size, count)
    if S == 0 or end <= begin:
        return np.zeros(0, dtype=np.int64)
    # smallest k with floor(k*count/S) >= begin is ceil(begin*S/count); the picks are ascending in k
    k0 = (begin * S + count - 1) // count
    k1 = min(S, (end * S + count - 1) // count)
    ks = np.arange(k0, k1, dtype=np.uint64)
    rows = (ks * np.uint64(count)) // np.uint64(S)
    return rows.astype(np.int64) - begin


def _f32_key(x) -> int:
    """Order-preserving integer key of an f32 (-0.0 < +0.0): the min / max fold is then an integer all-reduce, whose result
    does not depend on the backend's float min/max semantics."""
    b = int(np.float32(x).view(np.uint32))
    return b ^ 0xFFFFFFFF if b >> 31 else b | 0x80000000


def _key_f32(k: int) -> np.float32:
    b = (k & 0x7FFFFFFF) if k >> 31 else (k ^ 0xFFFFFFFF)
    return np.uint32(b).view(np.float32)


def alpha_offset_from_min_max(mn, mx) -> tuple[np.float32, np.float32]:
    """encoded_vectors_u8.rs:228-232, in f32."""
    with np.errstate(all="ignore"):
        return np.float32((np.float32(mx) - np.float32(mn)) / np.float32(127.0)), np.float32(mn)


class LibraryOps:
    """The per-rank operations of the distributed encode through the C ABI (the product path; GPU).  The CPU tests pass an
    object with the same methods backed by the oracle."""

    def find_min_max(self, rows):
        from .encoded_vectors_u8 import EncodedVectorsU8
        return EncodedVectorsU8.find_min_max(rows)

    def find_quantile_interval(self, rows, quantile):
        from .encoded_vectors_u8 import EncodedVectorsU8
        return EncodedVectorsU8.find_quantile_interval(rows, quantile)

    def find_centroids(self, rows, chunk_size, max_kmeans_threads):
        from .encoded_vectors_pq import EncodedVectorsPQ
        return EncodedVectorsPQ.find_centroids(rows, chunk_size, max_kmeans_threads)

    def encode_u8(self, rows, vp, alpha_offset):
        from .encoded_vectors_u8 import EncodedVectorsU8
        return EncodedVectorsU8.encode(rows, vp, alpha_offset=alpha_offset)

    def encode_pq(self, rows, vp, chunk_size, centroids):
        from .encoded_vectors_pq import EncodedVectorsPQ
        return EncodedVectorsPQ.encode(rows, vp, chunk_size, centroids=centroids)

    def encode_binary(self, rows, vp):
        from .encoded_vectors_binary import EncodedVectorsBin
        return EncodedVectorsBin.encode(rows, vp)


def _coll_device(dist, torch, rows):
    """Where the collectives' tensors live: HBM for nccl (= RCCL), host for gloo."""
    if dist.get_backend() == "gloo":
        return torch.device("cpu")
    if getattr(rows, "is_cuda", False):
        return rows.device
    return torch.device("cuda", torch.cuda.current_device())


def _local_vp(vp, n_local: int):
    return type(vp)(vp.dim, n_local, vp.distance_type, vp.invert)


def _gather_sample(dist, torch, local_rows, count: int, sample_size: int, dim: int, group=None):
    """The sampled rows of the whole data set, in sample order, on rank 0 (None elsewhere): every rank contributes the
    picks that fall into its shard (padded to the largest contribution: a gather wants equal sizes)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    picks = [sample_rows_of_shard(count, sample_size, *shard_range(count, r, world)) for r in range(world)]
    mine = picks[rank]
    dev = _coll_device(dist, torch, local_rows)
    if hasattr(local_rows, "is_cuda"):  # torch tensor
        part = local_rows.index_select(0, torch.from_numpy(mine).to(local_rows.device)).to(device=dev, dtype=torch.float32)
    else:
        part = torch.from_numpy(np.ascontiguousarray(np.asarray(local_rows, dtype=np.float32)[mine])).to(dev)
    part = part.reshape(len(mine), dim)
    pad = max(len(p) for p in picks)
    if world == 1:
        return part
    send = torch.zeros((pad, dim), dtype=torch.float32, device=dev)
    send[: len(mine)] = part
    glist = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, gather_list=glist, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([g[: len(p)] for g, p in zip(glist, picks)])


def encode_u8(dist, torch, local_rows, vector_parameters, quantile=None, *, ops=None, group=None):
    """EncodedVectorsU8::encode (encoded_vectors_u8.rs:34-140) over rows spread across the ranks: `local_rows` is this
    rank's shard [n_local, dim] (numpy or tensor, host or HBM) of the vector_parameters.count rows.  Returns
    (this rank's encoded shard, (alpha, offset)).

      pass 1   every rank: find_min_max over its rows (quantile.rs:5-19); one all-reduce of the two order-preserving
               integer keys (min of mins, max of maxes: order-free, so the bits of the single-handle encode)
      pass 1b  quantile given: the <= 100 000 sampled rows are gathered to rank 0, which runs find_quantile_interval
               (quantile.rs:21-71); the interval (or "None") is broadcast
      pass 2   every rank encodes its rows with the agreed (alpha, offset)."""
    ops = ops or LibraryOps()
    vp = vector_parameters
    world = dist.get_world_size(group)
    n_local = int(local_rows.shape[0])
    if vp.count == 0:  # encoded_vectors_u8.rs:43-54
        return ops.encode_u8(local_rows, _local_vp(vp, 0), None), (np.float32(0), np.float32(0))
    dev = _coll_device(dist, torch, local_rows)
    mn, mx = ops.find_min_max(local_rows)
    keys = torch.tensor([_f32_key(mn), -_f32_key(mx)], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)
    kmin, kmax = (int(v) for v in keys.cpu().tolist())
    alpha, offset = alpha_offset_from_min_max(_key_f32(kmin), _key_f32(-kmax))
    if quantile is not None and not (vp.count < 127 or quantile >= 1.0):  # quantile.rs:27-29
        sample = _gather_sample(dist, torch, local_rows, vp.count, QUANTILE_SAMPLE_SIZE, vp.dim, group)
        res = torch.zeros(3, dtype=torch.float32, device=dev)
        if dist.get_rank(group) == 0:
            found = ops.find_quantile_interval(sample, quantile)
            if found is not None:
                res = torch.tensor([1.0, float(found[0]), float(found[1])], dtype=torch.float32, device=dev)
        if world > 1:
            dist.broadcast(res, src=0, group=group)
        r = res.cpu().numpy()
        if r[0] != 0:
            alpha, offset = alpha_offset_from_min_max(r[1], r[2])
    enc = ops.encode_u8(local_rows, _local_vp(vp, n_local), (float(alpha), float(offset)))
    return enc, (alpha, offset)


def encode_pq(dist, torch, local_rows, vector_parameters, chunk_size: int, max_kmeans_threads: int = 1, *,
              centroids=None, ops=None, group=None):
    """EncodedVectorsPQ::encode (encoded_vectors_pq.rs:56-107) over rows spread across the ranks.  find_centroids
    (:278-342) needs the <= 10 000 sampled rows in one place: they are gathered to rank 0, which trains (k-means in the
    reference's summation order, csrc/pq.hip) and broadcasts the 256 x dim centroids; every rank then runs
    encode_storage (:136-226) on its own rows.  Returns (this rank's encoded shard, centroids)."""
    ops = ops or LibraryOps()
    vp = vector_parameters
    world = dist.get_world_size(group)
    dev = _coll_device(dist, torch, local_rows)
    if centroids is None:
        sample = _gather_sample(dist, torch, local_rows, vp.count, KMEANS_SAMPLE_SIZE, vp.dim, group)
        cen = torch.zeros((256, vp.dim), dtype=torch.float32, device=dev)
        if dist.get_rank(group) == 0:
            cen = torch.from_numpy(np.ascontiguousarray(ops.find_centroids(sample, chunk_size, max_kmeans_threads),
                                                        dtype=np.float32)).to(dev)
        if world > 1:
            dist.broadcast(cen, src=0, group=group)
        centroids = cen.cpu().numpy()
    enc = ops.encode_pq(local_rows, _local_vp(vp, int(local_rows.shape[0])), chunk_size, centroids)
    return enc, centroids


def encode_binary(dist, torch, local_rows, vector_parameters, *, ops=None, group=None):
    """EncodedVectorsBin::encode (encoded_vectors_binary.rs:165-191) has no global statistic: every rank packs its rows."""
    ops = ops or LibraryOps()
    return ops.encode_binary(local_rows, _local_vp(vector_parameters, int(local_rows.shape[0])))
