"""Scoring methods shared by the three quantizers: the C ABI gives them identical shapes
(`qamd_{u8,bin,pq}_score_*`), as `trait EncodedVectors` does in the reference
(quantization/src/encoded_vectors.rs:21-35)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .encoded_vectors import check, check_same_device, in_buf, out_buf, stream_ptr


class EncodedQueryBase:
    _prefix = ""

    def __init__(self, handle: C.c_void_p):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                getattr(_lib.lib(), f"qamd_{self._prefix}_query_free")(self._h)
            except Exception:  # interpreter shutdown: module globals already torn down
                pass
            self._h = None


class EncodedQueryBatch:
    """n encoded queries resident in HBM (n x EncodedQuery*), for the many-queries-at-once calls."""

    _prefix = ""

    def __init__(self, handle: C.c_void_p, n_queries: int, prefix: str | None = None):
        self._h = handle
        self.n_queries = n_queries
        if prefix is not None:
            self._prefix = prefix

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                getattr(_lib.lib(), f"qamd_{self._prefix}_query_batch_free")(self._h)
            except Exception:
                pass
        self._h = None


class EncodedVectorsBase:
    _prefix = ""
    _query_cls = EncodedQueryBase
    _batch_cls = EncodedQueryBatch

    def __init__(self, handle: C.c_void_p, device: int | None = None, owned: bool = True):
        self._h = handle
        self._device = device  # where the store lives; device buffers of every call must match
        self._owned = owned    # False: a shard borrowed from a sharded handle
        self._count = None

    def _fn(self, name):
        return getattr(_lib.lib(), f"qamd_{self._prefix}_{name}")

    @property
    def device(self) -> int | None:
        return self._device

    @property
    def count(self) -> int:
        if self._count is None:  # immutable after construction: one ABI round trip, not one per query
            self._count = int(self.vector_parameters.count)
        return self._count

    def encode_query(self, query, reuse=None, stream=None):
        """EncodedVectors::encode_query.  `reuse` recycles an encoded-query object (no
        allocation in a query loop)."""
        check_same_device(self._device, query)
        buf = in_buf(query, np.float32)
        n = int(np.prod(tuple(query.shape))) if hasattr(query, "shape") else len(query)
        h = reuse._h if reuse is not None else C.c_void_p()
        check(self._fn("encode_query")(self._h, buf.ptr, n, buf.mem, stream_ptr(stream), C.byref(h)))
        return reuse if reuse is not None else self._query_cls(h)

    def score_point(self, query, i: int) -> np.float32:
        """EncodedVectors::score_point — one (query, row) pair.  A kernel launch per call:
        use score_all / score_ids / topk on the hot path."""
        out = C.c_float()
        check(self._fn("score_point")(self._h, query._h, int(i), C.byref(out)))
        return np.float32(out.value)

    def score_internal(self, i: int, j: int) -> np.float32:
        """EncodedVectors::score_internal — rows i and j of the store."""
        out = C.c_float()
        check(self._fn("score_internal")(self._h, int(i), int(j), C.byref(out)))
        return np.float32(out.value)

    def score_all(self, query, out=None, stream=None):
        """scores[i] = score_point(query, i) for every row: the batched form of the caller
        loop in demos/src/ann_benchmark.rs:247-252."""
        check_same_device(self._device, out)
        buf, ret = out_buf(out, self.count, np.float32)
        check(self._fn("score_all")(self._h, query._h, buf.ptr, buf.mem, stream_ptr(stream)))
        return ret

    def score_ids(self, query, ids, out=None, stream=None):
        """scores[k] = score_point(query, ids[k]) (random access, demos/benches/encode.rs)."""
        check_same_device(self._device, ids, out)
        ib = in_buf(ids, np.uint32)
        n = int(ids.numel()) if hasattr(ids, "numel") else len(ids)
        buf, ret = out_buf(out, n, np.float32)
        check(self._fn("score_ids")(self._h, query._h, ib.ptr, n, ib.mem, buf.ptr, buf.mem,
                                    stream_ptr(stream)))
        return ret

    # ------------------------------------------------------------------ bursts of pairs (the HNSW caller)
    def score_internal_ids(self, i: int, ids, out=None, stream=None):
        """scores[k] = score_internal(i, ids[k]): one stored row against many, one launch."""
        check_same_device(self._device, ids, out)
        ib = in_buf(ids, np.uint32)
        n = int(ids.numel()) if hasattr(ids, "numel") else len(ids)
        buf, ret = out_buf(out, n, np.float32)
        check(self._fn("score_internal_ids")(self._h, int(i), ib.ptr, n, ib.mem, buf.ptr, buf.mem, stream_ptr(stream)))
        return ret

    def _lists(self, list_offsets, ids, rows=None):
        bufs = [in_buf(list_offsets, np.uint32), in_buf(ids, np.uint32)] + ([in_buf(rows, np.uint32)] if rows is not None else [])
        if len({b.mem for b in bufs}) != 1:
            raise ValueError("list_offsets, ids and rows must all be host or all be device buffers")
        count = lambda x: int(x.numel()) if hasattr(x, "numel") else len(x)
        n_lists, n_ids = count(list_offsets) - 1, count(ids)
        if rows is not None and count(rows) != n_lists:
            raise ValueError("one row per list")
        return bufs, n_lists, n_ids

    def score_ids_batch(self, batch, list_offsets, ids, out=None, stream=None):
        """One launch for many (query, id list) pairs - one hop of every in-flight HNSW search: list l =
        ids[list_offsets[l]:list_offsets[l + 1]] is scored against query l of `batch`;
        scores[p] = score_point(query l, ids[p])."""
        check_same_device(self._device, list_offsets, ids, out)
        (ob, ib), n_lists, n_ids = self._lists(list_offsets, ids)
        buf, ret = out_buf(out, n_ids, np.float32)
        check(self._fn("score_ids_batch")(self._h, batch._h, ob.ptr, n_lists, ib.ptr, n_ids, ib.mem, buf.ptr, buf.mem,
                                          stream_ptr(stream)))
        return ret

    def score_internal_ids_batch(self, rows, list_offsets, ids, out=None, stream=None):
        """One launch for many (stored row, id list) pairs - graph construction:
        scores[p] = score_internal(rows[l], ids[p]) for the ids p of list l."""
        check_same_device(self._device, rows, list_offsets, ids, out)
        (ob, ib, rb), n_lists, n_ids = self._lists(list_offsets, ids, rows)
        buf, ret = out_buf(out, n_ids, np.float32)
        check(self._fn("score_internal_ids_batch")(self._h, rb.ptr, ob.ptr, n_lists, ib.ptr, n_ids, ib.mem, buf.ptr,
                                                   buf.mem, stream_ptr(stream)))
        return ret

    def topk(self, query, k: int, largest: bool = True, out_ids=None, out_scores=None, stream=None):
        """Best-k rows of the scan (demos/src/ann_benchmark_data.rs:151-167 keeps 30 in a heap),
        sorted best-first; ties go to the lower row id.  Returns (ids, scores)."""
        check_same_device(self._device, out_ids, out_scores)
        ib, ids = out_buf(out_ids, k, np.uint32)
        sb, sc = out_buf(out_scores, k, np.float32)
        if ib.mem != sb.mem:
            raise ValueError("out_ids and out_scores must both be host or both be device buffers")
        check(self._fn("topk")(self._h, query._h, int(k), int(bool(largest)), ib.ptr, sb.ptr, sb.mem,
                               stream_ptr(stream)))
        return ids, sc

    # ------------------------------------------------------------------ many queries at once
    def encode_query_batch(self, queries, reuse=None, stream=None):
        """encode_query for a [n_queries, dim] block of queries (the caller's outer loop,
        demos/src/ann_benchmark.rs:245-260)."""
        nq, qdim = int(queries.shape[0]), int(queries.shape[1])
        check_same_device(self._device, queries)
        buf = in_buf(queries, np.float32)
        h = reuse._h if reuse is not None else C.c_void_p()
        check(self._fn("encode_query_batch")(self._h, buf.ptr, nq, qdim, buf.mem, stream_ptr(stream), C.byref(h)))
        if reuse is not None:
            reuse.n_queries = nq
            return reuse
        return self._batch_cls(h, nq, self._prefix)

    def score_batch(self, batch, out=None, stream=None):
        """scores[q, i] = score_point(query q, i) — bit-identical to score_all per query."""
        n = self.count
        check_same_device(self._device, out)
        buf, ret = out_buf(out, batch.n_queries * n, np.float32)
        check(self._fn("score_batch")(self._h, batch._h, buf.ptr, buf.mem, stream_ptr(stream)))
        return ret.reshape(batch.n_queries, n) if isinstance(ret, np.ndarray) else ret

    def topk_batch(self, batch, k: int, largest: bool = True, out_ids=None, out_scores=None, stream=None):
        """Per-query best-k (ann_benchmark_data.rs:151-167), [n_queries, k] ids and scores."""
        nq = batch.n_queries
        check_same_device(self._device, out_ids, out_scores)
        ib, ids = out_buf(out_ids, nq * k, np.uint32)
        sb, sc = out_buf(out_scores, nq * k, np.float32)
        if ib.mem != sb.mem:
            raise ValueError("out_ids and out_scores must both be host or both be device buffers")
        check(self._fn("topk_batch")(self._h, batch._h, int(k), int(bool(largest)), ib.ptr, sb.ptr, sb.mem,
                                     stream_ptr(stream)))
        if isinstance(ids, np.ndarray):
            return ids.reshape(nq, k), sc.reshape(nq, k)
        return ids, sc

    def __del__(self):
        if getattr(self, "_h", None) and getattr(self, "_owned", True):
            try:
                self._fn("free")(self._h)
            except Exception:  # interpreter shutdown
                pass
        self._h = None
