"""Row-sharded stores behind ONE handle, one process driving several GPUs — the host-side mirror of
the `qamd_*_sharded_*` entry points (include/quantization_amd.h, csrc/sharded.hip).

The reference's caller is a single process (demos/src/ann_benchmark.rs:245-260); this is what it
would hold instead of one `EncodedVectors*`: shard g of G owns rows [g*N/G, (g+1)*N/G) on
`devices[g]`, global row id = shard base + local id, every result equals the single-handle result
bit for bit.  `devices` may repeat a device (logical shards on one GPU).  The per-process layer over
`torch.distributed` (RCCL) lives in `sharded.py`.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .encoded_vectors import (VectorParameters, check, device_of, flatten_rows, in_buf, make_stop, out_buf,
                              stream_ptr, validate)
from .encoded_vectors_binary import BitsStoreType, EncodedVectorsBin
from .encoded_vectors_pq import CENTROIDS_COUNT, EncodedVectorsPQ
from .encoded_vectors_u8 import EncodedVectorsU8


def _devices(devices):
    devs = [int(d) for d in devices]
    if not devs:
        raise ValueError("devices must name at least one GPU")
    return (C.c_int32 * len(devs))(*devs), devs


def _caller_stream(stream, *buffers) -> C.c_void_p:
    """The stream argument of the sharded entry points: the stream that produced the call's device
    inputs / still uses the buffers its device outputs overwrite.  Given explicitly, or torch's current
    stream on the device of the first CUDA tensor among `buffers` (a sharded call's tensors may live
    on any GPU, not only the current one); the null stream otherwise."""
    if stream is not None:
        return stream_ptr(stream)
    for b in buffers:
        d = device_of(b) if b is not None else None
        if d is not None:
            import torch
            return C.c_void_p(torch.cuda.current_stream(d).cuda_stream)
    return C.c_void_p(0)


class _ShardedQuery:
    def __init__(self, handle, free, owner=None):
        self._h, self._free = handle, free
        self._owner = owner  # the per-shard query objects live on the store's devices: free them first

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self._free(self._h)
            except Exception:  # interpreter shutdown
                pass
        self._h = None


class _ShardedBase:
    _prefix = ""

    def __init__(self, handle, devices, vector_parameters: VectorParameters):
        self._h = handle
        self.devices = list(devices)
        self._vp = vector_parameters

    def _fn(self, name):
        return getattr(_lib.lib(), f"qamd_{self._prefix}_sharded_{name}")

    @property
    def vector_parameters(self) -> VectorParameters:
        return self._vp

    @property
    def count(self) -> int:
        return self._vp.count

    @property
    def n_shards(self) -> int:
        return int(self._fn("shard_count")(self._h))

    def shard_range(self, g: int) -> tuple[int, int]:
        n, G = self.count, self.n_shards
        return (g * n) // G, ((g + 1) * n) // G

    def _shard_raw(self, g: int):
        h, base, dev = C.c_void_p(), C.c_uint64(), C.c_int32()
        check(self._fn("shard")(self._h, int(g), C.byref(h), C.byref(base), C.byref(dev)))
        return h, int(base.value), int(dev.value)

    PEER_STATES = ("same device", "enabled", "unavailable", "failed")

    def peer_access(self, g: int) -> tuple[str, str]:
        """(state, reason) of the route between shard g's device and devices[0]: direct peer copies where
        hipDeviceEnablePeerAccess succeeded, else the runtime stages the copies through host memory - recorded, never
        silently ignored (include/quantization_amd.h qamd_peer_state)."""
        st, why = C.c_int32(), C.c_char_p()
        check(self._fn("peer_access")(self._h, int(g), C.byref(st), C.byref(why)))
        return self.PEER_STATES[st.value], (why.value or b"").decode()

    def _check_root(self, *buffers):
        for b in buffers:
            d = device_of(b)
            if d is not None and d != self.devices[0]:
                raise ValueError(f"device outputs of a sharded store must live on devices[0] (cuda:{self.devices[0]})")

    def encode_query(self, query, reuse=None, stream=None):
        buf = in_buf(query, np.float32)
        n = int(np.prod(tuple(query.shape))) if hasattr(query, "shape") else len(query)
        h = reuse._h if reuse is not None else C.c_void_p()
        check(self._fn("encode_query")(self._h, buf.ptr, n, buf.mem, _caller_stream(stream, query), C.byref(h)))
        return reuse if reuse is not None else _ShardedQuery(h, self._fn("query_free"), self)

    def score_all(self, query, out=None, stream=None):
        """scores[i] = score_point(query, i) over the GLOBAL row ids — each shard writes its slice."""
        buf, ret = out_buf(out, self.count, np.float32)
        check(self._fn("score_all")(self._h, query._h, buf.ptr, buf.mem, _caller_stream(stream, out)))
        return ret

    def topk(self, query, k: int, largest: bool = True, out_ids=None, out_scores=None, stream=None):
        """Global best-k (ids are global row ids), merged on devices[0]; same order as one handle."""
        self._check_root(out_ids, out_scores)
        ib, ids = out_buf(out_ids, k, np.uint32)
        sb, sc = out_buf(out_scores, k, np.float32)
        if ib.mem != sb.mem:
            raise ValueError("out_ids and out_scores must both be host or both be device buffers")
        check(self._fn("topk")(self._h, query._h, int(k), int(bool(largest)), ib.ptr, sb.ptr, sb.mem,
                               _caller_stream(stream, out_ids, out_scores)))
        return ids, sc

    def encode_query_batch(self, queries, reuse=None, stream=None):
        nq, qdim = int(queries.shape[0]), int(queries.shape[1])
        buf = in_buf(queries, np.float32)
        h = reuse._h if reuse is not None else C.c_void_p()
        check(self._fn("encode_query_batch")(self._h, buf.ptr, nq, qdim, buf.mem, _caller_stream(stream, queries),
                                             C.byref(h)))
        if reuse is not None:
            reuse.n_queries = nq
            return reuse
        b = _ShardedQuery(h, self._fn("query_batch_free"), self)
        b.n_queries = nq
        return b

    def topk_batch(self, batch, k: int, largest: bool = True, out_ids=None, out_scores=None, stream=None):
        nq = batch.n_queries
        self._check_root(out_ids, out_scores)
        ib, ids = out_buf(out_ids, nq * k, np.uint32)
        sb, sc = out_buf(out_scores, nq * k, np.float32)
        if ib.mem != sb.mem:
            raise ValueError("out_ids and out_scores must both be host or both be device buffers")
        check(self._fn("topk_batch")(self._h, batch._h, int(k), int(bool(largest)), ib.ptr, sb.ptr, sb.mem,
                                     _caller_stream(stream, out_ids, out_scores)))
        if isinstance(ids, np.ndarray):
            return ids.reshape(nq, k), sc.reshape(nq, k)
        return ids, sc

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                self._fn("free")(self._h)
            except Exception:
                pass
        self._h = None


class ShardedVectorsU8(_ShardedBase):
    _prefix = "u8"

    @classmethod
    def encode(cls, orig_data, vector_parameters: VectorParameters, devices, quantile: float | None = None,
               stop_condition=None, *, alpha_offset=None, stream=None) -> "ShardedVectorsU8":
        data = flatten_rows(orig_data, vector_parameters.dim)
        validate(data, vector_parameters)
        vp = vector_parameters.to_c()
        buf = in_buf(data, np.float32)
        q = C.c_float(quantile) if quantile is not None else None
        ao = (C.c_float * 2)(*alpha_offset) if alpha_offset is not None else None
        arr, devs = _devices(devices)
        out = C.c_void_p()
        check(_lib.lib().qamd_u8_sharded_encode(buf.ptr, buf.mem, C.byref(vp), C.byref(q) if q is not None else None,
                                                C.cast(ao, C.POINTER(C.c_float)) if ao is not None else None,
                                                make_stop(stop_condition), None, arr, len(devs),
                                                _caller_stream(stream, data), C.byref(out)))
        return cls(out, devs, vector_parameters)

    @classmethod
    def from_storage(cls, rows, metadata: dict, devices, stream=None) -> "ShardedVectorsU8":
        vp = metadata["vector_parameters"]
        meta = _lib.U8MetadataC(int(metadata["actual_dim"]), float(metadata["alpha"]), float(metadata["offset"]),
                                float(metadata["multiplier"]), vp.to_c())
        buf = in_buf(rows, np.uint8)
        arr, devs = _devices(devices)
        out = C.c_void_p()
        check(_lib.lib().qamd_u8_sharded_from_rows(buf.ptr, buf.mem, C.byref(meta), arr, len(devs),
                                                   _caller_stream(stream, rows), C.byref(out)))
        return cls(out, devs, vp)

    @property
    def metadata(self) -> dict:
        m = _lib.U8MetadataC()
        check(_lib.lib().qamd_u8_sharded_get_metadata(self._h, C.byref(m)))
        return {"actual_dim": int(m.actual_dim), "alpha": np.float32(m.alpha), "offset": np.float32(m.offset),
                "multiplier": np.float32(m.multiplier),
                "vector_parameters": VectorParameters.from_c(m.vector_parameters)}

    def shard(self, g: int) -> tuple[EncodedVectorsU8, int]:
        """(borrowed single-device view of shard g, its first global row).  Valid while `self` lives."""
        h, base, dev = self._shard_raw(g)
        view = EncodedVectorsU8(h, dev, owned=False)
        view._owner = self  # the shard's handle is freed with the sharded store: keep that alive
        return view, base



class ShardedVectorsBin(_ShardedBase):
    _prefix = "bin"

    def __init__(self, handle, devices, vector_parameters, store):
        super().__init__(handle, devices, vector_parameters)
        self._store = BitsStoreType(store)

    @classmethod
    def encode(cls, orig_data, vector_parameters: VectorParameters, devices, stop_condition=None, *,
               store: BitsStoreType = BitsStoreType.U8, stream=None) -> "ShardedVectorsBin":
        data = flatten_rows(orig_data, vector_parameters.dim)
        validate(data, vector_parameters)
        vp = vector_parameters.to_c()
        buf = in_buf(data, np.float32)
        arr, devs = _devices(devices)
        out = C.c_void_p()
        check(_lib.lib().qamd_bin_sharded_encode(buf.ptr, buf.mem, C.byref(vp), int(store), make_stop(stop_condition),
                                                 None, arr, len(devs), _caller_stream(stream, data), C.byref(out)))
        return cls(out, devs, vector_parameters, store)

    @classmethod
    def from_storage(cls, rows, vector_parameters: VectorParameters, devices,
                     store: BitsStoreType = BitsStoreType.U8, stream=None) -> "ShardedVectorsBin":
        vp = vector_parameters.to_c()
        buf = in_buf(rows, np.uint8)
        arr, devs = _devices(devices)
        out = C.c_void_p()
        check(_lib.lib().qamd_bin_sharded_from_rows(buf.ptr, buf.mem, C.byref(vp), int(store), arr, len(devs),
                                                    _caller_stream(stream, rows), C.byref(out)))
        return cls(out, devs, vector_parameters, store)

    def shard(self, g: int) -> tuple[EncodedVectorsBin, int]:
        h, base, dev = self._shard_raw(g)
        b, e = self.shard_range(g)
        vp = VectorParameters(self._vp.dim, e - b, self._vp.distance_type, self._vp.invert)
        view = EncodedVectorsBin(h, vp, self._store, dev, owned=False)
        view._owner = self
        return view, base


class ShardedVectorsPQ(_ShardedBase):
    _prefix = "pq"

    def __init__(self, handle, devices, vector_parameters, chunk_size):
        super().__init__(handle, devices, vector_parameters)
        self._chunk_size = int(chunk_size)

    @classmethod
    def encode(cls, data, vector_parameters: VectorParameters, chunk_size: int, devices, max_kmeans_threads: int = 1,
               stop_condition=None, *, centroids=None, stream=None) -> "ShardedVectorsPQ":
        data = flatten_rows(data, vector_parameters.dim)
        validate(data, vector_parameters)
        vp = vector_parameters.to_c()
        buf = in_buf(data, np.float32)
        cen = None
        if centroids is not None:
            cen = np.ascontiguousarray(centroids, dtype=np.float32)
            if cen.shape != (CENTROIDS_COUNT, vector_parameters.dim):
                raise ValueError("centroids must be [256, dim]")
        arr, devs = _devices(devices)
        out = C.c_void_p()
        check(_lib.lib().qamd_pq_sharded_encode(buf.ptr, buf.mem, C.byref(vp), int(chunk_size),
                                                C.c_void_p(cen.ctypes.data) if cen is not None else None,
                                                int(max_kmeans_threads), make_stop(stop_condition), None, arr, len(devs),
                                                _caller_stream(stream, data), C.byref(out)))
        return cls(out, devs, vector_parameters, chunk_size)

    @classmethod
    def from_storage(cls, rows, vector_parameters: VectorParameters, chunk_size: int, centroids,
                     devices, stream=None) -> "ShardedVectorsPQ":
        vp = vector_parameters.to_c()
        buf = in_buf(rows, np.uint8)
        cen = np.ascontiguousarray(centroids, dtype=np.float32)
        arr, devs = _devices(devices)
        out = C.c_void_p()
        check(_lib.lib().qamd_pq_sharded_from_rows(buf.ptr, buf.mem, C.byref(vp), int(chunk_size),
                                                   C.c_void_p(cen.ctypes.data), arr, len(devs),
                                                   _caller_stream(stream, rows), C.byref(out)))
        return cls(out, devs, vector_parameters, chunk_size)

    @property
    def centroids(self) -> np.ndarray:
        cen = np.zeros((CENTROIDS_COUNT, self._vp.dim), dtype=np.float32)
        check(_lib.lib().qamd_pq_sharded_get_centroids(self._h, C.c_void_p(cen.ctypes.data)))
        return cen

    def shard(self, g: int) -> tuple[EncodedVectorsPQ, int]:
        h, base, dev = self._shard_raw(g)
        b, e = self.shard_range(g)
        vp = VectorParameters(self._vp.dim, e - b, self._vp.distance_type, self._vp.invert)
        view = EncodedVectorsPQ(h, vp, self._chunk_size, dev, owned=False)
        view._owner = self
        return view, base
