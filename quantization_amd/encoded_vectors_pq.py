"""EncodedVectorsPQ — host-side mirror of quantization/src/encoded_vectors_pq.rs."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._base import EncodedQueryBase, EncodedVectorsBase
from .encoded_vectors import (DistanceType, EncodingError, VectorParameters, check, check_same_device,
                              creating_on, flatten_rows, get_device, in_buf, make_stop, out_buf, stream_ptr,
                              validate)

KMEANS_SAMPLE_SIZE = 10_000   # :22
KMEANS_MAX_ITERATIONS = 100   # :23
KMEANS_ACCURACY = 1e-5        # :24
CENTROIDS_COUNT = 256         # :25


class EncodedQueryPQ(EncodedQueryBase):
    """encoded_vectors_pq.rs:35-37 — the chunk-major lookup table."""

    _prefix = "pq"

    @property
    def lut(self) -> np.ndarray:
        n = C.c_uint64()
        check(_lib.lib().qamd_pq_query_read(self._h, None, 0, C.byref(n)))
        lut = np.zeros(n.value, dtype=np.float32)
        if n.value:
            check(_lib.lib().qamd_pq_query_read(self._h, C.c_void_p(lut.ctypes.data), n.value, None))
        return lut


class EncodedVectorsPQ(EncodedVectorsBase):
    _prefix = "pq"
    _query_cls = EncodedQueryPQ

    def __init__(self, handle, vector_parameters: VectorParameters, chunk_size: int, device=None,
                 owned: bool = True):
        super().__init__(handle, device, owned)
        self._vp = vector_parameters
        self._chunk_size = int(chunk_size)

    @property
    def vector_parameters(self) -> VectorParameters:
        return self._vp

    @property
    def vector_division(self) -> list[range]:
        """get_vector_division (:116-121)."""
        d, c = self._vp.dim, self._chunk_size
        return [range(i, min(i + c, d)) for i in range(0, d, c)]

    @property
    def centroids(self) -> np.ndarray:
        """Metadata.centroids (:39-44): [256, dim] f32, centroid-major."""
        cen = np.zeros((CENTROIDS_COUNT, self._vp.dim), dtype=np.float32)
        check(_lib.lib().qamd_pq_get_centroids(self._h, C.c_void_p(cen.ctypes.data)))
        return cen

    @property
    def metadata(self) -> dict:
        return {"centroids": self.centroids, "vector_division": self.vector_division,
                "vector_parameters": self._vp}

    @classmethod
    def encode(cls, data, vector_parameters: VectorParameters, chunk_size: int, max_kmeans_threads: int = 1,
               stop_condition=None, *, centroids=None, stream=None) -> "EncodedVectorsPQ":
        """EncodedVectorsPQ::encode (:56-107).  `centroids` (256 x dim) skips find_centroids.
        Without them k-means runs on the evenly strided <= 10 000-row sample; `max_kmeans_threads`
        fixes the order of the f64 partial sums exactly as the reference's worker count does
        (kmeans.rs:77-107)."""
        data = flatten_rows(data, vector_parameters.dim)
        validate(data, vector_parameters)
        vp = vector_parameters.to_c()
        buf = in_buf(data, np.float32)
        cen = None
        if centroids is not None:
            cen = np.ascontiguousarray(centroids, dtype=np.float32)
            if cen.shape != (CENTROIDS_COUNT, vector_parameters.dim):
                raise ValueError("centroids must be [256, dim]")
        stop = make_stop(stop_condition)
        out = C.c_void_p()
        with creating_on(data) as dev:
            check(_lib.lib().qamd_pq_encode(buf.ptr, buf.mem, C.byref(vp), int(chunk_size),
                                            C.c_void_p(cen.ctypes.data) if cen is not None else None,
                                            int(max_kmeans_threads), stop, None, stream_ptr(stream), C.byref(out)))
        return cls(out, vector_parameters, chunk_size, dev)

    @classmethod
    def encode_stream(cls, make_batches, vector_parameters: VectorParameters, chunk_size: int,
                      max_kmeans_threads: int = 1, stop_condition=None, *, centroids=None,
                      stream=None) -> "EncodedVectorsPQ":
        """encode from the reference's clonable-iterator contract (:56-107 walks it twice:
        find_centroids, then encode_storage).  `make_batches()` returns a fresh iterator over
        [n_i, dim] f32 batches each call.  Byte-identical to `encode` on the concatenated batches."""
        L = _lib.lib()
        vp = vector_parameters.to_c()
        cen = None
        if centroids is not None:
            cen = np.ascontiguousarray(centroids, dtype=np.float32)
            if cen.shape != (CENTROIDS_COUNT, vector_parameters.dim):
                raise ValueError("centroids must be [256, dim]")
        stop = make_stop(stop_condition)
        first = next(iter(make_batches()), None)
        enc = C.c_void_p()
        with creating_on(first) as dev:
            check(L.qamd_pq_encoder_begin(C.byref(vp), int(chunk_size),
                                          C.c_void_p(cen.ctypes.data) if cen is not None else None,
                                          int(max_kmeans_threads), stop, None, stream_ptr(stream), C.byref(enc)))
        try:
            for fn in ((L.qamd_pq_encoder_observe,) if cen is None else ()) + (L.qamd_pq_encoder_push,):
                for batch in make_batches():
                    if len(batch.shape) != 2 or (batch.shape[0] and batch.shape[1] != vector_parameters.dim):
                        raise EncodingError(_lib.ERR_ARGUMENTS, f"Vector length {batch.shape[-1]} does not match "
                                                                f"vector parameters dim {vector_parameters.dim}")
                    check_same_device(dev, batch)
                    buf = in_buf(batch, np.float32)
                    check(fn(enc, buf.ptr, int(batch.shape[0]), buf.mem))
            out = C.c_void_p()
            h, enc = enc, None
            check(L.qamd_pq_encoder_finish(h, C.byref(out)))
        finally:
            if enc is not None:
                L.qamd_pq_encoder_abort(enc)
        return cls(out, vector_parameters, chunk_size, dev)

    def kmeans_info(self) -> tuple[int, int]:
        """(iterations of the slowest chunk, empty-cluster re-seeds) of the training that built this
        store; (0, 0) when centroids were given."""
        it, em = C.c_uint32(), C.c_uint32()
        check(_lib.lib().qamd_pq_kmeans_info(self._h, C.byref(it), C.byref(em)))
        return int(it.value), int(em.value)

    @staticmethod
    def find_centroids(rows, chunk_size: int, max_kmeans_threads: int = 1, stop_condition=None, stream=None) -> np.ndarray:
        """find_centroids (encoded_vectors_pq.rs:278-342) on its own: the [256, dim] centroids `encode` would train for
        `rows` ([n, dim] f32, host or HBM) - for callers that hold the data in several places
        (quantization_amd.sharded.encode_pq gathers the k-means sample rows to one rank and broadcasts the result)."""
        n, dim = int(rows.shape[0]), int(rows.shape[1])
        vp = VectorParameters(dim, n, DistanceType.Dot, False).to_c()  # (the training does not depend on the metric)
        buf = in_buf(rows, np.float32)
        cen = np.empty((256, dim), dtype=np.float32)
        stop = make_stop(stop_condition)
        with creating_on(rows):
            check(_lib.lib().qamd_pq_find_centroids(buf.ptr, buf.mem, C.byref(vp), int(chunk_size), int(max_kmeans_threads),
                                                    stop, None, stream_ptr(stream), C.c_void_p(cen.ctypes.data), None, None))
        return cen

    def scan_kernel(self) -> tuple[str, int]:
        """(name of the whole-store scan kernel this store takes, launches per scan) - for measurement harnesses."""
        n = C.c_uint32()
        name = _lib.lib().qamd_pq_scan_kernel(self._h, C.byref(n))
        return name.decode(), int(n.value)

    @classmethod
    def from_storage(cls, rows, vector_parameters: VectorParameters, chunk_size: int, centroids,
                     stream=None) -> "EncodedVectorsPQ":
        vp = vector_parameters.to_c()
        buf = in_buf(rows, np.uint8)
        cen = np.ascontiguousarray(centroids, dtype=np.float32)
        out = C.c_void_p()
        with creating_on(rows) as dev:
            check(_lib.lib().qamd_pq_from_rows(buf.ptr, buf.mem, C.byref(vp), int(chunk_size),
                                               C.c_void_p(cen.ctypes.data), stream_ptr(stream), C.byref(out)))
        return cls(out, vector_parameters, chunk_size, dev)

    @classmethod
    def load(cls, data_path, meta_path, vector_parameters: VectorParameters) -> "EncodedVectorsPQ":
        """EncodedVectors::load (:508-523)."""
        vp = vector_parameters.to_c()
        out = C.c_void_p()
        check(_lib.lib().qamd_pq_load(os.fsencode(data_path), os.fsencode(meta_path), C.byref(vp), C.byref(out)))
        import json
        meta = json.load(open(meta_path))
        m = meta["vector_parameters"]
        div = meta["vector_division"]
        chunk = (div[0]["end"] - div[0]["start"]) if div else 1
        eff = VectorParameters(int(m["dim"]), vector_parameters.count, DistanceType[m["distance_type"]],
                               bool(m["invert"]))
        return cls(out, eff, chunk, get_device())

    def save(self, data_path, meta_path) -> None:
        """EncodedVectors::save (:498-506)."""
        check(_lib.lib().qamd_pq_save(self._h, os.fsencode(data_path), os.fsencode(meta_path)))

    @staticmethod
    def get_quantized_vector_size(vector_parameters: VectorParameters, chunk_size: int) -> int:
        """:109-114."""
        vp = vector_parameters.to_c()
        return int(_lib.lib().qamd_pq_quantized_vector_size(C.byref(vp), int(chunk_size)))

    def storage_bytes(self, out=None, stream=None):
        n = self._vp.count
        m = self.get_quantized_vector_size(self._vp, self._chunk_size)
        check_same_device(self._device, out)
        buf, ret = out_buf(out, n * m, np.uint8)
        check(_lib.lib().qamd_pq_export_rows(self._h, buf.ptr, buf.mem, stream_ptr(stream)))
        return ret.reshape(n, m) if isinstance(ret, np.ndarray) else ret

    def storage_rows(self, first_row: int, n_rows: int, out=None, stream=None):
        """Rows [first_row, first_row + n_rows) as push_vector_data would receive them (encoded_storage.rs:17-25)."""
        m = self.get_quantized_vector_size(self._vp, self._chunk_size)
        check_same_device(self._device, out)
        buf, ret = out_buf(out, n_rows * m, np.uint8)
        check(_lib.lib().qamd_pq_export_rows_range(self._h, int(first_row), int(n_rows), buf.ptr, buf.mem,
                                                   stream_ptr(stream)))
        return ret.reshape(n_rows, m) if isinstance(ret, np.ndarray) else ret
