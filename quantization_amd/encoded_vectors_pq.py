"""EncodedVectorsPQ — host-side mirror of quantization/src/encoded_vectors_pq.rs."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._base import EncodedQueryBase, EncodedVectorsBase
from .encoded_vectors import (DistanceType, VectorParameters, check, flatten_rows, in_buf, make_stop,
                              out_buf, stream_ptr, validate)

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

    def __init__(self, handle, vector_parameters: VectorParameters, chunk_size: int):
        super().__init__(handle)
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
        """EncodedVectorsPQ::encode (:56-107).  `centroids` (256 x dim) skips find_centroids —
        the conditional-parity form, since the reference's k-means is randomised."""
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
        check(_lib.lib().qamd_pq_encode(buf.ptr, buf.mem, C.byref(vp), int(chunk_size),
                                        C.c_void_p(cen.ctypes.data) if cen is not None else None,
                                        int(max_kmeans_threads), stop, None, stream_ptr(stream), C.byref(out)))
        return cls(out, vector_parameters, chunk_size)

    @classmethod
    def from_storage(cls, rows, vector_parameters: VectorParameters, chunk_size: int, centroids,
                     stream=None) -> "EncodedVectorsPQ":
        vp = vector_parameters.to_c()
        buf = in_buf(rows, np.uint8)
        cen = np.ascontiguousarray(centroids, dtype=np.float32)
        out = C.c_void_p()
        check(_lib.lib().qamd_pq_from_rows(buf.ptr, buf.mem, C.byref(vp), int(chunk_size),
                                           C.c_void_p(cen.ctypes.data), stream_ptr(stream), C.byref(out)))
        return cls(out, vector_parameters, chunk_size)

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
        return cls(out, eff, chunk)

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
        buf, ret = out_buf(out, n * m, np.uint8)
        check(_lib.lib().qamd_pq_export_rows(self._h, buf.ptr, buf.mem, stream_ptr(stream)))
        return ret.reshape(n, m) if isinstance(ret, np.ndarray) else ret
