"""EncodedVectorsU8 — host-side mirror of quantization/src/encoded_vectors_u8.rs over the C ABI.

Same method names, argument meaning and error behaviour as the reference; the batched
`score_all` / `score_ids` / `topk` replace the caller loop
`for i in 0..n { score_point(q, i) }` (demos/src/ann_benchmark.rs:247-252).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._base import EncodedQueryBase, EncodedQueryBatch, EncodedVectorsBase
from .encoded_vectors import (DistanceType, EncodingError, VectorParameters, check, check_same_device,
                              creating_on, flatten_rows, get_device, in_buf, make_stop, out_buf, stream_ptr,
                              validate)

ALIGNMENT = 16  # encoded_vectors_u8.rs:12


class EncodedQueryU8(EncodedQueryBase):
    """encoded_vectors_u8.rs:19-22 — {offset, encoded_query}; the bytes live in HBM."""

    _prefix = "u8"

    @property
    def offset(self) -> np.float32:
        off = C.c_float()
        check(_lib.lib().qamd_u8_query_read(self._h, C.byref(off), None, 0, None))
        return np.float32(off.value)

    @property
    def encoded_query(self) -> np.ndarray:
        n = C.c_uint64()
        check(_lib.lib().qamd_u8_query_read(self._h, None, None, 0, C.byref(n)))
        codes = np.empty(n.value, dtype=np.uint8)
        check(_lib.lib().qamd_u8_query_read(self._h, None, C.c_void_p(codes.ctypes.data), n.value, None))
        return codes


class EncodedQueryBatchU8(EncodedQueryBatch):
    """n encoded queries (n x EncodedQueryU8) resident in HBM, for the multi-query MFMA path."""

    _prefix = "u8"


class EncodedVectorsU8(EncodedVectorsBase):
    _prefix = "u8"
    _query_cls = EncodedQueryU8
    _batch_cls = EncodedQueryBatchU8

    # ------------------------------------------------------------------ construction
    @classmethod
    def encode(cls, orig_data, vector_parameters: VectorParameters, quantile: float | None = None,
               stop_condition=None, *, alpha_offset: tuple[float, float] | None = None,
               stream=None) -> "EncodedVectorsU8":
        """EncodedVectorsU8::encode (:34-140).  `orig_data`: [count, dim] f32 array (host) or
        CUDA tensor (HBM), or an iterable of rows.  The storage builder of the reference is
        implicit: the encoded rows live in library-owned HBM (see `storage_bytes`)."""
        data = flatten_rows(orig_data, vector_parameters.dim)
        validate(data, vector_parameters)
        vp = vector_parameters.to_c()
        buf = in_buf(data, np.float32)
        q = C.c_float(quantile) if quantile is not None else None
        ao = (C.c_float * 2)(*alpha_offset) if alpha_offset is not None else None
        stop = make_stop(stop_condition)
        out = C.c_void_p()
        with creating_on(data) as dev:
            check(_lib.lib().qamd_u8_encode(buf.ptr, buf.mem, C.byref(vp),
                                            C.byref(q) if q is not None else None,
                                            C.cast(ao, C.POINTER(C.c_float)) if ao is not None else None,
                                            stop, None, stream_ptr(stream), C.byref(out)))
        return cls(out, dev)

    @staticmethod
    def find_min_max(rows, stream=None) -> tuple[np.float32, np.float32]:
        """find_min_max_from_iter (quantile.rs:5-19) over `rows` ([n, dim] f32, host or HBM) - pass 1 of `encode` on its
        own, for callers that hold the data in several places (quantization_amd.sharded.encode_u8)."""
        n, dim = (int(rows.shape[0]), int(rows.shape[1])) if len(rows.shape) == 2 else (0, 0)
        buf = in_buf(rows, np.float32)
        mn, mx = C.c_float(), C.c_float()
        with creating_on(rows):
            check(_lib.lib().qamd_u8_find_min_max(buf.ptr, buf.mem, n, dim, stream_ptr(stream), C.byref(mn), C.byref(mx)))
        return np.float32(mn.value), np.float32(mx.value)

    @staticmethod
    def find_quantile_interval(rows, quantile: float, stream=None) -> tuple[np.float32, np.float32] | None:
        """find_quantile_interval (quantile.rs:21-71) over `rows`; None as in the reference."""
        n, dim = (int(rows.shape[0]), int(rows.shape[1])) if len(rows.shape) == 2 else (0, 0)
        buf = in_buf(rows, np.float32)
        found, mn, mx = C.c_int32(), C.c_float(), C.c_float()
        with creating_on(rows):
            check(_lib.lib().qamd_u8_find_quantile_interval(buf.ptr, buf.mem, n, dim, C.c_float(quantile), stream_ptr(stream),
                                                            C.byref(found), C.byref(mn), C.byref(mx)))
        return (np.float32(mn.value), np.float32(mx.value)) if found.value else None

    @classmethod
    def encode_stream(cls, make_batches, vector_parameters: VectorParameters, quantile: float | None = None,
                      stop_condition=None, *, alpha_offset: tuple[float, float] | None = None,
                      stream=None) -> "EncodedVectorsU8":
        """The reference's own contract: `encode` takes a CLONABLE ITERATOR and walks it twice
        (:34-40, pass 1 :57-71, pass 2 :73-118).  `make_batches()` returns a fresh iterator over
        [n_i, dim] f32 batches (numpy or CUDA tensors) each time it is called; the f32 data is never
        held as a whole.  Byte-identical to `encode` on the concatenated batches."""
        L = _lib.lib()
        vp = vector_parameters.to_c()
        q = C.c_float(quantile) if quantile is not None else None
        ao = (C.c_float * 2)(*alpha_offset) if alpha_offset is not None else None
        stop = make_stop(stop_condition)
        first = next(iter(make_batches()), None)
        enc = C.c_void_p()
        with creating_on(first) as dev:
            check(L.qamd_u8_encoder_begin(C.byref(vp), C.byref(q) if q is not None else None,
                                          C.cast(ao, C.POINTER(C.c_float)) if ao is not None else None,
                                          stop, None, stream_ptr(stream), C.byref(enc)))
        try:
            for fn in ((L.qamd_u8_encoder_observe,) if alpha_offset is None else ()) + (L.qamd_u8_encoder_push,):
                for batch in make_batches():
                    if len(batch.shape) != 2 or (batch.shape[0] and batch.shape[1] != vector_parameters.dim):
                        raise EncodingError(_lib.ERR_ARGUMENTS, f"Vector length {batch.shape[-1]} does not match "
                                                                f"vector parameters dim {vector_parameters.dim}")
                    check_same_device(dev, batch)
                    buf = in_buf(batch, np.float32)
                    check(fn(enc, buf.ptr, int(batch.shape[0]), buf.mem))
            out = C.c_void_p()
            h, enc = enc, None
            check(L.qamd_u8_encoder_finish(h, C.byref(out)))
        finally:
            if enc is not None:
                L.qamd_u8_encoder_abort(enc)
        return cls(out, dev)

    @classmethod
    def from_storage(cls, rows, metadata: dict, stream=None) -> "EncodedVectorsU8":
        """Adopt reference-format rows ([vector_offset f32][codes], stride actual_dim+4) plus
        their Metadata (:24-31) — e.g. a store encoded by the reference crate."""
        vp = metadata["vector_parameters"]
        if isinstance(vp, dict):
            dt = vp["distance_type"]
            vp = VectorParameters(vp["dim"], vp["count"],
                                  DistanceType[dt] if isinstance(dt, str) else DistanceType(dt), vp["invert"])
        meta = _lib.U8MetadataC(int(metadata["actual_dim"]), float(metadata["alpha"]),
                                float(metadata["offset"]), float(metadata["multiplier"]), vp.to_c())
        buf = in_buf(rows, np.uint8)
        out = C.c_void_p()
        with creating_on(rows) as dev:
            check(_lib.lib().qamd_u8_from_rows(buf.ptr, buf.mem, C.byref(meta), stream_ptr(stream), C.byref(out)))
        return cls(out, dev)

    @classmethod
    def load(cls, data_path, meta_path, vector_parameters: VectorParameters) -> "EncodedVectorsU8":
        """EncodedVectors::load (:273-288)."""
        vp = vector_parameters.to_c()
        out = C.c_void_p()
        check(_lib.lib().qamd_u8_load(os.fsencode(data_path), os.fsencode(meta_path), C.byref(vp), C.byref(out)))
        return cls(out, get_device())

    def save(self, data_path, meta_path) -> None:
        """EncodedVectors::save (:263-271): raw rows + serde_json metadata."""
        check(_lib.lib().qamd_u8_save(self._h, os.fsencode(data_path), os.fsencode(meta_path)))

    # ------------------------------------------------------------------ metadata
    @property
    def metadata(self) -> dict:
        m = _lib.U8MetadataC()
        check(_lib.lib().qamd_u8_get_metadata(self._h, C.byref(m)))
        return {"actual_dim": int(m.actual_dim), "alpha": np.float32(m.alpha), "offset": np.float32(m.offset),
                "multiplier": np.float32(m.multiplier),
                "vector_parameters": VectorParameters.from_c(m.vector_parameters)}

    @property
    def vector_parameters(self) -> VectorParameters:
        return self.metadata["vector_parameters"]

    @staticmethod
    def get_quantized_vector_size(vector_parameters: VectorParameters) -> int:
        vp = vector_parameters.to_c()
        return int(_lib.lib().qamd_u8_quantized_vector_size(C.byref(vp)))

    @staticmethod
    def get_actual_dim(vector_parameters: VectorParameters) -> int:
        vp = vector_parameters.to_c()
        return int(_lib.lib().qamd_u8_actual_dim(C.byref(vp)))

    def storage_bytes(self, out=None, stream=None):
        """The rows as the reference's storage holds them (what save_to_file writes):
        [count, actual_dim + 4] u8."""
        md = self.metadata
        n, stride = md["vector_parameters"].count, md["actual_dim"] + 4
        check_same_device(self._device, out)
        buf, ret = out_buf(out, n * stride, np.uint8)
        check(_lib.lib().qamd_u8_export_rows(self._h, buf.ptr, buf.mem, stream_ptr(stream)))
        return ret.reshape(n, stride) if isinstance(ret, np.ndarray) else ret

    def storage_rows(self, first_row: int, n_rows: int, out=None, stream=None):
        """Rows [first_row, first_row + n_rows) in the reference's storage format — what a caller-owned
        EncodedStorageBuilder receives through push_vector_data (encoded_storage.rs:17-25), piece by piece."""
        stride = self.metadata["actual_dim"] + 4
        check_same_device(self._device, out)
        buf, ret = out_buf(out, n_rows * stride, np.uint8)
        check(_lib.lib().qamd_u8_export_rows_range(self._h, int(first_row), int(n_rows), buf.ptr, buf.mem,
                                                   stream_ptr(stream)))
        return ret.reshape(n_rows, stride) if isinstance(ret, np.ndarray) else ret

    def set_lane_mode(self, mode: int) -> None:
        """0: integer sum rounded once (default); 1: avx2.c 8-lane f32 summation order."""
        check(_lib.lib().qamd_u8_set_lane_mode(self._h, mode))

    # multi-query (MFMA) path: encode_query_batch / score_batch / topk_batch come from EncodedVectorsBase
    # (on the GPU: a dense u8 x u8 -> i32 contraction on the matrix cores, csrc/u8_batch.hip).

    # queries and scores: encode_query / score_point / score_internal / score_all /
    # score_ids / topk come from EncodedVectorsBase.
    def scan_bytes_per_row(self) -> int:
        return int(_lib.lib().qamd_u8_scan_bytes_per_row(self._h))

