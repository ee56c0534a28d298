"""EncodedVectorsBin — host-side mirror of quantization/src/encoded_vectors_binary.rs."""
from __future__ import annotations

import ctypes as C
import enum
import os

import numpy as np

from . import _lib
from ._base import EncodedQueryBase, EncodedVectorsBase
from .encoded_vectors import (EncodingError, VectorParameters, check, check_same_device, creating_on,
                              flatten_rows, get_device, in_buf, make_stop, out_buf, stream_ptr, validate)


class BitsStoreType(enum.IntEnum):
    """The two `impl BitsStoreType` of the reference (:44, :119).  Byte layout is the same on
    little-endian; only the row padding differs (:99-116 vs :152-159)."""

    U8 = 0
    U128 = 1


class EncodedBinVector(EncodedQueryBase):
    """encoded_vectors_binary.rs:17-19."""

    _prefix = "bin"

    @property
    def encoded_vector(self) -> np.ndarray:
        n = C.c_uint64()
        check(_lib.lib().qamd_bin_query_read(self._h, None, 0, C.byref(n)))
        bits = np.zeros(n.value, dtype=np.uint8)
        if n.value:
            check(_lib.lib().qamd_bin_query_read(self._h, C.c_void_p(bits.ctypes.data), n.value, None))
        return bits


class EncodedVectorsBin(EncodedVectorsBase):
    _prefix = "bin"
    _query_cls = EncodedBinVector

    def __init__(self, handle, vector_parameters: VectorParameters, store: BitsStoreType, device=None,
                 owned: bool = True):
        super().__init__(handle, device, owned)
        self._vp = vector_parameters
        self._store = BitsStoreType(store)

    @property
    def vector_parameters(self) -> VectorParameters:
        return self._vp

    @property
    def metadata(self) -> dict:
        return {"vector_parameters": self._vp}  # :21-24

    @classmethod
    def encode(cls, orig_data, vector_parameters: VectorParameters, stop_condition=None, *,
               store: BitsStoreType = BitsStoreType.U8, stream=None) -> "EncodedVectorsBin":
        """EncodedVectorsBin::<TBitsStoreType, _>::encode (:165-191)."""
        data = flatten_rows(orig_data, vector_parameters.dim)
        validate(data, vector_parameters)
        vp = vector_parameters.to_c()
        buf = in_buf(data, np.float32)
        stop = make_stop(stop_condition)
        out = C.c_void_p()
        with creating_on(data) as dev:
            check(_lib.lib().qamd_bin_encode(buf.ptr, buf.mem, C.byref(vp), int(store), stop, None,
                                             stream_ptr(stream), C.byref(out)))
        return cls(out, vector_parameters, store, dev)

    @classmethod
    def encode_stream(cls, make_batches, vector_parameters: VectorParameters, stop_condition=None, *,
                      store: BitsStoreType = BitsStoreType.U8, stream=None) -> "EncodedVectorsBin":
        """encode from the reference's iterator contract (:165-191 walks it once): `make_batches()`
        returns an iterator over [n_i, dim] f32 batches; rows are appended in order."""
        L = _lib.lib()
        vp = vector_parameters.to_c()
        stop = make_stop(stop_condition)
        first = next(iter(make_batches()), None)
        enc = C.c_void_p()
        with creating_on(first) as dev:
            check(L.qamd_bin_encoder_begin(C.byref(vp), int(store), stop, None, stream_ptr(stream), C.byref(enc)))
        try:
            for batch in make_batches():
                if len(batch.shape) != 2 or (batch.shape[0] and batch.shape[1] != vector_parameters.dim):
                    raise EncodingError(_lib.ERR_ARGUMENTS, f"Vector length {batch.shape[-1]} does not match "
                                                            f"vector parameters dim {vector_parameters.dim}")
                check_same_device(dev, batch)
                buf = in_buf(batch, np.float32)
                check(L.qamd_bin_encoder_push(enc, buf.ptr, int(batch.shape[0]), buf.mem))
            out = C.c_void_p()
            h, enc = enc, None
            check(L.qamd_bin_encoder_finish(h, C.byref(out)))
        finally:
            if enc is not None:
                L.qamd_bin_encoder_abort(enc)
        return cls(out, vector_parameters, store, dev)

    @classmethod
    def from_storage(cls, rows, vector_parameters: VectorParameters,
                     store: BitsStoreType = BitsStoreType.U8, stream=None) -> "EncodedVectorsBin":
        vp = vector_parameters.to_c()
        buf = in_buf(rows, np.uint8)
        out = C.c_void_p()
        with creating_on(rows) as dev:
            check(_lib.lib().qamd_bin_from_rows(buf.ptr, buf.mem, C.byref(vp), int(store), stream_ptr(stream),
                                                C.byref(out)))
        return cls(out, vector_parameters, store, dev)

    @classmethod
    def load(cls, data_path, meta_path, vector_parameters: VectorParameters,
             store: BitsStoreType = BitsStoreType.U8) -> "EncodedVectorsBin":
        """EncodedVectors::load (:270-286)."""
        vp = vector_parameters.to_c()
        out = C.c_void_p()
        check(_lib.lib().qamd_bin_load(os.fsencode(data_path), os.fsencode(meta_path), C.byref(vp), int(store),
                                       C.byref(out)))
        import json
        m = json.load(open(meta_path))["vector_parameters"]
        from .encoded_vectors import DistanceType
        eff = VectorParameters(vector_parameters.dim, vector_parameters.count,
                               DistanceType[m["distance_type"]], bool(m["invert"]))
        return cls(out, eff, store, get_device())

    def save(self, data_path, meta_path) -> None:
        """EncodedVectors::save (:260-268)."""
        check(_lib.lib().qamd_bin_save(self._h, os.fsencode(data_path), os.fsencode(meta_path)))

    @staticmethod
    def get_quantized_vector_size_from_params(vector_parameters: VectorParameters,
                                              store: BitsStoreType = BitsStoreType.U8) -> int:
        """:210-213, bytes per row."""
        vp = vector_parameters.to_c()
        return int(_lib.lib().qamd_bin_quantized_vector_size(C.byref(vp), int(store)))

    def storage_bytes(self, out=None, stream=None):
        n = self._vp.count
        nb = self.get_quantized_vector_size_from_params(self._vp, self._store)
        check_same_device(self._device, out)
        buf, ret = out_buf(out, n * nb, np.uint8)
        check(_lib.lib().qamd_bin_export_rows(self._h, buf.ptr, buf.mem, stream_ptr(stream)))
        return ret.reshape(n, nb) if isinstance(ret, np.ndarray) else ret

    def storage_rows(self, first_row: int, n_rows: int, out=None, stream=None):
        """Rows [first_row, first_row + n_rows) as push_vector_data would receive them (encoded_storage.rs:17-25)."""
        nb = self.get_quantized_vector_size_from_params(self._vp, self._store)
        check_same_device(self._device, out)
        buf, ret = out_buf(out, n_rows * nb, np.uint8)
        check(_lib.lib().qamd_bin_export_rows_range(self._h, int(first_row), int(n_rows), buf.ptr, buf.mem,
                                                    stream_ptr(stream)))
        return ret.reshape(n_rows, nb) if isinstance(ret, np.ndarray) else ret
