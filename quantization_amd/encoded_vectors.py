"""Shared types of the quantizer API — mirrors quantization/src/encoded_vectors.rs and the
error enum of quantization/src/lib.rs:18-24.

Buffers: every method accepts numpy arrays (host memory) or torch CUDA tensors (HBM).  Torch is
only plumbing here (device memory + streams); nothing is computed with it.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import _lib


class DistanceType(enum.IntEnum):
    """encoded_vectors.rs:6-11.  Cosine = L2-normalise vectors and queries, then Dot
    (demos/src/ann_benchmark_data.rs:84-91,223-230)."""

    Dot = 0
    L1 = 1
    L2 = 2


@dataclass
class VectorParameters:
    """encoded_vectors.rs:13-19."""

    dim: int
    count: int
    distance_type: DistanceType
    invert: bool

    def to_c(self) -> _lib.VectorParametersC:
        return _lib.VectorParametersC(int(self.dim), int(self.count), int(self.distance_type),
                                      int(bool(self.invert)))

    @staticmethod
    def from_c(c: _lib.VectorParametersC) -> "VectorParameters":
        return VectorParameters(int(c.dim), int(c.count), DistanceType(c.distance_type), bool(c.invert))


class EncodingError(Exception):
    """lib.rs:18-24: IOError | EncodingError | ArgumentsError | Stopped (+ OutOfRange for the
    reference's slice-index panic, Device for HIP failures)."""

    KINDS = {1: "IOError", 2: "EncodingError", 3: "ArgumentsError", 4: "Stopped",
             5: "OutOfRange", 6: "DeviceError"}

    def __init__(self, status: int, message: str):
        self.status = status
        self.kind = self.KINDS.get(status, f"Status{status}")
        super().__init__(f"{self.kind}: {message}" if message and self.kind != "Stopped" else self.kind)

    @property
    def stopped(self) -> bool:
        return self.status == _lib.ERR_STOPPED


def check(status: int) -> None:
    if status != _lib.OK:
        msg = _lib.lib().qamd_last_error()
        err = EncodingError(status, msg.decode() if msg else "")
        if status == _lib.ERR_IO:
            raise OSError(str(err)) from err  # save/load return std::io::Result in the reference
        if status == _lib.ERR_OUT_OF_RANGE:
            raise IndexError(str(err)) from err
        raise err


def set_device(device: int) -> None:
    """Device of the handles this THREAD creates afterwards from host data (default 0).  Handles
    built from a CUDA tensor live on that tensor's device whatever this says."""
    check(_lib.lib().qamd_set_device(int(device)))


def get_device() -> int:
    return int(_lib.lib().qamd_get_device())


def device_of(x) -> int | None:
    """Device index of a CUDA tensor, None for host data."""
    if _is_torch(x) and x.is_cuda:
        return x.device.index if x.device.index is not None else 0
    return None


class creating_on:
    """`with creating_on(data): ...` -- handles created inside live on `data`'s device when it is a
    CUDA tensor (the thread's qamd device is switched and restored), else on the thread's device."""

    def __init__(self, *buffers):
        self.want = next((d for d in map(device_of, buffers) if d is not None), None)
        self.prev = None

    def __enter__(self):
        if self.want is not None:
            self.prev = get_device()
            if self.prev != self.want:
                set_device(self.want)
        return self.want if self.want is not None else get_device()

    def __exit__(self, *exc):
        if self.prev is not None and self.prev != self.want:
            set_device(self.prev)
        return False


def check_same_device(handle_device: int | None, *buffers) -> None:
    """A device buffer must belong to the handle's device (include/quantization_amd.h conventions)."""
    if handle_device is None:
        return
    for b in buffers:
        d = device_of(b)
        if d is not None and d != handle_device:
            raise ValueError(f"buffer is on cuda:{d} but the encoded store lives on cuda:{handle_device}")


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class Buf:
    """A caller buffer as (pointer, memory kind); keeps the backing object alive."""

    __slots__ = ("ptr", "mem", "obj")

    def __init__(self, ptr, mem, obj):
        self.ptr, self.mem, self.obj = ptr, mem, obj


def in_buf(x, dtype) -> Buf:
    """Read-only input: numpy array / sequence (host) or torch tensor (cuda => device)."""
    if x is None:
        return Buf(None, _lib.MEM_HOST, None)
    if _is_torch(x):
        import torch
        want = {np.float32: torch.float32, np.uint8: torch.uint8, np.uint32: torch.int32}[dtype]
        if x.dtype != want and not (dtype == np.uint32 and x.dtype in (torch.int32, torch.uint32)):
            x = x.to(want)
        x = x.contiguous()
        if x.is_cuda:
            return Buf(C.c_void_p(x.data_ptr()), _lib.MEM_DEVICE, x)
        a = x.numpy()
        return Buf(C.c_void_p(a.ctypes.data), _lib.MEM_HOST, (x, a))
    a = np.ascontiguousarray(x, dtype=dtype)
    return Buf(C.c_void_p(a.ctypes.data), _lib.MEM_HOST, a)


def out_buf(x, n: int, dtype) -> tuple[Buf, object]:
    """Output of n elements: `x` may be None (a fresh numpy array is returned), a numpy array
    or a torch CUDA tensor (written in place, stays in HBM)."""
    if x is None:
        a = np.empty(n, dtype=dtype)
        return Buf(C.c_void_p(a.ctypes.data), _lib.MEM_HOST, a), a
    if _is_torch(x):
        if x.numel() < n or not x.is_contiguous():
            raise ValueError("output tensor too small or not contiguous")
        if x.is_cuda:
            return Buf(C.c_void_p(x.data_ptr()), _lib.MEM_DEVICE, x), x
        a = x.numpy()
        return Buf(C.c_void_p(a.ctypes.data), _lib.MEM_HOST, (x, a)), x
    if x.dtype != dtype or x.size < n or not x.flags.c_contiguous:
        raise ValueError("output array has the wrong dtype/size or is not contiguous")
    return Buf(C.c_void_p(x.ctypes.data), _lib.MEM_HOST, x), x


def stream_ptr(stream) -> C.c_void_p:
    """None => torch's current stream when torch has CUDA initialised, else the null stream;
    int => a raw hipStream_t; torch.cuda.Stream => its handle."""
    if stream is None:
        import sys
        t = sys.modules.get("torch")
        if t is not None and t.cuda.is_available() and t.cuda.is_initialized():
            return C.c_void_p(t.cuda.current_stream().cuda_stream)
        return C.c_void_p(0)
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


def flatten_rows(data, dim_hint: int | None = None):
    """The reference takes an iterator of &[f32]; here a [count, dim] array/tensor, or an
    iterable of rows which is stacked."""
    if _is_torch(data) or isinstance(data, np.ndarray):
        return data
    rows = [np.asarray(r, dtype=np.float32) for r in data]
    if not rows:
        return np.zeros((0, dim_hint or 0), dtype=np.float32)
    return np.stack(rows)


def make_stop(stop_condition):
    """Wrap a Python `() -> bool` as the C stop callback (kept alive by the caller)."""
    if stop_condition is None:
        return C.cast(None, _lib.STOP_FN)
    return _lib.STOP_FN(lambda _user: 1 if stop_condition() else 0)


def validate(data, vp: VectorParameters) -> None:
    """validate_vector_parameters (encoded_vectors.rs:47-70)."""
    shape = tuple(data.shape)
    count = shape[0] if len(shape) >= 1 else 0
    dim = shape[1] if len(shape) >= 2 else 0
    if count and dim != vp.dim:
        raise EncodingError(_lib.ERR_ARGUMENTS,
                            f"Vector length {dim} does not match vector parameters dim {vp.dim}")
    if count != vp.count:
        raise EncodingError(_lib.ERR_ARGUMENTS,
                            f"Vector count {count} does not match vector parameters count {vp.count}")
