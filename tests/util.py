"""Small helpers shared by the parity tests."""
import numpy as np


def bits(a) -> np.ndarray:
    """f32 array -> its bit patterns (bit-exact comparisons; NaN-safe)."""
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(got, want, what=""):
    g, w = bits(got), bits(want)
    if not np.array_equal(g, w):
        bad = np.flatnonzero(g != w)
        i = int(bad[0])
        raise AssertionError(
            f"{what}: {bad.size}/{g.size} values differ; first at {i}: "
            f"got {np.asarray(got, dtype=np.float32).ravel()[i]!r} want {np.asarray(want, dtype=np.float32).ravel()[i]!r}")


def have_gpu() -> bool:
    try:
        from quantization_amd import _lib
        return _lib.lib().qamd_device_count() > 0
    except Exception:
        return False
