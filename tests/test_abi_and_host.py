"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol,
fails loudly without a GPU (no CPU fallback), and the host mirror validates like the reference."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import quantization_amd as qa
from quantization_amd import _lib
from util import have_gpu


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _lib.declared_symbols()
    assert len(declared) >= 56
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/quantization_amd.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    # the product library exports EXACTLY the declared entry points: no developer hooks, no tuning
    # harness (those live in libquantization_amd_dev.so, `make dev`)
    assert exported == set(declared), (sorted(exported - set(declared)), sorted(set(declared) - exported))
    assert not any(s.startswith("qamd_dev_") for s in exported)
    # nothing of the oracle is linked into or referenced by the product
    assert not any(s.startswith("qo_") for s in exported)
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "qoracle" not in needed


def test_product_sources_never_touch_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dirpath, _, files in os.walk(os.path.join(root, "quantization_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "qoracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_size_helpers_need_no_gpu():
    L = _lib.lib()
    vp = qa.VectorParameters(65, 129, qa.DistanceType.Dot, False)
    assert qa.EncodedVectorsU8.get_actual_dim(vp) == 80
    assert qa.EncodedVectorsU8.get_quantized_vector_size(vp) == 84
    assert qa.EncodedVectorsU8.get_quantized_vector_size(qa.VectorParameters(768, 1, qa.DistanceType.L2, False)) == 772
    assert qa.EncodedVectorsPQ.get_quantized_vector_size(qa.VectorParameters(768, 1, qa.DistanceType.Dot, False), 8) == 96
    for dim, nb in {0: 0, 1: 1, 33: 8, 65: 16, 387: 64, 1024: 128}.items():
        v = qa.VectorParameters(dim, 1, qa.DistanceType.Dot, False)
        assert qa.EncodedVectorsBin.get_quantized_vector_size_from_params(v, qa.BitsStoreType.U8) == nb
    assert qa.EncodedVectorsBin.get_quantized_vector_size_from_params(
        qa.VectorParameters(1, 1, qa.DistanceType.Dot, False), qa.BitsStoreType.U128) == 16
    assert L.qamd_version().startswith(b"quantization_amd")


def test_argument_validation_mirrors_reference():
    """validate_vector_parameters (encoded_vectors.rs:47-70) runs before anything touches the GPU."""
    data = np.zeros((4, 8), dtype=np.float32)
    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsU8.encode(data, qa.VectorParameters(8, 5, qa.DistanceType.Dot, False))
    assert e.value.kind == "ArgumentsError" and "count" in str(e.value)
    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsBin.encode(data, qa.VectorParameters(9, 4, qa.DistanceType.Dot, False))
    assert "dim" in str(e.value)


@pytest.mark.skipif(have_gpu(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback_fails_loudly():
    data = np.zeros((4, 16), dtype=np.float32)
    for make in (
        lambda: qa.EncodedVectorsU8.encode(data, qa.VectorParameters(16, 4, qa.DistanceType.Dot, False)),
        lambda: qa.EncodedVectorsBin.encode(data, qa.VectorParameters(16, 4, qa.DistanceType.Dot, False)),
        lambda: qa.EncodedVectorsPQ.encode(data, qa.VectorParameters(16, 4, qa.DistanceType.Dot, False), 4),
        # round 2 entry points: streaming encoders and row-sharded handles
        lambda: qa.EncodedVectorsU8.encode_stream(lambda: iter([data]), qa.VectorParameters(16, 4, qa.DistanceType.Dot, False)),
        lambda: qa.EncodedVectorsBin.encode_stream(lambda: iter([data]), qa.VectorParameters(16, 4, qa.DistanceType.Dot, False)),
        lambda: qa.ShardedVectorsU8.encode(data, qa.VectorParameters(16, 4, qa.DistanceType.Dot, False), [0, 0]),
        lambda: qa.ShardedVectorsBin.encode(data, qa.VectorParameters(16, 4, qa.DistanceType.Dot, False), [0]),
    ):
        with pytest.raises(qa.EncodingError) as e:
            make()
        assert e.value.kind == "DeviceError" and "no CPU fallback" in str(e.value)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "absent.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
