"""ctypes binding of libquantization_amd.so (the C ABI in include/quantization_amd.h).

There is no CPU fallback: if the HIP library is missing, or no GPU is visible when a
function needs one, this raises.  `build()` compiles the library in-tree with hipcc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QAMD_LIB_PATH") or os.path.join(HERE, "libquantization_amd.so")  # env: developer A/B of two builds
CSRC = os.path.join(HERE, "csrc")

MEM_HOST, MEM_DEVICE = 0, 1
OK, ERR_IO, ERR_ENCODING, ERR_ARGUMENTS, ERR_STOPPED, ERR_OUT_OF_RANGE, ERR_DEVICE = range(7)


class VectorParametersC(C.Structure):
    _fields_ = [("dim", C.c_uint64), ("count", C.c_uint64),
                ("distance_type", C.c_int32), ("invert", C.c_int32)]


class U8MetadataC(C.Structure):
    _fields_ = [("actual_dim", C.c_uint64), ("alpha", C.c_float), ("offset", C.c_float),
                ("multiplier", C.c_float), ("vector_parameters", VectorParametersC)]


STOP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


PROBE_PATH = os.path.join(HERE, "..", "tools", "probe", "libqamd_probe.so")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into quantization_amd/libquantization_amd.so (and the bench's
    streaming-read probe into tools/probe/libqamd_probe.so)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith((".hip", ".cpp", ".hpp"))] + [
        os.path.join(HERE, "..", "include", "quantization_amd.h")]
    newest = max(os.path.getmtime(p) for p in srcs)
    probe_src = os.path.join(HERE, "..", "tools", "probe", "stream_read.hip")
    probe_stale = os.path.exists(probe_src) and (not os.path.exists(PROBE_PATH)
                                                  or os.path.getmtime(PROBE_PATH) < os.path.getmtime(probe_src))
    if force or probe_stale or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        cmd = ["make", "-C", CSRC, "-j8"]
        res = subprocess.run(cmd, capture_output=not verbose, text=True)
        if res.returncode != 0:
            raise RuntimeError("building libquantization_amd.so failed:\n" +
                               (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`); "
            "quantization_amd has no CPU fallback")
    # One HIP runtime per process: when torch is (going to be) in use, let it load its
    # bundled libamdhip64 first so that this library binds to the same one.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:  # torch absent: the system ROCm runtime is used
            pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32, f32p = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_float)
    VP = C.POINTER(VectorParametersC)
    pp = C.POINTER(C.c_void_p)
    sig = {
        "qamd_last_error": (C.c_char_p, []),
        "qamd_version": (C.c_char_p, []),
        "qamd_device_count": (i32, []),
        "qamd_set_device": (i32, [i32]),
        "qamd_get_device": (i32, []),
        "qamd_thread_release": (None, []),
        # u8
        "qamd_u8_quantized_vector_size": (u64, [VP]),
        "qamd_u8_actual_dim": (u64, [VP]),
        "qamd_u8_encode": (i32, [vp, i32, VP, f32p, f32p, STOP_FN, vp, vp, pp]),
        "qamd_u8_from_rows": (i32, [vp, i32, C.POINTER(U8MetadataC), vp, pp]),
        "qamd_u8_export_rows": (i32, [vp, vp, i32, vp]),
        "qamd_u8_export_rows_range": (i32, [vp, u64, u64, vp, i32, vp]),
        "qamd_u8_get_metadata": (i32, [vp, C.POINTER(U8MetadataC)]),
        "qamd_u8_save": (i32, [vp, C.c_char_p, C.c_char_p]),
        "qamd_u8_load": (i32, [C.c_char_p, C.c_char_p, VP, pp]),
        "qamd_u8_encode_query": (i32, [vp, vp, u64, i32, vp, pp]),
        "qamd_u8_query_read": (i32, [vp, f32p, vp, u64, C.POINTER(u64)]),
        "qamd_u8_query_free": (None, [vp]),
        "qamd_u8_score_point": (i32, [vp, vp, u32, f32p]),
        "qamd_u8_score_internal": (i32, [vp, u32, u32, f32p]),
        "qamd_u8_score_all": (i32, [vp, vp, vp, i32, vp]),
        "qamd_u8_score_internal_ids": (i32, [vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_u8_score_internal_ids_batch": (i32, [vp, vp, vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_u8_score_ids_batch": (i32, [vp, vp, vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_u8_score_ids": (i32, [vp, vp, vp, u64, i32, vp, i32, vp]),
        "qamd_u8_topk": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_u8_free": (None, [vp]),
        "qamd_u8_set_lane_mode": (i32, [vp, i32]),
        "qamd_u8_encode_query_batch": (i32, [vp, vp, u64, u64, i32, vp, pp]),
        "qamd_u8_query_batch_free": (None, [vp]),
        "qamd_u8_score_batch": (i32, [vp, vp, vp, i32, vp]),
        "qamd_u8_topk_batch": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_u8_scan_bytes_per_row": (u64, [vp]),
        "qamd_u8_encoder_begin": (i32, [VP, f32p, f32p, STOP_FN, vp, vp, pp]),
        "qamd_u8_encoder_observe": (i32, [vp, vp, u64, i32]),
        "qamd_u8_encoder_push": (i32, [vp, vp, u64, i32]),
        "qamd_u8_encoder_finish": (i32, [vp, pp]),
        "qamd_u8_encoder_abort": (None, [vp]),
        # binary
        "qamd_bin_quantized_vector_size": (u64, [VP, i32]),
        "qamd_bin_encode": (i32, [vp, i32, VP, i32, STOP_FN, vp, vp, pp]),
        "qamd_bin_from_rows": (i32, [vp, i32, VP, i32, vp, pp]),
        "qamd_bin_export_rows": (i32, [vp, vp, i32, vp]),
        "qamd_bin_export_rows_range": (i32, [vp, u64, u64, vp, i32, vp]),
        "qamd_bin_save": (i32, [vp, C.c_char_p, C.c_char_p]),
        "qamd_bin_load": (i32, [C.c_char_p, C.c_char_p, VP, i32, pp]),
        "qamd_bin_encode_query": (i32, [vp, vp, u64, i32, vp, pp]),
        "qamd_bin_query_read": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "qamd_bin_query_free": (None, [vp]),
        "qamd_bin_score_point": (i32, [vp, vp, u32, f32p]),
        "qamd_bin_score_internal": (i32, [vp, u32, u32, f32p]),
        "qamd_bin_score_all": (i32, [vp, vp, vp, i32, vp]),
        "qamd_bin_score_internal_ids": (i32, [vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_bin_score_internal_ids_batch": (i32, [vp, vp, vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_bin_score_ids_batch": (i32, [vp, vp, vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_bin_score_ids": (i32, [vp, vp, vp, u64, i32, vp, i32, vp]),
        "qamd_bin_topk": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_bin_free": (None, [vp]),
        "qamd_bin_encode_query_batch": (i32, [vp, vp, u64, u64, i32, vp, pp]),
        "qamd_bin_query_batch_free": (None, [vp]),
        "qamd_bin_score_batch": (i32, [vp, vp, vp, i32, vp]),
        "qamd_bin_topk_batch": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_bin_encoder_begin": (i32, [VP, i32, STOP_FN, vp, vp, pp]),
        "qamd_bin_encoder_push": (i32, [vp, vp, u64, i32]),
        "qamd_bin_encoder_finish": (i32, [vp, pp]),
        "qamd_bin_encoder_abort": (None, [vp]),
        # pq
        "qamd_pq_quantized_vector_size": (u64, [VP, u64]),
        "qamd_pq_encode": (i32, [vp, i32, VP, u64, vp, u32, STOP_FN, vp, vp, pp]),
        "qamd_pq_from_rows": (i32, [vp, i32, VP, u64, vp, vp, pp]),
        "qamd_pq_export_rows": (i32, [vp, vp, i32, vp]),
        "qamd_pq_export_rows_range": (i32, [vp, u64, u64, vp, i32, vp]),
        "qamd_pq_get_centroids": (i32, [vp, vp]),
        "qamd_pq_save": (i32, [vp, C.c_char_p, C.c_char_p]),
        "qamd_pq_load": (i32, [C.c_char_p, C.c_char_p, VP, pp]),
        "qamd_pq_encode_query": (i32, [vp, vp, u64, i32, vp, pp]),
        "qamd_pq_query_read": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "qamd_pq_query_free": (None, [vp]),
        "qamd_pq_score_point": (i32, [vp, vp, u32, f32p]),
        "qamd_pq_score_internal": (i32, [vp, u32, u32, f32p]),
        "qamd_pq_score_all": (i32, [vp, vp, vp, i32, vp]),
        "qamd_pq_score_internal_ids": (i32, [vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_pq_score_internal_ids_batch": (i32, [vp, vp, vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_pq_score_ids_batch": (i32, [vp, vp, vp, u32, vp, u64, i32, vp, i32, vp]),
        "qamd_pq_score_ids": (i32, [vp, vp, vp, u64, i32, vp, i32, vp]),
        "qamd_pq_topk": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_pq_free": (None, [vp]),
        "qamd_pq_encode_query_batch": (i32, [vp, vp, u64, u64, i32, vp, pp]),
        "qamd_pq_query_batch_free": (None, [vp]),
        "qamd_pq_score_batch": (i32, [vp, vp, vp, i32, vp]),
        "qamd_pq_topk_batch": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_pq_kmeans_info": (i32, [vp, C.POINTER(u32), C.POINTER(u32)]),
        "qamd_pq_scan_kernel": (C.c_char_p, [vp, C.POINTER(u32)]),
        "qamd_pq_find_centroids": (i32, [vp, i32, VP, u64, u32, STOP_FN, vp, vp, vp, C.POINTER(u32), C.POINTER(u32)]),
        "qamd_u8_find_min_max": (i32, [vp, i32, u64, u64, vp, f32p, f32p]),
        "qamd_u8_find_quantile_interval": (i32, [vp, i32, u64, u64, C.c_float, vp, C.POINTER(i32), f32p, f32p]),
        "qamd_pq_encoder_begin": (i32, [VP, u64, vp, u32, STOP_FN, vp, vp, pp]),
        "qamd_pq_encoder_observe": (i32, [vp, vp, u64, i32]),
        "qamd_pq_encoder_push": (i32, [vp, vp, u64, i32]),
        "qamd_pq_encoder_finish": (i32, [vp, pp]),
        "qamd_pq_encoder_abort": (None, [vp]),
        # row-sharded stores (one process, several GPUs)
        "qamd_u8_sharded_encode": (i32, [vp, i32, VP, f32p, f32p, STOP_FN, vp, C.POINTER(i32), u32, vp, pp]),
        "qamd_u8_sharded_from_rows": (i32, [vp, i32, C.POINTER(U8MetadataC), C.POINTER(i32), u32, vp, pp]),
        "qamd_u8_sharded_shard_count": (u32, [vp]),
        "qamd_u8_sharded_peer_access": (i32, [vp, u32, C.POINTER(i32), C.POINTER(C.c_char_p)]),
        "qamd_u8_sharded_shard": (i32, [vp, u32, pp, C.POINTER(u64), C.POINTER(i32)]),
        "qamd_u8_sharded_get_metadata": (i32, [vp, C.POINTER(U8MetadataC)]),
        "qamd_u8_sharded_encode_query": (i32, [vp, vp, u64, i32, vp, pp]),
        "qamd_u8_sharded_query_free": (None, [vp]),
        "qamd_u8_sharded_score_all": (i32, [vp, vp, vp, i32, vp]),
        "qamd_u8_sharded_topk": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_u8_sharded_encode_query_batch": (i32, [vp, vp, u64, u64, i32, vp, pp]),
        "qamd_u8_sharded_query_batch_free": (None, [vp]),
        "qamd_u8_sharded_topk_batch": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_u8_sharded_free": (None, [vp]),
        "qamd_bin_sharded_encode": (i32, [vp, i32, VP, i32, STOP_FN, vp, C.POINTER(i32), u32, vp, pp]),
        "qamd_bin_sharded_from_rows": (i32, [vp, i32, VP, i32, C.POINTER(i32), u32, vp, pp]),
        "qamd_bin_sharded_shard_count": (u32, [vp]),
        "qamd_bin_sharded_peer_access": (i32, [vp, u32, C.POINTER(i32), C.POINTER(C.c_char_p)]),
        "qamd_bin_sharded_shard": (i32, [vp, u32, pp, C.POINTER(u64), C.POINTER(i32)]),
        "qamd_bin_sharded_encode_query": (i32, [vp, vp, u64, i32, vp, pp]),
        "qamd_bin_sharded_query_free": (None, [vp]),
        "qamd_bin_sharded_score_all": (i32, [vp, vp, vp, i32, vp]),
        "qamd_bin_sharded_topk": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_bin_sharded_encode_query_batch": (i32, [vp, vp, u64, u64, i32, vp, pp]),
        "qamd_bin_sharded_query_batch_free": (None, [vp]),
        "qamd_bin_sharded_topk_batch": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_bin_sharded_free": (None, [vp]),
        "qamd_pq_sharded_encode": (i32, [vp, i32, VP, u64, vp, u32, STOP_FN, vp, C.POINTER(i32), u32, vp, pp]),
        "qamd_pq_sharded_from_rows": (i32, [vp, i32, VP, u64, vp, C.POINTER(i32), u32, vp, pp]),
        "qamd_pq_sharded_shard_count": (u32, [vp]),
        "qamd_pq_sharded_peer_access": (i32, [vp, u32, C.POINTER(i32), C.POINTER(C.c_char_p)]),
        "qamd_pq_sharded_shard": (i32, [vp, u32, pp, C.POINTER(u64), C.POINTER(i32)]),
        "qamd_pq_sharded_get_centroids": (i32, [vp, vp]),
        "qamd_pq_sharded_encode_query": (i32, [vp, vp, u64, i32, vp, pp]),
        "qamd_pq_sharded_query_free": (None, [vp]),
        "qamd_pq_sharded_score_all": (i32, [vp, vp, vp, i32, vp]),
        "qamd_pq_sharded_topk": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_pq_sharded_encode_query_batch": (i32, [vp, vp, u64, u64, i32, vp, pp]),
        "qamd_pq_sharded_query_batch_free": (None, [vp]),
        "qamd_pq_sharded_topk_batch": (i32, [vp, vp, u32, i32, vp, vp, i32, vp]),
        "qamd_pq_sharded_free": (None, [vp]),
        "qamd_topk_scores": (i32, [vp, u64, u32, i32, vp, vp, i32, vp]),
        "qamd_topk_merge": (i32, [vp, vp, u64, vp, u32, u32, u32, i32, vp, vp, i32, vp]),
    }
    undeclared = set(sig) - set(declared_symbols())
    if undeclared:  # the binding may only name what include/quantization_amd.h declares
        raise RuntimeError(f"_lib.py binds symbols the header does not declare: {sorted(undeclared)}")
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def declared_symbols() -> list[str]:
    """Every QAMD_API function declared in include/quantization_amd.h."""
    import re
    hdr = open(os.path.join(HERE, "..", "include", "quantization_amd.h")).read()
    return sorted(set(re.findall(r"QAMD_API[^;(]*?\b(qamd_\w+)\s*\(", hdr)))
