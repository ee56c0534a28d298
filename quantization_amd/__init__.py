"""quantization_amd — MI355X-native encode-and-score path of qdrant/quantization.

The product is the C ABI in include/quantization_amd.h (hand-written HIP kernels for gfx950,
quantization_amd/csrc).  These classes are the host-side mirror of the reference's
`EncodedVectors{U8,PQ,Bin}` API over that ABI.  No CPU fallback exists: without the built
HIP library and a GPU every call raises.
"""
from ._lib import build, lib  # noqa: F401
from .encoded_vectors import DistanceType, EncodingError, VectorParameters, get_device, set_device  # noqa: F401
from .encoded_vectors_binary import BitsStoreType, EncodedBinVector, EncodedVectorsBin  # noqa: F401
from .encoded_vectors_pq import EncodedQueryPQ, EncodedVectorsPQ  # noqa: F401
from .encoded_vectors_u8 import EncodedQueryBatchU8, EncodedQueryU8, EncodedVectorsU8  # noqa: F401
from .sharded_store import ShardedVectorsBin, ShardedVectorsPQ, ShardedVectorsU8  # noqa: F401


def thread_release() -> None:
    """Free what the calling thread has cached on the GPUs (see qamd_thread_release)."""
    lib().qamd_thread_release()



def topk_scores(scores, n: int, k: int, largest: bool = True, out_ids=None, out_scores=None, stream=None):
    """Best-k entries of a device score array (torch CUDA tensor), same ordering contract as
    `EncodedVectors*.topk`.  Returns (ids, scores)."""
    import ctypes as C

    import numpy as np

    from .encoded_vectors import check, out_buf, stream_ptr

    if not (hasattr(scores, "is_cuda") and scores.is_cuda):
        raise ValueError("scores must be a CUDA tensor (device memory)")
    ib, ids = out_buf(out_ids, k, np.uint32)
    sb, sc = out_buf(out_scores, k, np.float32)
    if ib.mem != sb.mem:
        raise ValueError("out_ids and out_scores must both be host or both be device buffers")
    check(lib().qamd_topk_scores(C.c_void_p(scores.data_ptr()), int(n), int(k), int(bool(largest)), ib.ptr, sb.ptr,
                                 sb.mem, stream_ptr(stream)))
    return ids, sc


__all__ = [
    "topk_scores", "set_device", "get_device", "thread_release",
    "ShardedVectorsU8", "ShardedVectorsBin", "ShardedVectorsPQ",
    "DistanceType", "VectorParameters", "EncodingError",
    "EncodedVectorsU8", "EncodedQueryU8", "EncodedQueryBatchU8",
    "EncodedVectorsPQ", "EncodedQueryPQ",
    "EncodedVectorsBin", "EncodedBinVector", "BitsStoreType",
    "build", "lib",
]
