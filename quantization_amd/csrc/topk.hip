// Device-side top-k over a score array in HBM (exact, deterministic).
//
// Replaces the per-query 30-element BinaryHeap of the reference's caller
// (demos/src/ann_benchmark_data.rs:151-167).  Selection is an MSB-first radix select on a
// 64-bit composite key  (order-preserving score bits << 32 | row index):  all keys are
// distinct, so exactly k rows are selected and ties break to the LOWER index, whatever the
// launch geometry.  Eight histogram passes (HBM-bound reads of 4 B/row), one gather and a
// single-workgroup bitonic sort of the k winners.
#include "topk.hpp"

namespace qamd {
namespace {

constexpr int kBlock = 256;

struct SelState {
    unsigned long long prefix;
    unsigned long long mask;
    uint32_t k_rem;
    uint32_t out_count;
    uint32_t hist[256];
};

__device__ __forceinline__ uint32_t ordered_bits(float f, bool largest) {
    uint32_t u = __float_as_uint(f);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // ascending total order
    return largest ? ~u : u;                      // we always select the k SMALLEST keys
}

__device__ __forceinline__ unsigned long long composite(float f, uint32_t idx, bool largest) {
    return ((unsigned long long)ordered_bits(f, largest) << 32) | idx;
}

__global__ void init_kernel(SelState *st, uint32_t k) {
    int t = threadIdx.x;
    if (t == 0) {
        st->prefix = 0;
        st->mask = 0;
        st->k_rem = k;
        st->out_count = 0;
    }
    st->hist[t] = 0;
}

__global__ __launch_bounds__(kBlock) void hist_kernel(const float *__restrict__ scores, uint64_t n,
                                                     int shift, bool largest, SelState *st) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, mask = st->mask;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        unsigned long long key = composite(scores[i], (uint32_t)i, largest);
        if ((key & mask) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    uint32_t c = h[threadIdx.x];
    if (c) atomicAdd(&st->hist[threadIdx.x], c);
}

// One workgroup: find the bucket holding the k_rem-th smallest key, extend the prefix.
__global__ void pick_kernel(SelState *st, int shift) {
    __shared__ uint32_t cum[256];
    int t = threadIdx.x;
    uint32_t c = st->hist[t];
    cum[t] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // inclusive scan
        uint32_t v = t >= off ? cum[t - off] : 0;
        __syncthreads();
        cum[t] += v;
        __syncthreads();
    }
    uint32_t k_rem = st->k_rem;
    uint32_t before = cum[t] - c;
    __syncthreads();
    if (before < k_rem && k_rem <= cum[t]) {
        st->prefix |= (unsigned long long)t << shift;
        st->mask |= 0xFFull << shift;
        st->k_rem = k_rem - before;
    }
    st->hist[t] = 0;
}

__global__ __launch_bounds__(kBlock) void gather_kernel(const float *__restrict__ scores, uint64_t n,
                                                       bool largest, SelState *st,
                                                       unsigned long long *__restrict__ cand,
                                                       uint32_t cap) {
    const unsigned long long thr = st->prefix;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        unsigned long long key = composite(scores[i], (uint32_t)i, largest);
        if (key <= thr) {
            uint32_t pos = atomicAdd(&st->out_count, 1u);
            if (pos < cap) cand[pos] = key;
        }
    }
}

// One workgroup of 1024 threads: bitonic sort of up to 1024 keys, then decode.
__global__ __launch_bounds__(1024) void sort_emit_kernel(const float *__restrict__ scores,
                                                        const unsigned long long *__restrict__ cand,
                                                        uint32_t n_valid, uint32_t k, bool largest,
                                                        uint32_t *__restrict__ out_ids,
                                                        float *__restrict__ out_scores) {
    __shared__ unsigned long long s[1024];
    int t = threadIdx.x;
    s[t] = (uint32_t)t < n_valid ? cand[t] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            int partner = t ^ stride;
            if (partner > t) {
                bool up = (t & size) == 0;
                unsigned long long a = s[t], b = s[partner];
                if ((a > b) == up) {
                    s[t] = b;
                    s[partner] = a;
                }
            }
            __syncthreads();
        }
    }
    if ((uint32_t)t < k) {
        if ((uint32_t)t < n_valid) {
            uint32_t idx = (uint32_t)(s[t] & 0xFFFFFFFFull);
            out_ids[t] = idx;
            out_scores[t] = scores[idx];
        } else {  // fewer rows than k: pad with the worst possible entry
            out_ids[t] = 0xFFFFFFFFu;
            out_scores[t] = largest ? -__builtin_huge_valf() : __builtin_huge_valf();
        }
    }
}

}  // namespace

size_t topk_workspace_bytes(uint32_t k) {
    (void)k;
    return sizeof(SelState) + 1024 * sizeof(unsigned long long) + 64;
}

qamd_status topk_f32(const float *scores_dev, uint64_t n, uint32_t k, bool largest,
                     uint32_t *out_ids_dev, float *out_scores_dev, void *workspace_dev,
                     hipStream_t stream) {
    if (k == 0) return QAMD_OK;
    if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
    SelState *st = static_cast<SelState *>(workspace_dev);
    unsigned long long *cand =
        reinterpret_cast<unsigned long long *>(static_cast<char *>(workspace_dev) +
                                               round_up(sizeof(SelState), 64));
    uint32_t k_eff = n < k ? (uint32_t)n : k;
    hipLaunchKernelGGL(init_kernel, dim3(1), dim3(256), 0, stream, st, k_eff);
    if (k_eff > 0) {
        uint64_t want = (n + kBlock * 8 - 1) / (kBlock * 8);
        uint64_t cap = (uint64_t)device_info().cu_count * 8;
        int grid = (int)(want < 1 ? 1 : (want > cap ? cap : want));
        for (int shift = 56; shift >= 0; shift -= 8) {
            hipLaunchKernelGGL(hist_kernel, dim3(grid), dim3(kBlock), 0, stream, scores_dev, n, shift,
                               largest, st);
            hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(256), 0, stream, st, shift);
        }
        hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(kBlock), 0, stream, scores_dev, n, largest,
                           st, cand, 1024u);
    }
    hipLaunchKernelGGL(sort_emit_kernel, dim3(1), dim3(1024), 0, stream, scores_dev, cand, k_eff, k,
                       largest, out_ids_dev, out_scores_dev);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// Workspace + result staging in stream order; host outputs make the call synchronous.
qamd_status topk_finish(const float *scores_dev, uint64_t n, uint32_t k, int largest,
                        uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t stream) {
    if (k == 0) return QAMD_OK;
    if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
    void *ws = nullptr;
    size_t ws_bytes = round_up(topk_workspace_bytes(k), 256);
    size_t extra = out_mem == QAMD_MEM_HOST ? (size_t)k * 8 : 0;
    QAMD_HIP(hipMallocAsync(&ws, ws_bytes + extra, stream));
    uint32_t *ids_dev = out_ids;
    float *sc_dev = out_scores;
    if (out_mem == QAMD_MEM_HOST) {
        ids_dev = reinterpret_cast<uint32_t *>(static_cast<char *>(ws) + ws_bytes);
        sc_dev = reinterpret_cast<float *>(ids_dev + k);
    }
    qamd_status st = topk_f32(scores_dev, n, k, largest != 0, ids_dev, sc_dev, ws, stream);
    if (st == QAMD_OK && out_mem == QAMD_MEM_HOST) {
        st = copy_out(out_ids, QAMD_MEM_HOST, ids_dev, (size_t)k * 4, stream);
        if (st == QAMD_OK) st = copy_out(out_scores, QAMD_MEM_HOST, sc_dev, (size_t)k * 4, stream);
    }
    (void)hipFreeAsync(ws, stream);
    return st;
}

}  // namespace qamd

extern "C" qamd_status qamd_topk_scores(const float *scores_dev, uint64_t n, uint32_t k, int largest,
                                        uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (k == 0) return QAMD_OK;
    if (!scores_dev || !out_ids || !out_scores) return qamd::fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_TRY(qamd::ensure_device(qamd::current_device()));
    return qamd::topk_finish(scores_dev, n, k, largest, out_ids, out_scores, out_mem, qamd::as_stream(stream));
}
