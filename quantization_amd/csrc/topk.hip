// Device-side top-k over a score array in HBM (exact, deterministic).
//
// Replaces the per-query 30-element BinaryHeap of the reference's caller
// (demos/src/ann_benchmark_data.rs:151-167).  Selection is an MSB-first radix select on a
// 64-bit composite key  (order-preserving score bits << 32 | row index):  all keys are
// distinct, so exactly k rows are selected and ties break to the LOWER index, whatever the
// launch geometry.  Eight histogram passes (HBM-bound reads of 4 B/row), one gather and a
// single-workgroup bitonic sort of the k winners.
#include "topk.hpp"
#include "topk_device.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdlib>

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

// k-th smallest (or largest) VALUE of an f32 array, k 1-based: the four score-bit passes of
// the radix select (ties do not matter for a value).  Used by the u8 encoder's quantile
// interval (quantile.rs:52-61, two select_nth_unstable calls in the reference).
qamd_status select_kth_f32(const float *vals_dev, uint64_t n, uint64_t k, bool largest, float *out_host,
                           hipStream_t stream) {
    if (n == 0 || k == 0 || k > n || n > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "select_kth: bad k or n");
    SelState *st = nullptr;
    QAMD_TRY(thread_ws_acquire(WS_SELECT, sizeof(SelState), stream, reinterpret_cast<void **>(&st)));
    hipLaunchKernelGGL(init_kernel, dim3(1), dim3(256), 0, stream, st, (uint32_t)k);
    uint64_t want = (n + kBlock * 8 - 1) / (kBlock * 8);
    uint64_t cap = (uint64_t)device_info().cu_count * 8;
    int grid = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    for (int shift = 56; shift >= 32; shift -= 8) {
        hipLaunchKernelGGL(hist_kernel, dim3(grid), dim3(kBlock), 0, stream, vals_dev, n, shift, largest, st);
        hipLaunchKernelGGL(pick_kernel, dim3(1), dim3(256), 0, stream, st, shift);
    }
    unsigned long long prefix = 0;
    qamd_status r = hipGetLastError() == hipSuccess ? QAMD_OK : fail(QAMD_ERR_DEVICE, "select_kth launch failed");
    if (r == QAMD_OK) r = copy_out(&prefix, QAMD_MEM_HOST, &st->prefix, 8, stream);
    thread_ws_release(WS_SELECT, stream, r == QAMD_OK);
    if (r != QAMD_OK) return r;
    uint32_t u = (uint32_t)(prefix >> 32);  // invert ordered_bits()
    if (largest) u = ~u;
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    memcpy(out_host, &u, 4);
    return QAMD_OK;
}

// ------------------------------------------------------------------------ fused scan + top-k
namespace {

struct FusedState {
    uint32_t pivot_key;  // read by every wave of the filtering scan (scalar load)
    uint32_t status;     // 0: exact result produced; 1: candidate set unusable -> caller falls back
    uint32_t total;      // candidates seen (debug)
    uint32_t pad[61];
    uint32_t counters[kTopkShards * kTopkCounterStride];  // one per shard, 256 bytes apart
};

// One workgroup picks the pivot from the S sample scores (16384 .. 131072).  ANY pivot is correct (the
// final selection is exact); it only has to let roughly `target` rows through.  Each thread
// takes the best of its S/1024 strided samples and the r-th best of those 1024 per-thread bests is
// the pivot (for r << 1024 the r best samples sit in different threads with high probability,
// so this is the sample's r-th best up to a rank or two).  The 1024 bests are bitonic-sorted
// in LDS (55 stages).  A 32-round bisection and a rank-counting loop both measured ~60 us on
// the single CU this runs on; this form is a few us.
__global__ __launch_bounds__(1024) void pivot_kernel(const float *__restrict__ sample, uint32_t S, uint32_t r,
                                                    int largest, FusedState *st) {
    __shared__ uint32_t best[1024];
    const int t = threadIdx.x;
    uint32_t mine = 0xFFFFFFFFu;
#pragma unroll 4
    for (uint32_t i = t; i < S; i += 1024u) {
        const uint32_t key = topk_ordered_bits(sample[i], largest != 0);
        mine = key < mine ? key : mine;
    }
    r = r < 1 ? 1 : (r > 1024 ? 1024 : r);
    uint32_t pivot_key;
    if (r <= 64) {
        // the usual case (fused_policy keeps r <= 64): a sort per wave and a pairwise fold of the 16 waves'
        // 64 best, in registers (21 + 4 x 6 DPP stages) instead of 55 barrier stages
        __shared__ unsigned long long lists[kSmallTopkWaves][64];
        const int lane = t & 63, wave = t >> 6;
        unsigned long long b64 = wave_sort64((unsigned long long)mine << 32, lane);
        b64 = small_topk_fold_waves(b64, lists, wave, lane);
        pivot_key = (uint32_t)(shfl_u64(b64, (int)r - 1) >> 32);  // meaningful in wave 0 only
    } else {
        best[t] = mine;
        __syncthreads();
        for (int size = 2; size <= 1024; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const int partner = t ^ stride;
                if (partner > t) {
                    const bool up = (t & size) == 0;
                    const uint32_t a = best[t], b = best[partner];
                    if ((a > b) == up) {
                        best[t] = b;
                        best[partner] = a;
                    }
                }
                __syncthreads();
            }
        }
        pivot_key = best[r - 1];
    }
    if (t == 0) {
        st->pivot_key = pivot_key;
        st->status = 0;
        st->total = 0;
    }
    if (t < (int)kTopkShards) st->counters[t * kTopkCounterStride] = 0;
}

// One workgroup of 1024 threads: bitonic sort of the <= 8192 candidates in LDS, emit the k best.
__global__ __launch_bounds__(1024) void fused_emit_kernel(const unsigned long long *__restrict__ cand,
                                                         FusedState *st, uint64_t n, uint32_t k, int largest,
                                                         uint32_t *__restrict__ out_ids,
                                                         float *__restrict__ out_scores,
                                                         uint32_t *__restrict__ status_host /* mapped host word or null */) {
    __shared__ unsigned long long s[kTopkCandCap];
    __shared__ uint32_t offs[kTopkShards + 1];
    __shared__ uint32_t overflow;
    const int t = threadIdx.x;
    if (t == 0) {  // shard counts -> exclusive offsets
        uint32_t acc = 0, over = 0;
        for (uint32_t sh = 0; sh < kTopkShards; sh++) {
            const uint32_t c = st->counters[sh * kTopkCounterStride];
            over |= c > kTopkShardCap ? 1u : 0u;
            offs[sh] = acc;
            acc += c > kTopkShardCap ? kTopkShardCap : c;
        }
        offs[kTopkShards] = acc;
        overflow = over;
        st->total = acc;
    }
    __syncthreads();
    const uint32_t pushed = offs[kTopkShards];
    const uint32_t k_eff = n < k ? (uint32_t)n : k;
    if (overflow || pushed > kTopkCandCap || pushed < k_eff) {  // overflow, or the pivot cut below k rows
        if (t == 0) {
            st->status = 1;
            if (status_host) *status_host = 1;
        }
        return;
    }
    if (k <= kSmallTopkMaxK) {
        // k <= 64: every wave keeps the 64 best of its share of the candidates (wave_sort64 of 64 at a time +
        // a 6-stage merge, in registers), the 16 waves fold pairwise; the lists alias the sort buffer
        unsigned long long(*lists)[64] = reinterpret_cast<unsigned long long(*)[64]>(s);
        const int lane = t & 63, wave = t >> 6;
        unsigned long long best = ~0ull;
        for (uint32_t base = (uint32_t)wave * 64; base < pushed; base += 1024) {
            unsigned long long v = ~0ull;
            const uint32_t f = base + lane;
            if (f < pushed) {  // flat index -> (shard, slot): the last shard whose offset is <= f
                uint32_t lo = 0, hi = kTopkShards;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (offs[mid] <= f) lo = mid;
                    else hi = mid;
                }
                v = cand[lo * kTopkShardCap + (f - offs[lo])];
            }
            v = wave_sort64(v, lane);
            best = wave_merge64_rev(best, shfl_u64(v, 63 - lane), lane);
        }
        best = small_topk_fold_waves(best, lists, wave, lane);
        if (wave == 0) {
            for (uint32_t i = lane; i < k; i += 64) {
                if (i < k_eff) {
                    out_ids[i] = (uint32_t)(best & 0xFFFFFFFFull);
                    out_scores[i] = topk_score_of_key((uint32_t)(best >> 32), largest != 0);
                } else {
                    out_ids[i] = 0xFFFFFFFFu;
                    out_scores[i] = largest ? -__builtin_huge_valf() : __builtin_huge_valf();
                }
            }
            if (lane == 0 && status_host) *status_host = 0;
        }
        return;
    }
    uint32_t N = 64;  // sort only the next power of two above the candidate count
    while (N < pushed) N <<= 1;
    for (uint32_t i = t; i < N; i += 1024) s[i] = ~0ull;
    __syncthreads();
    for (uint32_t i = t; i < kTopkShards * kTopkShardCap; i += 1024) {
        const uint32_t sh = i / kTopkShardCap, j = i % kTopkShardCap;
        if (j < offs[sh + 1] - offs[sh]) s[offs[sh] + j] = cand[i];
    }
    __syncthreads();
    for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = t; i < N / 2; i += 1024) {
                // i-th compare-exchange of this stage: pair (a, a + stride)
                const uint32_t a = 2 * i - (i & (stride - 1));
                const uint32_t b = a + stride;
                const bool up = (a & size) == 0;
                const unsigned long long x = s[a], y = s[b];
                if ((x > y) == up) {
                    s[a] = y;
                    s[b] = x;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = t; i < k; i += 1024) {
        if (i < k_eff) {
            out_ids[i] = (uint32_t)(s[i] & 0xFFFFFFFFull);
            out_scores[i] = topk_score_of_key((uint32_t)(s[i] >> 32), largest != 0);
        } else {
            out_ids[i] = 0xFFFFFFFFu;
            out_scores[i] = largest ? -__builtin_huge_valf() : __builtin_huge_valf();
        }
    }
    if (t == 0 && status_host) *status_host = 0;  // (pivot_kernel set st->status = 0)
}

__global__ void sample_ids_kernel(uint32_t *ids, uint32_t S, uint64_t n) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= S) return;
    // golden-ratio scatter of j over [0, n): deterministic, decorrelated from any row order
    const unsigned long long h = (unsigned long long)j * 0x9E3779B97F4A7C15ull;
    ids[j] = (uint32_t)(((h >> 32) * n) >> 32);
}

}  // namespace

// Sample size S and pivot rank r of the fused path; false = take the classic path.
static bool fused_policy(uint64_t n, uint32_t k, uint32_t &S, uint32_t &r) {
    const bool small = n < (1u << 20);
    const double target = std::max<double>(small ? 512.0 : 2048.0, 3.0 * k);
    S = (uint32_t)std::min<double>(131072.0, std::max<double>(small ? 2048.0 : (double)kTopkSample,
                                                              round_up((uint64_t)(8.0 * (double)n / target), 1024)));
    r = (uint32_t)std::ceil((double)S * target / (double)std::max<uint64_t>(n, 1));
    // Tiny stores: sampling buys nothing (and r must stay << 1024 for the pivot rule) --
    // classic path (scores + exact radix select).
    return n >= 32768 && r <= 64;
}

qamd_status fused_topk(uint64_t n, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                       qamd_mem out_mem, hipStream_t stream, const FusedScan &scan) {
    if (k == 0) return QAMD_OK;
    if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
    // expected candidates ~ max(2048, 3k): the pivot is (about) the r-th best of the sample, so the
    // count of rows at least as good is ~ n*Beta(r, S-r+1): mean n*r/S, relative spread 1/sqrt(r).
    // The sample grows with n to keep r near 8 (a fixed 16384-row sample has r = 4 at 10M rows --
    // one overflow of the 8192 candidate slots per ~1600 queries -- and r = 1 at 100M, where the
    // filter pass was wasted almost every time).
    // Small stores (32k .. 1M rows) take the same route with a smaller sample and fewer expected
    // candidates: there the classic path's chain of ~10 tiny kernels (score array + four radix
    // passes + gather + sort) is what a top-k costs, not HBM.
    uint32_t S = 0, r = 0;
    const bool use_fused = fused_policy(n, k, S, r);
    char *ws = nullptr;
    const size_t off_state = 0, off_cand = round_up(sizeof(FusedState), 256), off_sample = off_cand + (size_t)kTopkShards * kTopkShardCap * 8,
                 off_ids = off_sample + (size_t)S * 4, off_out = off_ids + (size_t)S * 4,
                 ws_bytes = off_out + (size_t)k * 8 + 256;
    if (use_fused) {
        // Workspace (~1 MB) kept per calling thread and device (WS_FUSED): this path always
        // synchronises before it returns, so the next call may reuse it; hipMallocAsync +
        // hipFreeAsync per call cost ~70 us of a 1.2 ms top-k on this runtime.
        uint64_t *tags = nullptr;  // [0] n, [1] S, [2] off_ids + 1 of the sample ids cached in the buffer
        QAMD_TRY(thread_ws_acquire(WS_FUSED, ws_bytes, stream, reinterpret_cast<void **>(&ws), &tags));
        FusedState *st = reinterpret_cast<FusedState *>(ws + off_state);
        unsigned long long *cand = reinterpret_cast<unsigned long long *>(ws + off_cand);
        float *sample = reinterpret_cast<float *>(ws + off_sample);
        uint32_t *ids = reinterpret_cast<uint32_t *>(ws + off_ids);
        // host outputs: the emit kernel writes them into the calling thread's mapped host scratch
        // (k <= 1024 ids + 1024 scores fit), so the status read below is the call's only copy
        const HostScratch hs = out_mem == QAMD_MEM_HOST ? host_scratch() : HostScratch{};
        uint32_t *ids_dev = out_mem == QAMD_MEM_DEVICE ? out_ids
                            : hs.host               ? hs.dev
                                                    : reinterpret_cast<uint32_t *>(ws + off_out);
        float *sc_dev = out_mem == QAMD_MEM_DEVICE ? out_scores
                        : hs.host               ? reinterpret_cast<float *>(hs.dev + 1024)
                                                : reinterpret_cast<float *>(ws + off_out) + k;
        // the sample ids depend on (n, S) only: the cached workspace keeps them from call to call
        if (tags[0] != n || tags[1] != S || tags[2] != off_ids + 1) {
            hipLaunchKernelGGL(sample_ids_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, ids, S, n);
            tags[0] = n;
            tags[1] = S;
            tags[2] = off_ids + 1;
        }
        qamd_status stt = scan.score_ids(ids, S, sample, stream);
        if (stt == QAMD_OK) {
            hipLaunchKernelGGL(pivot_kernel, dim3(1), dim3(1024), 0, stream, sample, S, r, largest, st);
            TopkFilter f{&st->pivot_key, st->counters, cand, largest};
            stt = scan.scan_filter(f, stream);
        }
        uint32_t status = 1;
        if (stt == QAMD_OK) {
            // the status word comes back through the thread's mapped host scratch: one stream
            // synchronisation, no copy call
            const HostScratch hst = host_scratch();
            if (hst.host) hst.host[2048] = 1;
            hipLaunchKernelGGL(fused_emit_kernel, dim3(1), dim3(1024), 0, stream, cand, st, n, k, largest, ids_dev,
                               sc_dev, hst.host ? hst.dev + 2048 : nullptr);
            stt = hipGetLastError() == hipSuccess ? QAMD_OK : fail(QAMD_ERR_DEVICE, "fused top-k launch failed");
            if (stt == QAMD_OK) {
                if (hst.host) {
                    if (hipStreamSynchronize(stream) != hipSuccess) stt = fail(QAMD_ERR_DEVICE, "fused top-k: synchronisation failed");
                    status = hst.host[2048];
                } else {
                    stt = copy_out(&status, QAMD_MEM_HOST, &st->status, 4, stream);  // syncs the stream
                }
            }
        }
        static const bool debug_topk = dev_env("QAMD_DEBUG_TOPK") != nullptr;
        if (debug_topk) {
            struct { uint32_t pivot_key, status, total; } dbg{};
            (void)hipMemcpy(&dbg, st, sizeof dbg, hipMemcpyDeviceToHost);
            fprintf(stderr, "[qamd topk] n=%llu k=%u r=%u pivot_key=%08x candidates=%u status=%u\n",
                    (unsigned long long)n, k, r, dbg.pivot_key, dbg.total, dbg.status);
        }
        if (stt == QAMD_OK && status == 0 && out_mem == QAMD_MEM_HOST) {
            if (hs.host) {  // the stream was synchronised by the status read
                memcpy(out_ids, hs.host, (size_t)k * 4);
                memcpy(out_scores, hs.host + 1024, (size_t)k * 4);
            } else {
                stt = copy_out(out_ids, QAMD_MEM_HOST, ids_dev, (size_t)k * 4, stream);
                if (stt == QAMD_OK) stt = copy_out(out_scores, QAMD_MEM_HOST, sc_dev, (size_t)k * 4, stream);
            }
        }
        if (stt != QAMD_OK) return stt;
        if (status == 0) return QAMD_OK;
        // fall through: pivot missed (heavy ties / adversarial order) -> exact classic path
    }
    float *scores = nullptr;
    QAMD_TRY(thread_ws_acquire(WS_SCORES, std::max<uint64_t>(n, 1) * 4, stream, reinterpret_cast<void **>(&scores)));
    qamd_status st2 = scan.scan_scores(scores, stream);
    if (st2 == QAMD_OK) st2 = topk_finish(scores, n, k, largest, out_ids, out_scores, out_mem, stream);
    thread_ws_release(WS_SCORES, stream);
    return st2;
}

qamd_status fused_topk_batch(uint64_t n, uint32_t Q, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                             qamd_mem out_mem, hipStream_t stream, const BatchScan &scan) {
    if (k == 0 || Q == 0) return QAMD_OK;
    if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
    StreamBuf res;  // host outputs: results are assembled on the device, one download at the end
    uint32_t *ids_dev = out_ids;
    float *sc_dev = out_scores;
    if (out_mem == QAMD_MEM_HOST) {
        QAMD_TRY(res.alloc((size_t)Q * k * 8, stream));
        ids_dev = res.as<uint32_t>();
        sc_dev = reinterpret_cast<float *>(ids_dev + (size_t)Q * k);
    }
    uint32_t S = 0, r = 0;
    const bool use_fused = scan.filter_capable && fused_policy(n, k, S, r);
    float *scores = nullptr;  // classic path: one score array, reused query after query in stream order
    void *sel = nullptr;
    auto classic = [&](uint32_t q) -> qamd_status {
        if (!scores) {
            QAMD_TRY(thread_ws_acquire(WS_SCORES, std::max<uint64_t>(n, 1) * 4, stream, reinterpret_cast<void **>(&scores)));
            QAMD_TRY(thread_ws_acquire(WS_SELECT, round_up(topk_workspace_bytes(k), 256), stream, &sel));
        }
        QAMD_TRY(scan.scan_scores(q, scores, stream));
        return topk_f32(scores, n, k, largest != 0, ids_dev + (size_t)q * k, sc_dev + (size_t)q * k, sel, stream);
    };
    qamd_status st = QAMD_OK;
    bool small_done = false;
    if (scan.topk_small && n <= (2u << 20) && k <= kSmallTopkMaxK) {  // small stores: one launch per query, no sync
        small_done = true;
        for (uint32_t q = 0; q < Q && st == QAMD_OK; q++)
            if (!scan.topk_small(q, ids_dev + (size_t)q * k, sc_dev + (size_t)q * k, stream, st)) {
                small_done = false;  // (the first query decides: the plan depends on the store only)
                break;
            }
    }
    if (small_done) {
    } else if (!use_fused) {
        for (uint32_t q = 0; q < Q && st == QAMD_OK; q++) st = classic(q);
    } else {
        constexpr uint32_t kChunk = 32;
        const size_t off_cand = round_up(sizeof(FusedState), 256), off_sample = off_cand + (size_t)kTopkShards * kTopkShardCap * 8,
                     per = round_up(off_sample + (size_t)S * 4, 256);
        const uint32_t C = std::min(Q, kChunk);
        StreamBuf ws;
        const size_t off_ids = (size_t)C * per, off_status = off_ids + round_up((size_t)S * 4, 256);
        QAMD_TRY(ws.alloc(off_status + (size_t)C * 4, stream));
        char *base = ws.as<char>();
        uint32_t *sample_ids = reinterpret_cast<uint32_t *>(base + off_ids);
        uint32_t *status_dev = reinterpret_cast<uint32_t *>(base + off_status);
        hipLaunchKernelGGL(sample_ids_kernel, dim3((S + 255) / 256), dim3(256), 0, stream, sample_ids, S, n);
        std::vector<uint32_t> status(C);
        for (uint32_t q0 = 0; q0 < Q && st == QAMD_OK; q0 += C) {
            const uint32_t nq = std::min(C, Q - q0);
            // 1. every query of the chunk: sample scores -> pivot
            for (uint32_t j = 0; j < nq && st == QAMD_OK; j++) {
                char *slice = base + (size_t)j * per;
                FusedState *fs = reinterpret_cast<FusedState *>(slice);
                float *sample = reinterpret_cast<float *>(slice + off_sample);
                st = scan.score_ids(q0 + j, sample_ids, S, sample, stream);
                if (st != QAMD_OK) break;
                hipLaunchKernelGGL(pivot_kernel, dim3(1), dim3(1024), 0, stream, sample, S, r, largest, fs);
            }
            // 2. filtering scans: several queries per pass over the rows where the quantizer can
            for (uint32_t j = 0; j < nq && st == QAMD_OK;) {
                char *slice = base + (size_t)j * per;
                uint32_t took = 0;
                if (scan.scan_filter_multi) {
                    TopkFilterSlices sl{slice, per, offsetof(FusedState, pivot_key), offsetof(FusedState, counters), off_cand,
                                        largest};
                    took = scan.scan_filter_multi(q0 + j, nq - j, sl, stream, st);
                    if (st != QAMD_OK) break;
                }
                if (took == 0) {
                    FusedState *fs = reinterpret_cast<FusedState *>(slice);
                    TopkFilter f{&fs->pivot_key, fs->counters, reinterpret_cast<unsigned long long *>(slice + off_cand), largest};
                    st = scan.scan_filter(q0 + j, f, stream);
                    took = 1;
                }
                j += took;
            }
            // 3. per query: sort the candidates, emit, report
            for (uint32_t j = 0; j < nq && st == QAMD_OK; j++) {
                char *slice = base + (size_t)j * per;
                FusedState *fs = reinterpret_cast<FusedState *>(slice);
                unsigned long long *cand = reinterpret_cast<unsigned long long *>(slice + off_cand);
                hipLaunchKernelGGL(fused_emit_kernel, dim3(1), dim3(1024), 0, stream, cand, fs, n, k, largest,
                                   ids_dev + (size_t)(q0 + j) * k, sc_dev + (size_t)(q0 + j) * k, status_dev + j);
            }
            if (st == QAMD_OK && hipGetLastError() != hipSuccess) st = fail(QAMD_ERR_DEVICE, "batched top-k launch failed");
            if (st == QAMD_OK) st = copy_out(status.data(), QAMD_MEM_HOST, status_dev, (size_t)nq * 4, stream);  // one sync per chunk
            for (uint32_t j = 0; j < nq && st == QAMD_OK; j++)
                if (status[j] != 0) st = classic(q0 + j);  // heavy ties or an unlucky pivot: exact path
        }
    }
    if (scores) {
        thread_ws_release(WS_SCORES, stream);
        thread_ws_release(WS_SELECT, stream);
    }
    if (st == QAMD_OK && out_mem == QAMD_MEM_HOST) {
        st = copy_out(out_ids, QAMD_MEM_HOST, ids_dev, (size_t)Q * k * 4, stream);
        if (st == QAMD_OK) st = copy_out(out_scores, QAMD_MEM_HOST, sc_dev, (size_t)Q * k * 4, stream);
    } else if (st == QAMD_OK) {
        if (hipStreamSynchronize(stream) != hipSuccess) st = fail(QAMD_ERR_DEVICE, "batched top-k: synchronisation failed");
    }
    return st;
}

// Workspace + result staging in stream order; host outputs make the call synchronous.
qamd_status topk_finish(const float *scores_dev, uint64_t n, uint32_t k, int largest,
                        uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t stream) {
    if (k == 0) return QAMD_OK;
    if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
    void *ws = nullptr;
    size_t ws_bytes = round_up(topk_workspace_bytes(k), 256);
    size_t extra = out_mem == QAMD_MEM_HOST ? (size_t)k * 8 : 0;
    QAMD_TRY(thread_ws_acquire(WS_SELECT, ws_bytes + extra, stream, &ws));
    uint32_t *ids_dev = out_ids;
    float *sc_dev = out_scores;
    const HostScratch hs = out_mem == QAMD_MEM_HOST ? host_scratch() : HostScratch{};
    if (out_mem == QAMD_MEM_HOST) {
        ids_dev = hs.host ? hs.dev : reinterpret_cast<uint32_t *>(static_cast<char *>(ws) + ws_bytes);
        sc_dev = hs.host ? reinterpret_cast<float *>(hs.dev + 1024) : reinterpret_cast<float *>(ids_dev + k);
    }
    qamd_status st = topk_f32(scores_dev, n, k, largest != 0, ids_dev, sc_dev, ws, stream);
    if (st == QAMD_OK && out_mem == QAMD_MEM_HOST) {
        if (hs.host) {
            if (hipStreamSynchronize(stream) != hipSuccess) st = fail(QAMD_ERR_DEVICE, "top-k: stream synchronisation failed");
            memcpy(out_ids, hs.host, (size_t)k * 4);
            memcpy(out_scores, hs.host + 1024, (size_t)k * 4);
        } else {
            st = copy_out(out_ids, QAMD_MEM_HOST, ids_dev, (size_t)k * 4, stream);
            if (st == QAMD_OK) st = copy_out(out_scores, QAMD_MEM_HOST, sc_dev, (size_t)k * 4, stream);
        }
    }
    thread_ws_release(WS_SELECT, stream);
    return st;
}

// ------------------------------------------------------------------------ single-launch top-k
bool small_topk_plan(uint64_t n, uint32_t k, uint32_t rows_per_tile, uint32_t min_rows_per_wg, SmallTopkPlan &plan) {
    if (n == 0 || n > (2u << 20) || k == 0 || k > kSmallTopkMaxK) return false;
    uint32_t wgs = std::min<uint32_t>((uint32_t)device_info().cu_count, 256u);  // one 16-wave workgroup per CU;
                                                                                // <= 256: the last workgroup folds 16 lists per wave
    // A workgroup is worth starting only for a share of the rows that keeps its 16 waves busy for a few
    // passes (and, per caller, that outweighs what a workgroup loads before its first row): on a 10k-row
    // store 256 workgroups of 40 rows each spend their time in the cross-workgroup fold of 256 lists, and
    // under concurrent searches every such launch occupies every CU.
    const uint64_t least = round_up(std::max<uint64_t>(min_rows_per_wg, rows_per_tile), rows_per_tile);
    wgs = (uint32_t)std::min<uint64_t>(wgs, (n + least - 1) / least);
    const uint32_t per = (uint32_t)round_up((n + wgs - 1) / wgs, rows_per_tile);
    wgs = (uint32_t)((n + per - 1) / per);  // drop workgroups that would own no row
    plan.workgroups = wgs;
    plan.rows_per_wg = per;
    return true;
}

qamd_status small_topk(const SmallTopkPlan &plan, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                       qamd_mem out_mem, hipStream_t stream,
                       const std::function<qamd_status(const SmallTopk &, hipStream_t)> &launch) {
    // workspace: ticket (own 256-byte line) | per-workgroup bests | staging for host outputs without scratch
    const size_t off_best = 256, off_out = off_best + round_up((size_t)plan.workgroups * k * 8, 256),
                 bytes = off_out + (size_t)k * 8;
    char *ws = nullptr;
    uint64_t *tags = nullptr;
    QAMD_TRY(thread_ws_acquire(WS_SMALL, bytes, stream, reinterpret_cast<void **>(&ws), &tags));
    qamd_status st = QAMD_OK;
    if (tags[0] != 0x5154u) {  // fresh (re)allocation: the ticket starts at zero, the last arriver keeps it there
        if (hipMemsetAsync(ws, 0, 256, stream) != hipSuccess) st = fail(QAMD_ERR_DEVICE, "top-k: ticket reset failed");
        else tags[0] = 0x5154u;
    }
    const HostScratch hs = out_mem == QAMD_MEM_HOST ? host_scratch() : HostScratch{};
    SmallTopk p;
    p.ticket = reinterpret_cast<uint32_t *>(ws);
    p.wg_best = reinterpret_cast<unsigned long long *>(ws + off_best);
    p.k = k;
    p.largest = largest;
    p.done_flag = nullptr;
    p.done_value = 0;
    static thread_local uint32_t done_seq = 0;
    if (hs.host) {  // the kernel raises a flag in the mapped scratch behind its results: poll, do not sleep
        if (++done_seq == 0) done_seq = 1;
        hs.host[kHostDoneAt] = 0;
        p.done_flag = hs.dev + kHostDoneAt;
        p.done_value = done_seq;
    }
    p.out_ids = out_mem == QAMD_MEM_DEVICE ? out_ids : hs.host ? hs.dev : reinterpret_cast<uint32_t *>(ws + off_out);
    p.out_scores = out_mem == QAMD_MEM_DEVICE ? out_scores
                   : hs.host               ? reinterpret_cast<float *>(hs.dev + 1024)
                                           : reinterpret_cast<float *>(ws + off_out) + k;
    if (st == QAMD_OK) st = launch(p, stream);
    if (st != QAMD_OK) tags[0] = 0;  // a launch that may not have run leaves the ticket unknown: reset it next time
    if (st == QAMD_OK && out_mem == QAMD_MEM_HOST) {
        if (hs.host) {
            // A stream synchronisation costs a 20-30 us wake-up on this runtime -- as much as the whole
            // kernel on a 100k-row store.  Poll the flag the kernel writes (system-scope release) after its
            // results for up to ~2 ms, then fall back to the synchronisation (a long queue ahead of us).
            volatile uint32_t *flag = hs.host + kHostDoneAt;
            bool seen = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spin = 0; !seen; spin++) {
                seen = *flag == p.done_value;
                if (!seen && (spin & 255u) == 255u &&
                    std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
                    break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            if (!seen && hipStreamSynchronize(stream) != hipSuccess) {
                st = fail(QAMD_ERR_DEVICE, "top-k: stream synchronisation failed");
                tags[0] = 0;
            }
            if (st == QAMD_OK) {  // never hand back what an earlier query left in the scratch
                memcpy(out_ids, hs.host, (size_t)k * 4);
                memcpy(out_scores, hs.host + 1024, (size_t)k * 4);
            }
        } else {
            st = copy_out(out_ids, QAMD_MEM_HOST, p.out_ids, (size_t)k * 4, stream);
            if (st == QAMD_OK) st = copy_out(out_scores, QAMD_MEM_HOST, p.out_scores, (size_t)k * 4, stream);
        }
    }
    thread_ws_release(WS_SMALL, stream, st == QAMD_OK && out_mem == QAMD_MEM_HOST);  // host results: the kernel has finished
    return st;
}

}  // namespace qamd

extern "C" qamd_status qamd_topk_scores(const float *scores_dev, uint64_t n, uint32_t k, int largest,
                                        uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (k == 0) return QAMD_OK;
    if (!scores_dev || !out_ids || !out_scores) return qamd::fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(qamd::current_device());
    return qamd::topk_finish(scores_dev, n, k, largest, out_ids, out_scores, out_mem, qamd::as_stream(stream));
}
