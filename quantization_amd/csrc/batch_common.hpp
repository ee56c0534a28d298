// Pieces shared by the many-queries-at-once paths (csrc/u8_batch.hip; the binary matrix-core path): the
// candidate-filter descriptor, the integer pre-filter bound, and the per-query kernels around a filtering
// pass (pivot sample gather, pivot selection, scatter of the wave-private candidate lists, emit).
// Internal linkage: every translation unit that includes this gets its own copies.
#pragma once
#include <cmath>

#include "common.hpp"
#include "topk.hpp"
#include "topk_device.hpp"

#pragma clang fp contract(off)

namespace qamd {
namespace {

constexpr uint32_t kBatchCap = kTopkCandCap;  // candidate slots per query
constexpr uint32_t kCounterStride = 16;       // u32: one counter per 64-byte line

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct BatchFilter {
    const float *pivot_scores;       // [Qpad] pivot as a score; +-inf for padding queries (nothing passes)
    uint32_t *counters;              // [Qpad * kCounterStride]
    unsigned long long *candidates;  // [Qpad][kBatchCap]
    int largest;
    // ping-pong kernel: candidates are first appended to wave-private lists (no global atomic in
    // the GEMM's epilogue), then distributed to the per-query lists by wave_scatter_kernel
    uint4 *wave_cand;       // [n_waves][wave_cap]  {key, row, query, 0}
    uint32_t *wave_counts;  // [n_waves] entries appended (may exceed wave_cap: overflow)
    uint32_t wave_cap;
    uint32_t wave_base;     // first wave list of this launch
    uint32_t query_base;    // global index of the launch's query 0
    int *query_bounds;      // [Qpad] scratch for the query-streaming kernel's integer bounds
};


// Integer pre-filter (MODE 1/2).  The filter "score at least as good as the pivot" is, in exact
// arithmetic, s >= P_q + R_row (or <=, template LOW: when multiplier < 0 xor smallest-first) with
// P_q = (pivot_q - q_offset_q) / multiplier and R_row = -v_offset_row / multiplier.  Both are
// rounded to integers on the safe side by more than the f32 epilogue can be off (pp_bound), and
// the accumulators START at -(B_q + B_row): after the K loop "may pass" is a sign test on the
// AND (OR) of four accumulators; only groups that may pass run the exact f32 epilogue, which
// alone decides.  A wrong bound could only cost time, never a result... provided it is on the
// safe side, which is what pp_bound's slack is for:
//   |computed score - real score| <= 2^-21 (|m s| + |q_off| + |v_off|)  (four roundings), and for a
//   candidate the filter rejects, either |m s| <= 2 (|pivot| + |q_off| + |v_off|), which the 2^-19
//   terms cover, or the real score misses the pivot by more than |m s| / 2 >> that error.
constexpr float kPpLim = 536870912.0f;  // 2^29: |B_q| + |B_row| + s < 2^31 for actual_dim <= 32768
template <bool LOW>
__device__ __forceinline__ int pp_bound(float num /* pivot - q_off, or -v_off */, float mag /* |pivot|+|q_off| or |v_off| */,
                                        float m, int extra) {
    const float x = num / m;
    const float slack = 1.0f + (mag * 0x1p-19f) / fabsf(m) + fabsf(x) * 0x1p-22f;
    float t = LOW ? ceilf(x + slack) + (float)extra : floorf(x - slack);
    const float all = LOW ? kPpLim : -kPpLim;
    if (!(t == t)) t = all;  // NaN: let the exact epilogue decide
    t = fminf(fmaxf(t, -kPpLim), kPpLim);
    return (int)t;
}


// Gather `n` sampled rows (codes + offsets) into a dense sub-store for the pivot pass, followed by
// `pad` zero rows (the GEMM kernels read whole row tiles).
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4 *__restrict__ codes,
                                                         const float *__restrict__ offsets, uint32_t row_chunks,
                                                         uint64_t n_rows, uint32_t n, uint32_t pad,
                                                         uint4 *__restrict__ out_codes,
                                                         float *__restrict__ out_offsets) {
    const uint64_t total = (uint64_t)(n + pad) * row_chunks;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t j = (uint32_t)(i / row_chunks), c = (uint32_t)(i % row_chunks);
        if (j >= n) {
            out_codes[i] = make_uint4(0, 0, 0, 0);
            if (c == 0) out_offsets[j] = 0.0f;
            continue;
        }
        const unsigned long long hsh = (unsigned long long)j * 0x9E3779B97F4A7C15ull;
        const uint64_t src = ((hsh >> 32) * n_rows) >> 32;  // same golden-ratio scatter as topk.hip
        out_codes[i] = codes[src * row_chunks + c];
        if (c == 0) out_offsets[j] = offsets[src];
    }
}

// Wave-private candidate lists -> per-query lists (grid = wave lists).  A list that overflowed
// sets *overflow: the caller then redoes every query exactly.
__global__ __launch_bounds__(256) void wave_scatter_kernel(const uint4 *__restrict__ wave_cand,
                                                          const uint32_t *__restrict__ wave_counts, uint32_t wave_cap,
                                                          uint32_t *__restrict__ counters,
                                                          unsigned long long *__restrict__ candidates,
                                                          uint32_t *__restrict__ overflow) {
    const uint32_t w = blockIdx.x;
    uint32_t count = wave_counts[w];
    if (count > wave_cap) {
        if (threadIdx.x == 0) *overflow = 1;
        count = wave_cap;
    }
    const uint4 *list = wave_cand + (uint64_t)w * wave_cap;
    for (uint32_t i = threadIdx.x; i < count; i += 256) {
        const uint4 c = list[i];
        const uint32_t pos = atomicAdd(counters + (uint64_t)c.z * kCounterStride, 1u);
        if (pos < kBatchCap) candidates[(uint64_t)c.z * kBatchCap + pos] = ((unsigned long long)c.x << 32) | c.y;
    }
}

// The same with the global atomics aggregated: a workgroup takes a range of wave lists, counts its entries per
// query in LDS, reserves one range per (workgroup, query) with ONE global atomic and places the entries by
// LDS ranks.  Same-address global atomics serialise at ~25 ns each: one per candidate was 15-50 us of every
// call (1024-2048 per query); this way a query's counter sees one add per workgroup.  n_queries <= 4096.
__global__ __launch_bounds__(1024) void wave_scatter_grouped_kernel(const uint4 *__restrict__ wave_cand,
                                                                   const uint32_t *__restrict__ wave_counts,
                                                                   uint32_t wave_cap, uint32_t n_lists, uint32_t n_queries,
                                                                   uint32_t *__restrict__ counters,
                                                                   unsigned long long *__restrict__ candidates,
                                                                   uint32_t *__restrict__ overflow) {
    extern __shared__ uint32_t scatter_lds[];
    uint32_t *hist = scatter_lds, *base = scatter_lds + n_queries;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (uint32_t i = t; i < n_queries; i += 1024) hist[i] = 0;
    __syncthreads();
    const uint32_t l0 = (uint32_t)((uint64_t)blockIdx.x * n_lists / gridDim.x);
    const uint32_t l1 = (uint32_t)((uint64_t)(blockIdx.x + 1) * n_lists / gridDim.x);
    for (uint32_t l = l0 + wave; l < l1; l += 16) {  // one wave per list
        uint32_t count = wave_counts[l];
        if (count > wave_cap) {
            if (lane == 0) *overflow = 1;
            count = wave_cap;
        }
        const uint4 *list = wave_cand + (uint64_t)l * wave_cap;
        for (uint32_t i = lane; i < count; i += 64) atomicAdd(&hist[list[i].z], 1u);
    }
    __syncthreads();
    for (uint32_t i = t; i < n_queries; i += 1024) {
        const uint32_t cnt = hist[i];
        base[i] = cnt ? atomicAdd(counters + (uint64_t)i * kCounterStride, cnt) : 0u;
        hist[i] = 0;
    }
    __syncthreads();
    for (uint32_t l = l0 + wave; l < l1; l += 16) {
        const uint32_t count = min(wave_counts[l], wave_cap);
        const uint4 *list = wave_cand + (uint64_t)l * wave_cap;
        for (uint32_t i = lane; i < count; i += 64) {
            const uint4 c = list[i];
            const uint32_t pos = base[c.z] + atomicAdd(&hist[c.z], 1u);
            if (pos < kBatchCap) candidates[(uint64_t)c.z * kBatchCap + pos] = ((unsigned long long)c.x << 32) | c.y;
        }
    }
}

// Per-query pivot (grid = queries): r-th best of 1024 per-thread bests of the query's sample
// scores (see pivot_kernel in topk.hip), and counter reset.
__global__ __launch_bounds__(1024) void batch_pivot_kernel(const float *__restrict__ sample, uint32_t S,
                                                          uint64_t pitch, uint32_t r, int largest,
                                                          uint32_t n_queries, float *__restrict__ pivot_scores,
                                                          uint32_t *__restrict__ counters) {
    __shared__ unsigned long long lists[kSmallTopkWaves][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (blockIdx.x >= n_queries) {  // padding query of the last tile: nothing may pass
        if (t == 0) pivot_scores[blockIdx.x] = largest ? __builtin_huge_valf() : -__builtin_huge_valf();
        return;
    }
    const float *mine_row = sample + (uint64_t)blockIdx.x * pitch;
    uint32_t mine = 0xFFFFFFFFu;
    for (uint32_t i = t; i < S; i += 1024) {
        const uint32_t key = topk_ordered_bits(mine_row[i], largest != 0);
        mine = key < mine ? key : mine;
    }
    // the r-th best (r <= 64) of the 1024 per-thread bests: a sort per wave, the 16 waves folded pairwise
    // (21 + 4 x 6 register stages instead of the 55 barrier stages of a 1024-key bitonic sort)
    unsigned long long best = wave_sort64((unsigned long long)mine << 32, lane);
    best = small_topk_fold_waves(best, lists, wave, lane);
    r = r < 1 ? 1 : (r > 64 ? 64 : r);
    if (wave == 0 && lane == (int)r - 1) {
        pivot_scores[blockIdx.x] = topk_score_of_key((uint32_t)(best >> 32), largest != 0);
        counters[(uint64_t)blockIdx.x * kCounterStride] = 0;
    }
}

// Per-query emit (grid = queries): sort the query's candidates, write its k best; status 1
// when the list over- or under-flowed (that query is redone exactly by the caller).
__global__ __launch_bounds__(1024) void batch_emit_kernel(const unsigned long long *__restrict__ cand,
                                                         const uint32_t *__restrict__ counters, uint64_t n,
                                                         uint32_t k, int largest, uint32_t *__restrict__ out_ids,
                                                         float *__restrict__ out_scores,
                                                         uint32_t *__restrict__ status) {
    __shared__ unsigned long long s[kBatchCap];
    const int t = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t pushed = counters[(uint64_t)q * kCounterStride];
    const uint32_t k_eff = n < k ? (uint32_t)n : k;
    if (pushed > kBatchCap || pushed < k_eff) {
        if (t == 0) status[q] = 1;
        return;
    }
    uint32_t N = 64;
    while (N < pushed) N <<= 1;
    const unsigned long long *mine = cand + (uint64_t)q * kBatchCap;
    for (uint32_t i = t; i < N; i += 1024) s[i] = i < pushed ? mine[i] : ~0ull;
    __syncthreads();
    for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = t; i < N / 2; i += 1024) {
                const uint32_t a = 2 * i - (i & (stride - 1));
                const uint32_t bb = a + stride;
                const bool up = (a & size) == 0;
                const unsigned long long x = s[a], y = s[bb];
                if ((x > y) == up) {
                    s[a] = y;
                    s[bb] = x;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = t; i < k; i += 1024) {
        if (i < k_eff) {
            out_ids[(uint64_t)q * k + i] = (uint32_t)(s[i] & 0xFFFFFFFFull);
            out_scores[(uint64_t)q * k + i] = topk_score_of_key((uint32_t)(s[i] >> 32), largest != 0);
        } else {
            out_ids[(uint64_t)q * k + i] = 0xFFFFFFFFu;
            out_scores[(uint64_t)q * k + i] = largest ? -__builtin_huge_valf() : __builtin_huge_valf();
        }
    }
    if (t == 0) status[q] = 0;
}

// The same for k <= 64 without the full sort: every wave keeps the 64 best of its share of the candidates
// (wave_sort64 of 64 at a time + a 6-stage merge, all in registers), the 16 waves fold pairwise through
// LDS: ~30 DPP/LDS stages instead of the 66-91 barrier stages of the bitonic sort above (15-40 us per
// call whatever the batch size, since the queries' workgroups run side by side).
__global__ __launch_bounds__(1024) void batch_emit_wave_kernel(const unsigned long long *__restrict__ cand,
                                                              const uint32_t *__restrict__ counters, uint64_t n,
                                                              uint32_t k, int largest, uint32_t *__restrict__ out_ids,
                                                              float *__restrict__ out_scores,
                                                              uint32_t *__restrict__ status) {
    __shared__ unsigned long long lists[kSmallTopkWaves][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t q = blockIdx.x;
    const uint32_t pushed = counters[(uint64_t)q * kCounterStride];
    const uint32_t k_eff = n < k ? (uint32_t)n : k;
    if (pushed > kBatchCap || pushed < k_eff) {
        if (t == 0) status[q] = 1;
        return;
    }
    const unsigned long long *mine = cand + (uint64_t)q * kBatchCap;
    unsigned long long best = ~0ull;
    for (uint32_t base = (uint32_t)wave * 64; base < pushed; base += 1024) {
        unsigned long long v = base + lane < pushed ? mine[base + lane] : ~0ull;
        v = wave_sort64(v, lane);
        best = wave_merge64_rev(best, shfl_u64(v, 63 - lane), lane);
    }
    best = small_topk_fold_waves(best, lists, wave, lane);
    if (wave == 0) {
        for (uint32_t i = lane; i < k; i += 64) {
            if (i < k_eff) {
                out_ids[(uint64_t)q * k + i] = (uint32_t)(best & 0xFFFFFFFFull);
                out_scores[(uint64_t)q * k + i] = topk_score_of_key((uint32_t)(best >> 32), largest != 0);
            } else {
                out_ids[(uint64_t)q * k + i] = 0xFFFFFFFFu;
                out_scores[(uint64_t)q * k + i] = largest ? -__builtin_huge_valf() : __builtin_huge_valf();
            }
        }
        if (lane == 0) status[q] = 0;
    }
}


// waves (= candidate lists) of one launch of a persistent 8-wave-per-CU matrix-core kernel
inline uint32_t pp_waves_per_launch() { return (uint32_t)std::max(1, device_info().cu_count / 8) * 8 * 8; }

}  // namespace
}  // namespace qamd
