// Device-side pieces of the fused scan+top-k path, shared by the scan kernels (u8/bin/pq) and
// topk.hip.  A scan launched in FILTER mode does not write scores: the lane that owns a row's
// score compares its order-preserving key with a pivot and appends (key << 32 | row) to a
// small candidate buffer.  The pivot comes from a sample of the same store, so ~2-4 k rows pass.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace qamd {

constexpr uint32_t kTopkCandCap = 8192;   // most candidates one LDS sort takes (64 KiB of keys)
constexpr uint32_t kTopkShards = 64;      // candidate lists, one counter each (on its own 256-B line)
constexpr uint32_t kTopkShardCap = 256;   // slots per shard
constexpr uint32_t kTopkCounterStride = 64;  // in u32: counters 256 bytes apart

// Why shards: with ONE counter the ~3000 returning atomics of a 10M-row scan all hit one
// address; they serialise at the memory side (~45 ns each) and the channel that owns the line
// stalls the streaming reads behind them — the filtering scan measured 1.20 ms instead of
// 1.06 ms.  64 counters on 64 lines cut that to the noise.
struct TopkFilter {
    const uint32_t *pivot_key;       // device: order-preserving key of the pivot score
    uint32_t *counters;              // device: kTopkShards counters, kTopkCounterStride apart
    unsigned long long *candidates;  // device: kTopkShards * kTopkShardCap slots
    int largest;
};

// FILTER mode for several queries in one scan (binary: bin_scan_multi_kernel): query j's pivot,
// shard counters and candidate slots sit in slice j of one workspace, `stride` bytes apart.
struct TopkFilterSlices {
    char *base;           // slice of the first query
    size_t stride;        // bytes between two queries' slices
    size_t off_pivot;     // u32 pivot key inside a slice
    size_t off_counters;  // kTopkShards counters, kTopkCounterStride apart
    size_t off_cand;      // kTopkShards * kTopkShardCap candidate slots
    int largest;
};
__device__ __forceinline__ TopkFilter topk_filter_of(const TopkFilterSlices &s, uint32_t j) {
    char *sl = s.base + (size_t)j * s.stride;
    return TopkFilter{reinterpret_cast<const uint32_t *>(sl + s.off_pivot), reinterpret_cast<uint32_t *>(sl + s.off_counters),
                      reinterpret_cast<unsigned long long *>(sl + s.off_cand), s.largest};
}

// Ascending total order on f32 bit patterns; `largest` flips it so that "best" == smallest key.
__device__ __forceinline__ uint32_t topk_ordered_bits(float f, bool largest) {
    uint32_t u = __float_as_uint(f);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return largest ? ~u : u;
}
__device__ __forceinline__ float topk_score_of_key(uint32_t key, bool largest) {
    uint32_t u = largest ? ~key : key;
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}

__device__ __forceinline__ void topk_offer(const TopkFilter &f, uint32_t pivot, float score, uint32_t row) {
    const uint32_t key = topk_ordered_bits(score, f.largest != 0);
    if (key <= pivot) {
        const uint32_t shard = blockIdx.x & (kTopkShards - 1);
        const uint32_t pos = atomicAdd(f.counters + shard * kTopkCounterStride, 1u);
        if (pos < kTopkShardCap)
            f.candidates[shard * kTopkShardCap + pos] = ((unsigned long long)key << 32) | row;
    }
}

// ---------------------------------------------------------------------------------------------
// Single-launch top-k of a small store (u8.hip / bin.hip / pq.hip `*_topk_small_kernel`).
//
// Everything is built from ONE primitive that needs no workgroup barrier: a wave holds 64 keys
// sorted ascending, one per lane ("a list"; ~0 = empty slot), and merge64(a, b) keeps the 64
// smallest of two lists: t[i] = min(a[i], b[63 - i]) is bitonic and holds exactly those keys, six
// compare-exchange steps across lanes sort it.  A wave streams its rows through a 64-slot LDS
// staging row (keys are produced by every G-th lane only), sorts each full row across lanes and
// merges it into its running list; the 16 waves of a workgroup, then the workgroups (last arriver,
// release / ticket / acquire), are folded with the same merge in a binary tournament.
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64);
    const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// Bitonic sort of 64 keys, one per lane, ascending by lane.
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long v, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j > 0; j >>= 1) {
            const unsigned long long o = shfl_xor_u64(v, j);
            const bool keep_min = ((lane & j) == 0) == ((lane & size) == 0);
            v = keep_min ? (o < v ? o : v) : (o > v ? o : v);
        }
    }
    return v;
}
// The 64 smallest keys of two ascending lists; b_rev[i] must be b[63 - i].
__device__ __forceinline__ unsigned long long wave_merge64_rev(unsigned long long a, unsigned long long b_rev, int lane) {
    unsigned long long v = b_rev < a ? b_rev : a;
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const unsigned long long o = shfl_xor_u64(v, j);
        v = ((lane & j) == 0) ? (o < v ? o : v) : (o > v ? o : v);
    }
    return v;
}

constexpr int kSmallTopkWaves = 16;  // 1024-thread workgroups
constexpr uint32_t kSmallTopkMaxK = 64;

struct SmallTopk {
    unsigned long long *wg_best;  // [workgroups][k] keys, best first
    uint32_t *ticket;             // zero between launches (the last arriver resets it)
    uint32_t *out_ids;            // [k]
    float *out_scores;            // [k]
    uint32_t k;
    int largest;
};

// Per-wave accumulator: feed keys through `stage` (this wave's 64 LDS slots), flush when full.
struct SmallTopkWave {
    unsigned long long best = ~0ull;  // lane i: the i-th smallest key seen so far
    uint32_t fill = 0;                // wave-uniform: slots of `stage` written since the last flush
};
__device__ __forceinline__ void small_topk_flush(SmallTopkWave &w, unsigned long long *stage, int lane) {
    // same-wave LDS accesses are served in order: the reads below see the writes of every lane
    unsigned long long v = (uint32_t)lane < w.fill ? stage[lane] : ~0ull;
    v = wave_sort64(v, lane);
    w.best = wave_merge64_rev(w.best, shfl_u64(v, 63 - lane), lane);
    w.fill = 0;
}

// Folds the 16 per-wave lists of a workgroup (binary tournament through `lists`, [16][64] in LDS);
// returns wave 0's list = the workgroup's 64 best.  Every wave must call it.
__device__ __forceinline__ unsigned long long small_topk_fold_waves(unsigned long long best, unsigned long long (*lists)[64],
                                                                    int wave, int lane) {
    lists[wave][lane] = best;
    __syncthreads();
#pragma unroll
    for (int step = 1; step < kSmallTopkWaves; step <<= 1) {
        if ((wave & (2 * step - 1)) == 0) {
            best = wave_merge64_rev(best, lists[wave + step][63 - lane], lane);
            lists[wave][lane] = best;
        }
        __syncthreads();
    }
    return best;
}

// Tail of a *_topk_small_kernel.  Cross-workgroup visibility follows the release / ticket / acquire
// recipe of cdna_hip_programming.md (Guideline 16, counter form): the publishing wave drains its
// stores, lane 0 runs an agent-scope release + drain and a relaxed agent fetch_add; the workgroup
// that draws the last ticket runs one agent-scope acquire + drain + barrier before its plain loads.
__device__ __forceinline__ void small_topk_finish(unsigned long long best, unsigned long long (*lists)[64],
                                                  const SmallTopk &p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k = p.k, wgs = gridDim.x;
    best = small_topk_fold_waves(best, lists, wave, lane);
    if (wave == 0) {
        if ((uint32_t)lane < k) p.wg_best[(size_t)blockIdx.x * k + lane] = best;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t last = 0;
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t drawn = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = drawn == wgs - 1 ? 1u : 0u;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lists[0][0] = last;  // "I am last" through the one LDS array
        }
    }
    __syncthreads();
    const bool last = lists[0][0] != 0ull;
    __syncthreads();
    if (!last) return;
    // every wave folds its share of the workgroups' lists (ascending, k valid keys each), then the tournament
    unsigned long long mine = ~0ull;
    for (uint32_t g = wave; g < wgs; g += kSmallTopkWaves) {
        const uint32_t r = 63 - lane;  // reversed read
        const unsigned long long o = r < k ? p.wg_best[(size_t)g * k + r] : ~0ull;
        mine = wave_merge64_rev(mine, o, lane);
    }
    mine = small_topk_fold_waves(mine, lists, wave, lane);
    if (wave == 0) {
        if ((uint32_t)lane < k) {
            if (mine != ~0ull) {
                p.out_ids[lane] = (uint32_t)(mine & 0xFFFFFFFFull);
                p.out_scores[lane] = topk_score_of_key((uint32_t)(mine >> 32), p.largest != 0);
            } else {  // fewer rows than k: pad with the worst possible entry
                p.out_ids[lane] = 0xFFFFFFFFu;
                p.out_scores[lane] = p.largest ? -__builtin_huge_valf() : __builtin_huge_valf();
            }
        }
        if (lane == 0) __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
    }
}

}  // namespace qamd
