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
// The key of lane ^ J: distances 1 and 2 stay inside a quad (DPP quad_perm, a plain VALU move instead of
// a ds_bpermute round trip through the LDS crossbar).
template <int J> __device__ __forceinline__ unsigned long long xor_exchange(unsigned long long v) {
    if (J == 1 || J == 2) {
        constexpr int CTRL = J == 1 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
        return ((unsigned long long)hi << 32) | lo;
    }
    return shfl_xor_u64(v, J);
}
template <int J> __device__ __forceinline__ unsigned long long cmpx(unsigned long long v, int lane, bool ascending) {
    const unsigned long long o = xor_exchange<J>(v);
    const bool keep_min = ((lane & J) == 0) == ascending;
    return keep_min ? (o < v ? o : v) : (o > v ? o : v);
}
// Bitonic sort of 64 keys, one per lane, ascending by lane.
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long v, int lane) {
    v = cmpx<1>(v, lane, (lane & 2) == 0);
    v = cmpx<2>(v, lane, (lane & 4) == 0);
    v = cmpx<1>(v, lane, (lane & 4) == 0);
    v = cmpx<4>(v, lane, (lane & 8) == 0);
    v = cmpx<2>(v, lane, (lane & 8) == 0);
    v = cmpx<1>(v, lane, (lane & 8) == 0);
    v = cmpx<8>(v, lane, (lane & 16) == 0);
    v = cmpx<4>(v, lane, (lane & 16) == 0);
    v = cmpx<2>(v, lane, (lane & 16) == 0);
    v = cmpx<1>(v, lane, (lane & 16) == 0);
    v = cmpx<16>(v, lane, (lane & 32) == 0);
    v = cmpx<8>(v, lane, (lane & 32) == 0);
    v = cmpx<4>(v, lane, (lane & 32) == 0);
    v = cmpx<2>(v, lane, (lane & 32) == 0);
    v = cmpx<1>(v, lane, (lane & 32) == 0);
    v = cmpx<32>(v, lane, true);
    v = cmpx<16>(v, lane, true);
    v = cmpx<8>(v, lane, true);
    v = cmpx<4>(v, lane, true);
    v = cmpx<2>(v, lane, true);
    v = cmpx<1>(v, lane, true);
    return v;
}
// The 64 smallest keys of two ascending lists; b_rev[i] must be b[63 - i].
__device__ __forceinline__ unsigned long long wave_merge64_rev(unsigned long long a, unsigned long long b_rev, int lane) {
    unsigned long long v = b_rev < a ? b_rev : a;
    v = cmpx<32>(v, lane, true);
    v = cmpx<16>(v, lane, true);
    v = cmpx<8>(v, lane, true);
    v = cmpx<4>(v, lane, true);
    v = cmpx<2>(v, lane, true);
    v = cmpx<1>(v, lane, true);
    return v;
}
// Two independent 32-key lists per wave (lanes 0-31 and 32-63, each ascending within its half): the 32
// smallest of a half's list and an incoming list given reversed within the half (b_rev[i] = b[31 - i]).
// Nothing crosses the halves (distances <= 16), so a wave folds two lists per step.
__device__ __forceinline__ unsigned long long wave_merge32x2_rev(unsigned long long a, unsigned long long b_rev, int lane) {
    unsigned long long v = b_rev < a ? b_rev : a;
    v = cmpx<16>(v, lane, true);
    v = cmpx<8>(v, lane, true);
    v = cmpx<4>(v, lane, true);
    v = cmpx<2>(v, lane, true);
    v = cmpx<1>(v, lane, true);
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
    // host outputs through the mapped scratch: after the results, the kernel publishes `done_value`
    // here (system scope) so that the host can poll for it instead of paying a stream
    // synchronisation's wake-up latency; nullptr = no flag
    uint32_t *done_flag;
    uint32_t done_value;
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

// Tail of a *_topk_small_kernel.  Cross-workgroup hand-off in the guide's write-through form
// (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "hand-offs measured with sc1 loads"):
// the workgroup's list is stored with sc1 (agent-scope relaxed atomic stores: write-through, so no
// L2 write-back fence is needed), the storing wave drains its stores (s_waitcnt vmcnt(0)), ONE lane
// makes a relaxed agent fetch_add on the ticket; the workgroup whose add came last reads every list
// with sc1 loads -- its other waves behind the barrier that the adding wave joins.  (The first
// version used plain stores + an agent release fence and an acquire fence + plain loads: the two
// fences cost 4-8 us of a 35 us kernel.)
__device__ __forceinline__ void small_topk_finish(unsigned long long best, unsigned long long (*lists)[64],
                                                  const SmallTopk &p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k = p.k, wgs = gridDim.x;
    best = small_topk_fold_waves(best, lists, wave, lane);
    if (wave == 0) {
        if ((uint32_t)lane < k)
            __hip_atomic_store(&p.wg_best[(size_t)blockIdx.x * k + lane], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const uint32_t drawn = __hip_atomic_fetch_add(p.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lists[0][0] = drawn == wgs - 1 ? 1ull : 0ull;  // "I am last" through the one LDS array (the add has returned)
        }
    }
    __syncthreads();
    const bool last = lists[0][0] != 0ull;
    __syncthreads();
    if (!last) return;
    // every wave folds its share of the workgroups' lists (ascending, k valid keys each), then the
    // tournament.  All of a wave's lists are loaded BEFORE the first merge: sixteen dependent
    // L2 round trips, one per merge, were most of this workgroup's time.
    constexpr int kListsPerWave = 16;  // workgroups <= 256 (small_topk_plan)
    unsigned long long mine = ~0ull;
    if (k <= 32) {
        // a list holds at most 32 keys: the two halves of the wave fold two lists per step (half h takes
        // the wave's lists 2j + h), then the upper half's result is folded into the lower one
        const uint32_t half = (uint32_t)lane >> 5, r = 31 - ((uint32_t)lane & 31u);  // reversed read inside the half
        unsigned long long theirs[kListsPerWave / 2];
#pragma unroll
        for (int j = 0; j < kListsPerWave / 2; j++) {
            const uint32_t g = wave + kSmallTopkWaves * (2 * j + half);
            theirs[j] = (g < wgs && r < k) ? __hip_atomic_load(&p.wg_best[(size_t)g * k + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : ~0ull;
        }
#pragma unroll
        for (int j = 0; j < kListsPerWave / 2; j++)
            if (wave + kSmallTopkWaves * 2 * j < (int)wgs) mine = wave_merge32x2_rev(mine, theirs[j], lane);  // wave-uniform test
        const unsigned long long upper_rev = shfl_u64(mine, 63 - lane);  // lane i < 32 receives the upper half's key 31 - i
        mine = wave_merge32x2_rev(mine, upper_rev, lane);
        if (lane >= 32) mine = ~0ull;
    } else {
        unsigned long long theirs[kListsPerWave];
        const uint32_t r = 63 - lane;  // reversed read
#pragma unroll
        for (int i = 0; i < kListsPerWave; i++) {
            const uint32_t g = wave + kSmallTopkWaves * i;
            theirs[i] = (g < wgs && r < k) ? __hip_atomic_load(&p.wg_best[(size_t)g * k + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                           : ~0ull;
        }
#pragma unroll
        for (int i = 0; i < kListsPerWave; i++)
            if (wave + kSmallTopkWaves * i < (int)wgs) mine = wave_merge64_rev(mine, theirs[i], lane);
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
        if (p.done_flag) {  // wave-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's result stores have been acknowledged
            if (lane == 0) __hip_atomic_store(p.done_flag, p.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace qamd
