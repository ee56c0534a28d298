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

}  // namespace qamd
