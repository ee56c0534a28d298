// Developer-only tuning harness (not part of the C ABI in include/): sweeps launch geometry
// and load policy of the u8 dot scan on a resident store and reports median kernel times.
// Used to choose the shipped configuration in u8.hip; see DESIGN.md "Tuning log".
#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "common.hpp"

using namespace qamd;

namespace {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ uint4 ld(const uint4 *p) {
    if (NT) {
        u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
        return make_uint4(t.x, t.y, t.z, t.w);
    }
    return *p;
}
__device__ __forceinline__ uint32_t dot16(const uint4 &a, const uint4 &b, uint32_t acc) {
    acc = __builtin_amdgcn_udot4(a.x, b.x, acc, false);
    acc = __builtin_amdgcn_udot4(a.y, b.y, acc, false);
    acc = __builtin_amdgcn_udot4(a.z, b.z, acc, false);
    acc = __builtin_amdgcn_udot4(a.w, b.w, acc, false);
    return acc;
}

// G = 16, ITERS = 3 (dim 768).  tiles_per_wave == 0: persistent grid-stride; else each wave
// owns `tiles_per_wave` consecutive tiles.
template <int UNROLL, bool NT, int BLOCK, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void tune_scan(const uint4 *__restrict__ codes,
                                                        const float *__restrict__ offsets,
                                                        const uint4 *__restrict__ qcodes,
                                                        const float *__restrict__ q_off_p, float multiplier,
                                                        uint32_t n_rows, uint32_t tiles_per_wave,
                                                        float *__restrict__ out) {
    constexpr int G = 16, ITERS = 3, RW = 4, TILE = RW * UNROLL;
    const int lane = threadIdx.x & 63, sub = lane % G, rslot = lane / G;
    const uint32_t wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * BLOCK) >> 6;
    uint4 q[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) q[it] = qcodes[sub + it * G];
    const float q_off = *q_off_p;
    uint64_t base, end, step;
    if (tiles_per_wave == 0) {
        base = (uint64_t)wave * TILE;
        end = n_rows;
        step = (uint64_t)n_waves * TILE;
    } else {
        base = (uint64_t)wave * tiles_per_wave * TILE;
        end = std::min<uint64_t>(n_rows, base + (uint64_t)tiles_per_wave * TILE);
        step = TILE;
    }
    for (; base < end; base += step) {
        uint4 v[UNROLL][ITERS];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint4 *p = codes + (base + u * RW + rslot) * 48 + sub;
#pragma unroll
            for (int it = 0; it < ITERS; it++) v[u][it] = ld<NT>(p + it * G);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            uint32_t acc = 0;
#pragma unroll
            for (int it = 0; it < ITERS; it++) acc = dot16(v[u][it], q[it], acc);
#pragma unroll
            for (int m = 1; m < G; m <<= 1) acc += __shfl_xor(acc, m, 64);
            const uint64_t row = base + u * RW + rslot;
            if (sub == 0 && row < n_rows)
                out[row] = (multiplier * (float)(int32_t)acc + q_off) + offsets[row];
        }
    }
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::vector<float> ms;
};
}  // namespace

extern "C" __attribute__((visibility("default"))) qamd_status qamd_dev_u8_sweep(
    const void *codes, const void *offsets, const void *qbuf, float multiplier, uint32_t n_rows, float *out_dev,
    int rounds, char *report, size_t cap) {
    QAMD_TRY(ensure_device(current_device()));
    const uint4 *c = static_cast<const uint4 *>(codes);
    const float *o = static_cast<const float *>(offsets);
    const uint4 *qc = reinterpret_cast<const uint4 *>(static_cast<const uint8_t *>(qbuf) + 16);
    const float *qo = static_cast<const float *>(qbuf);
    const int cu = device_info().cu_count;
    std::vector<Variant> vs;
#define ADD(UN, NT, BL, MW, TPW, BPC)                                                                     \
    do {                                                                                                  \
        constexpr int TILE = 4 * UN;                                                                      \
        uint64_t waves_needed = ((uint64_t)n_rows + TILE - 1) / TILE;                                     \
        int grid;                                                                                         \
        if (TPW == 0) grid = cu * BPC;                                                                    \
        else grid = (int)((waves_needed + (uint64_t)TPW * (BL / 64) - 1) / ((uint64_t)TPW * (BL / 64)));  \
        char nm[128];                                                                                     \
        snprintf(nm, sizeof nm, "unroll%d nt%d block%d minw%d tpw%d bpc%d grid%d", UN, NT, BL, MW, TPW, BPC, grid); \
        vs.push_back({nm, [=](hipStream_t s) {                                                            \
                          hipLaunchKernelGGL((tune_scan<UN, NT, BL, MW>), dim3(grid), dim3(BL), 0, s, c, o, qc, qo, \
                                             multiplier, n_rows, (uint32_t)TPW, out_dev);                 \
                      }, {}});                                                                            \
    } while (0)
    // round 2 of the sweep: non-persistent grids around the round-1 winner (tpw 1)
    ADD(4, true, 256, 1, 1, 0);
    ADD(4, true, 256, 1, 2, 0);
    ADD(2, true, 256, 1, 1, 0);
    ADD(2, true, 256, 1, 2, 0);
    ADD(8, true, 256, 1, 1, 0);
    ADD(8, true, 256, 1, 2, 0);
    ADD(4, true, 64, 1, 1, 0);
    ADD(4, true, 128, 1, 1, 0);
    ADD(4, true, 512, 1, 1, 0);
    ADD(4, true, 1024, 1, 1, 0);
    ADD(8, true, 512, 1, 1, 0);
    ADD(8, true, 1024, 1, 1, 0);
    ADD(2, true, 1024, 1, 1, 0);
    ADD(4, true, 256, 2, 1, 0);
    ADD(8, true, 256, 2, 1, 0);
    ADD(4, true, 1024, 1, 0, 2);
    ADD(8, true, 1024, 1, 0, 2);
    ADD(4, true, 1024, 1, 0, 1);
    ADD(8, true, 1024, 1, 0, 1);
    ADD(4, true, 256, 1, 0, 2);
    ADD(4, true, 256, 1, 0, 3);
    ADD(8, true, 256, 1, 0, 2);
    ADD(8, true, 256, 1, 0, 3);
#undef ADD
    hipEvent_t e0, e1;
    QAMD_HIP(hipEventCreate(&e0));
    QAMD_HIP(hipEventCreate(&e1));
    for (int r = 0; r < rounds; r++) {
        for (auto &v : vs) {
            v.launch(nullptr);  // warm
            QAMD_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < 5; i++) v.launch(nullptr);
            QAMD_HIP(hipEventRecord(e1, nullptr));
            QAMD_HIP(hipEventSynchronize(e1));
            float ms = 0;
            QAMD_HIP(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / 5);
        }
    }
    std::string rep;
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2], mn = v.ms.front();
        char line[256];
        snprintf(line, sizeof line, "%-58s median %.4f ms  min %.4f ms  %.0f GB/s\n", v.name.c_str(), med, mn,
                 (double)n_rows * 772.0 / (med * 1e-3) / 1e9);
        rep += line;
    }
    snprintf(report, cap, "%s", rep.c_str());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return QAMD_OK;
}
