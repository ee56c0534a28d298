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
template <int UNROLL, bool NT, int BLOCK, int MINW, int NTS = 0>
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
        float mine = 0.0f;
        const float my_off = (NTS >= 2 && sub < UNROLL) ? offsets[base + sub * RW + rslot] : 0.0f;
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            uint32_t acc = 0;
#pragma unroll
            for (int it = 0; it < ITERS; it++) acc = dot16(v[u][it], q[it], acc);
#pragma unroll
            for (int m = 1; m < G; m <<= 1) acc += __shfl_xor(acc, m, 64);
            const uint64_t row = base + u * RW + rslot;
            if (NTS < 2) {
                if (sub == 0 && row < n_rows) {
                    const float r = (multiplier * (float)(int32_t)acc + q_off) + offsets[row];
                    if (NTS == 1) __builtin_nontemporal_store(r, out + row);
                    else out[row] = r;
                }
            } else if (sub == u) {
                mine = (multiplier * (float)(int32_t)acc + q_off) + my_off;
            }
        }
        if (NTS >= 2 && sub < UNROLL) {  // one store instruction: RW*UNROLL consecutive rows
            const uint64_t row = base + sub * RW + rslot;
            if (row < n_rows) {
                if (NTS == 3 || NTS == 5) __builtin_nontemporal_store(mine, out + row);
                else if (NTS == 2) out[row] = mine;
                if (NTS >= 4) {  // filter experiment: pivot in out[n_rows + 1] (u32), counter at out[n_rows + 64]
                    const uint32_t piv = reinterpret_cast<const uint32_t *>(out)[n_rows + 1];
                    uint32_t key = __float_as_uint(mine);
                    key ^= (key >> 31) ? 0xFFFFFFFFu : 0x80000000u;
                    key = ~key;
                    if (key <= piv) atomicAdd(reinterpret_cast<uint32_t *>(out) + n_rows + 64, 1u);
                }
            }
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
    QAMD_ON_DEVICE(current_device());
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
#define ADDS(UN, NT, BL, MW, TPW, BPC, SM)                                                                     \
    do {                                                                                                  \
        constexpr int TILE = 4 * UN;                                                                      \
        uint64_t waves_needed = ((uint64_t)n_rows + TILE - 1) / TILE;                                     \
        int grid;                                                                                         \
        if (TPW == 0) grid = cu * BPC;                                                                    \
        else grid = (int)((waves_needed + (uint64_t)TPW * (BL / 64) - 1) / ((uint64_t)TPW * (BL / 64)));  \
        char nm[128];                                                                                     \
        snprintf(nm, sizeof nm, "store%d unroll%d nt%d block%d minw%d tpw%d bpc%d grid%d", SM, UN, NT, BL, MW, TPW, BPC, grid); \
        vs.push_back({nm, [=](hipStream_t s) {                                                            \
                          hipLaunchKernelGGL((tune_scan<UN, NT, BL, MW, SM>), dim3(grid), dim3(BL), 0, s, c, o, qc, qo, \
                                             multiplier, n_rows, (uint32_t)TPW, out_dev);                 \
                      }, {}});                                                                            \
    } while (0)
    // round 5: does dropping the score store (fused top-k filter mode) cost time?
    ADDS(4, true, 512, 1, 1, 0, 3);
    ADDS(4, true, 512, 1, 1, 0, 4);
    ADDS(4, true, 512, 1, 1, 0, 5);
#undef ADD
#undef ADDS
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

// ---------------------------------------------------------------- row streaming access patterns
// Pure-load kernels for the design of u8_gemm_rs_kernel: a wave streams 64-row chunks of the u8 store
// K-block (128 B per row) after K-block, the way the MFMA B operand wants them, and only XORs the
// bytes.  PATTERN 0: lane (r, h) takes the 64 contiguous bytes [64h, +64) of rows r and r + 32 (four
// dwordx4 each: the direct-to-register operand layout).  PATTERN 1: fully coalesced (8 lanes per
// 128-byte line, 8 rows per instruction).  AHEAD = K-blocks requested before the one being consumed.
namespace {
typedef int sv4 __attribute__((ext_vector_type(4)));
template <int PATTERN, bool NT, int AHEAD, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void tune_stream(const uint8_t *__restrict__ codes, uint32_t n_rows, uint32_t ad,
                                                          uint32_t *__restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t nkb = (ad + 127) / 128;
    const uint32_t n_chunks = (n_rows + 63) / 64;
    const uint32_t stride = gridDim.x * WAVES;
    uint32_t chunk = blockIdx.x * WAVES + wave;
    auto ld = [&](const uint8_t *p) {
        return NT ? __builtin_nontemporal_load(reinterpret_cast<const sv4 *>(p)) : *reinterpret_cast<const sv4 *>(p);
    };
    sv4 acc = {0, 0, 0, 0};
    sv4 buf[AHEAD + 1][8];
    uint32_t pf_chunk = chunk, pf_kb = 0;
    auto issue = [&](sv4(&b)[8]) {
        const uint32_t c = pf_chunk < n_chunks ? pf_chunk : n_chunks - 1;
        if (PATTERN == 0) {
            const uint8_t *p = codes + ((uint64_t)c * 64 + (lane & 31)) * ad + pf_kb * 128 + 64 * (lane >> 5);
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int x = 0; x < 4; x++) b[jj * 4 + x] = ld(p + (uint64_t)jj * 32 * ad + 16 * x);
        } else {
            const uint8_t *p = codes + ((uint64_t)c * 64 + (lane >> 3)) * ad + pf_kb * 128 + 16 * (lane & 7);
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = ld(p + (uint64_t)i * 8 * ad);
        }
        if (++pf_kb == nkb) {
            pf_kb = 0;
            pf_chunk += stride;
        }
    };
#pragma unroll
    for (int a = 0; a < AHEAD; a++) issue(buf[a]);
    for (; chunk < n_chunks; chunk += stride) {
        for (uint32_t kb = 0; kb < nkb; kb += AHEAD + 1) {
#pragma unroll
            for (int a = 0; a <= AHEAD; a++) {
                if (kb + a < nkb) {  // nkb is a multiple of AHEAD + 1 in the sweep (768 = 6 x 128)
                    issue(buf[(a + AHEAD) % (AHEAD + 1)]);
#pragma unroll
                    for (int i = 0; i < 8; i++) acc ^= buf[a][i];
                }
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) sink[0] = 1;  // keeps the loads alive
}
}  // namespace

extern "C" __attribute__((visibility("default"))) qamd_status qamd_dev_stream_sweep(const void *codes, uint32_t n_rows,
                                                                                  uint32_t ad, uint32_t *sink,
                                                                                  int rounds, char *report, size_t cap) {
    QAMD_ON_DEVICE(current_device());
    const uint8_t *c = static_cast<const uint8_t *>(codes);
    const int cu = device_info().cu_count;
    std::vector<Variant> vs;
#define ADDV(P, NT, AH, W, BPC)                                                                                          \
    do {                                                                                                                 \
        char nm[128];                                                                                                    \
        snprintf(nm, sizeof nm, "pattern%d nt%d ahead%d waves%d blocks/cu%d", P, NT, AH, W, BPC);                        \
        vs.push_back({nm, [=](hipStream_t s) {                                                                           \
                          hipLaunchKernelGGL((tune_stream<P, NT, AH, W>), dim3(cu * BPC), dim3(64 * W), 0, s, c, n_rows, \
                                             ad, sink);                                                                  \
                      }, {}});                                                                                           \
    } while (0)
    ADDV(0, false, 1, 8, 1);
    ADDV(0, true, 1, 8, 1);
    ADDV(0, false, 2, 8, 1);
    ADDV(0, true, 2, 8, 1);
    ADDV(0, true, 1, 16, 1);
    ADDV(0, true, 2, 16, 1);
    ADDV(0, true, 1, 8, 2);
    ADDV(1, false, 1, 8, 1);
    ADDV(1, true, 1, 8, 1);
    ADDV(1, true, 2, 8, 1);
    ADDV(1, true, 1, 16, 1);
    ADDV(1, true, 2, 16, 1);
    ADDV(1, true, 1, 8, 2);
#undef ADDV
    hipEvent_t e0, e1;
    QAMD_HIP(hipEventCreate(&e0));
    QAMD_HIP(hipEventCreate(&e1));
    for (int r = 0; r < rounds; r++) {
        for (auto &v : vs) {
            v.launch(nullptr);
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
        snprintf(line, sizeof line, "%-44s median %.4f ms  min %.4f ms  %.0f GB/s\n", v.name.c_str(), med, mn,
                 (double)n_rows * ad / (med * 1e-3) / 1e9);
        rep += line;
    }
    snprintf(report, cap, "%s", rep.c_str());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return QAMD_OK;
}

// ---------------------------------------------------------------- int8 MFMA issue-rate ceiling
// What the matrix pipe delivers on this box with nothing else in the way: every wave keeps NACC
// independent 32x32 accumulators and issues v_mfma_i32_32x32x32_i8 back to back on operands held in
// registers (pseudo-random bytes, so the multipliers toggle like real codes), WPS waves per SIMD.
namespace {
typedef int mv4 __attribute__((ext_vector_type(4)));
typedef int mv16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(512) void tune_mfma_peak(uint32_t iters, uint32_t seed, int *__restrict__ sink) {
    mv16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[a][e] = 0;
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    auto rnd = [&]() {
        x = x * 1664525u + 1013904223u;
        return (int)(x & 0x7F7F7F7Fu);  // codes <= 127
    };
    mv4 a0 = {rnd(), rnd(), rnd(), rnd()}, a1 = {rnd(), rnd(), rnd(), rnd()};
    mv4 b0 = {rnd(), rnd(), rnd(), rnd()}, b1 = {rnd(), rnd(), rnd(), rnd()};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < NACC; a++)
            acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8((a & 1) ? a1 : a0, (a & 2) ? b1 : b0, acc[a], 0, 0, 0);
        // rotate the operands a little between rounds (two VALU ops per NACC MFMAs)
        a0.x ^= (int)it;
        b1.y ^= (int)(it << 3);
    }
    int t = 0;
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int e = 0; e < 16; e++) t ^= acc[a][e];
    if (t == 0x7fffffff) sink[0] = t;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {  // shader clocks and 10 ns ticks of the loop: the clock it ran at
        reinterpret_cast<unsigned long long *>(sink)[1] = __builtin_amdgcn_s_memtime() - c0;
        reinterpret_cast<unsigned long long *>(sink)[2] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
// the same loop on v_mfma_i32_16x16x64_i8 (half the ops per instruction, 4 accumulator registers instead of 16):
// the same arithmetic per cycle on paper - does the part clock it differently?
template <int NACC>
__global__ __launch_bounds__(512) void tune_mfma_peak16(uint32_t iters, uint32_t seed, int *__restrict__ sink) {
    mv4 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[a] = mv4{0, 0, 0, 0};
    uint32_t x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    auto rnd = [&]() {
        x = x * 1664525u + 1013904223u;
        return (int)(x & 0x7F7F7F7Fu);
    };
    mv4 a0 = {rnd(), rnd(), rnd(), rnd()}, a1 = {rnd(), rnd(), rnd(), rnd()};
    mv4 b0 = {rnd(), rnd(), rnd(), rnd()}, b1 = {rnd(), rnd(), rnd(), rnd()};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < NACC; a++)
            acc[a] = __builtin_amdgcn_mfma_i32_16x16x64_i8((a & 1) ? a1 : a0, (a & 2) ? b1 : b0, acc[a], 0, 0, 0);
        a0.x ^= (int)it;
        b1.y ^= (int)(it << 3);
    }
    int t = 0;
#pragma unroll
    for (int a = 0; a < NACC; a++)
#pragma unroll
        for (int e = 0; e < 4; e++) t ^= acc[a][e];
    if (t == 0x7fffffff) sink[0] = t;
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
        reinterpret_cast<unsigned long long *>(sink)[1] = __builtin_amdgcn_s_memtime() - c0;
        reinterpret_cast<unsigned long long *>(sink)[2] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}
}  // namespace

extern "C" __attribute__((visibility("default"))) qamd_status qamd_dev_mfma_peak(int *sink, char *report, size_t cap) {
    QAMD_ON_DEVICE(current_device());
    const int cu = device_info().cu_count;
    hipEvent_t e0, e1;
    QAMD_HIP(hipEventCreate(&e0));
    QAMD_HIP(hipEventCreate(&e1));
    std::string rep;
    auto run = [&](const char *name, int threads, int nacc, auto kernel, double ops_per_mfma = 32.0 * 32.0 * 32.0 * 2.0) -> qamd_status {
        const uint32_t iters = 20000;
        std::vector<float> ms;
        for (int r = 0; r < 5; r++) {
            QAMD_HIP(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(kernel, dim3(cu), dim3(threads), 0, nullptr, iters, 12345u + r, sink);
            QAMD_HIP(hipEventRecord(e1, nullptr));
            QAMD_HIP(hipEventSynchronize(e1));
            float t = 0;
            QAMD_HIP(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double ops = (double)cu * (threads / 64) * (double)iters * nacc * ops_per_mfma;
        unsigned long long clk[3] = {0, 0, 0};
        QAMD_HIP(hipMemcpy(clk, sink, sizeof clk, hipMemcpyDeviceToHost));
        const double mhz = clk[2] ? (double)clk[1] / (double)clk[2] * 100.0 : 0.0;
        char line[256];
        snprintf(line, sizeof line, "%-34s median %.3f ms  %.0f TOP/s  (%.3f of 5000)  shader clock %.0f MHz, %.1f cycles per MFMA and SIMD\n",
                 name, ms[2], ops / (ms[2] * 1e-3) / 1e12, ops / (ms[2] * 1e-3) / 5e15, mhz,
                 (double)clk[1] / ((double)iters * nacc * (threads / 256)));
        rep += line;
        return QAMD_OK;
    };
    QAMD_TRY(run("4 waves/CU, 8 accumulators", 256, 8, tune_mfma_peak<8>));
    QAMD_TRY(run("8 waves/CU, 8 accumulators", 512, 8, tune_mfma_peak<8>));
    QAMD_TRY(run("8 waves/CU, 4 accumulators", 512, 4, tune_mfma_peak<4>));
    QAMD_TRY(run("4 waves/CU, 16 accumulators", 256, 16, tune_mfma_peak<16>));
    QAMD_TRY(run("16x16x64: 8 waves/CU, 8 accumulators", 512, 8, tune_mfma_peak16<8>, 16.0 * 16.0 * 64.0 * 2.0));
    QAMD_TRY(run("16x16x64: 8 waves/CU, 16 accumulators", 512, 16, tune_mfma_peak16<16>, 16.0 * 16.0 * 64.0 * 2.0));
    QAMD_TRY(run("16x16x64: 4 waves/CU, 16 accumulators", 256, 16, tune_mfma_peak16<16>, 16.0 * 16.0 * 64.0 * 2.0));
    snprintf(report, cap, "%s", rep.c_str());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return QAMD_OK;
}

// ---------------------------------------------------------------- binary rows on the matrix cores (probe)
// Feasibility probe for a many-queries-at-once binary path: a wave streams 64 rows of 1024 bits (128 B), expands
// bits to 0/1 bytes in registers (nibble * 0x00204081 & 0x01010101) and feeds int8 MFMAs against a query
// tile of 0/1 bytes resident in LDS; acc = popcount(q AND v).  No epilogue: this measures the K loop only.
namespace {
typedef int bv4 __attribute__((ext_vector_type(4)));
typedef int bv16 __attribute__((ext_vector_type(16)));
template <int MI>
__global__ __launch_bounds__(512) void tune_bits_gemm(const uint8_t *__restrict__ rows, uint32_t n_rows, int *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t bits_lds[];
    constexpr int PA = 1024 + 16;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    for (uint32_t i = t; i < (uint32_t)(32 * MI) * PA / 4; i += 512)
        reinterpret_cast<uint32_t *>(bits_lds)[i] = (i * 2654435761u >> 7) & 0x01010101u;
    __syncthreads();
    const uint32_t n_chunks = n_rows / 64, stride = gridDim.x * 8;
    const uint8_t *a_base = bits_lds + r * PA + 64 * h;
    bv16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][jj][e] = 0;
    uint2 cur[2][8], nxt[2][8];
    auto load = [&](uint2(&w)[2][8], uint32_t chunk) {
        const uint32_t c = chunk < n_chunks ? chunk : n_chunks - 1;
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const uint8_t *p = rows + ((uint64_t)c * 64 + jj * 32 + r) * 128 + 8 * h;
#pragma unroll
            for (int kb = 0; kb < 8; kb++) w[jj][kb] = *reinterpret_cast<const uint2 *>(p + 16 * kb);
        }
    };
    auto expand = [&](uint32_t bits16) {  // 16 bits -> 16 bytes of 0/1
        bv4 v;
        v.x = (int)(((bits16 & 0xFu) * 0x00204081u) & 0x01010101u);
        v.y = (int)((((bits16 >> 4) & 0xFu) * 0x00204081u) & 0x01010101u);
        v.z = (int)((((bits16 >> 8) & 0xFu) * 0x00204081u) & 0x01010101u);
        v.w = (int)((((bits16 >> 12) & 0xFu) * 0x00204081u) & 0x01010101u);
        return v;
    };
    uint32_t chunk = blockIdx.x * 8 + wave;
    load(cur, chunk);
    for (; chunk < n_chunks; chunk += stride) {
        load(nxt, chunk + stride);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) {
#pragma unroll
            for (int x = 0; x < 4; x++) {
                bv4 bf[2];
#pragma unroll
                for (int jj = 0; jj < 2; jj++) {
                    const uint32_t word = (x < 2) ? cur[jj][kb].x : cur[jj][kb].y;
                    bf[jj] = expand((word >> (16 * (x & 1))) & 0xFFFFu);
                }
#pragma unroll
                for (int i = 0; i < MI; i++) {
                    const bv4 a = *reinterpret_cast<const bv4 *>(a_base + i * 32 * PA + kb * 128 + 16 * x);
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf[jj], acc[i][jj], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int kb = 0; kb < 8; kb++) cur[jj][kb] = nxt[jj][kb];
    }
    int tsum = 0;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int e = 0; e < 16; e++) tsum ^= acc[i][jj][e];
    if (tsum == 0x7fffffff) sink[0] = tsum;
}
}  // namespace

extern "C" __attribute__((visibility("default"))) qamd_status qamd_dev_bits_gemm_probe(const void *rows, uint32_t n_rows, int *sink,
                                                                                     char *report, size_t cap) {
    QAMD_ON_DEVICE(current_device());
    const int cu = device_info().cu_count;
    hipEvent_t e0, e1;
    QAMD_HIP(hipEventCreate(&e0));
    QAMD_HIP(hipEventCreate(&e1));
    std::string rep;
    auto run = [&](const char *name, int mi, auto kernel) -> qamd_status {
        const size_t lds = (size_t)32 * mi * (1024 + 16);
        QAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        std::vector<float> ms;
        for (int rr = 0; rr < 6; rr++) {
            QAMD_HIP(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(kernel, dim3(cu), dim3(512), lds, nullptr, static_cast<const uint8_t *>(rows), n_rows, sink);
            QAMD_HIP(hipEventRecord(e1, nullptr));
            QAMD_HIP(hipEventSynchronize(e1));
            float t = 0;
            QAMD_HIP(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        char line[256];
        snprintf(line, sizeof line, "%-28s median %.3f ms  %.0f G (query,row) pairs/s  %.0f int8 TOP/s\n", name, ms[3],
                 32.0 * mi * n_rows / (ms[3] * 1e-3) / 1e9, 32.0 * mi * n_rows * 2048.0 / (ms[3] * 1e-3) / 1e12);
        rep += line;
        return QAMD_OK;
    };
    QAMD_TRY(run("32 queries per tile", 1, tune_bits_gemm<1>));
    QAMD_TRY(run("64 queries per tile", 2, tune_bits_gemm<2>));
    QAMD_TRY(run("128 queries per tile", 4, tune_bits_gemm<4>));
    snprintf(report, cap, "%s", rep.c_str());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return QAMD_OK;
}

// ---------------------------------------------------------------- binary scan sweep (dim 1024)
namespace {
template <int UNROLL, int BLOCK, int STORE_MODE>
__global__ __launch_bounds__(BLOCK) void tune_bin(const uint4 *__restrict__ rows, const uint4 *__restrict__ qbits,
                                                 float dim_f, uint32_t n_rows, float *__restrict__ out) {
    constexpr int G = 8, RW = 8, TILE = RW * UNROLL;
    const int lane = threadIdx.x & 63, sub = lane % G, rslot = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint64_t base = wave * TILE;
    if (base >= n_rows) return;
    const uint4 q = qbits[sub];
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + (base + u * RW + rslot) * 8 + sub));
        v[u] = make_uint4(t.x, t.y, t.z, t.w);
    }
    float mine = 0.0f;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        uint32_t acc = __popc(v[u].x ^ q.x) + __popc(v[u].y ^ q.y) + __popc(v[u].z ^ q.z) + __popc(v[u].w ^ q.w);
#pragma unroll
        for (int m = 1; m < G; m <<= 1) acc += __shfl_xor(acc, m, 64);
        const float x = (float)acc;
        const float s = (dim_f - x) - x;
        if (STORE_MODE == 0) {
            const uint64_t row = base + u * RW + rslot;
            if (sub == 0 && row < n_rows) out[row] = s;
        } else if (STORE_MODE == 1) {
            if (sub == (u & 7)) mine = s;
            if ((u & 7) == 7) {  // 64 consecutive rows in one wave-store
                const uint64_t row = base + (u - 7 + sub) * RW + rslot;
                if (row < n_rows) out[row] = mine;
            }
        } else {
            if (sub == (u & 7)) mine = s;
            if ((u & 7) == 7) {
                const uint64_t row = base + (u - 7 + sub) * RW + rslot;
                if (row < n_rows) __builtin_nontemporal_store(mine, out + row);
            }
        }
    }
}
}  // namespace

extern "C" __attribute__((visibility("default"))) qamd_status qamd_dev_bin_sweep(const void *rows, const void *qbits,
                                                                                float dim_f, uint32_t n_rows,
                                                                                float *out_dev, int rounds,
                                                                                char *report, size_t cap) {
    QAMD_ON_DEVICE(current_device());
    const uint4 *r = static_cast<const uint4 *>(rows);
    const uint4 *q = static_cast<const uint4 *>(qbits);
    std::vector<Variant> vs;
#define ADDB(UN, BL, SM)                                                                                   \
    do {                                                                                                   \
        uint64_t waves = ((uint64_t)n_rows + 8 * UN - 1) / (8 * UN);                                       \
        unsigned grid = (unsigned)((waves + BL / 64 - 1) / (BL / 64));                                     \
        char nm[128];                                                                                      \
        snprintf(nm, sizeof nm, "bin unroll%d block%d store%d grid%u", UN, BL, SM, grid);                  \
        vs.push_back({nm, [=](hipStream_t s) {                                                             \
                          hipLaunchKernelGGL((tune_bin<UN, BL, SM>), dim3(grid), dim3(BL), 0, s, r, q, dim_f, n_rows, out_dev); \
                      }, {}});                                                                             \
    } while (0)
    ADDB(8, 512, 0);
    ADDB(8, 512, 1);
    ADDB(8, 512, 2);
    ADDB(8, 256, 1);
    ADDB(8, 1024, 1);
    ADDB(16, 512, 0);
    ADDB(16, 512, 1);
    ADDB(16, 512, 2);
    ADDB(16, 256, 1);
    ADDB(16, 1024, 1);
    ADDB(32, 512, 1);
    ADDB(32, 256, 1);
#undef ADDB
    hipEvent_t e0, e1;
    QAMD_HIP(hipEventCreate(&e0));
    QAMD_HIP(hipEventCreate(&e1));
    for (int rd = 0; rd < rounds; rd++)
        for (auto &v : vs) {
            v.launch(nullptr);
            QAMD_HIP(hipEventRecord(e0, nullptr));
            for (int i = 0; i < 5; i++) v.launch(nullptr);
            QAMD_HIP(hipEventRecord(e1, nullptr));
            QAMD_HIP(hipEventSynchronize(e1));
            float ms = 0;
            QAMD_HIP(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / 5);
        }
    std::string rep;
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const float med = v.ms[v.ms.size() / 2];
        char line[256];
        snprintf(line, sizeof line, "%-44s median %.4f ms  min %.4f ms  %.0f GB/s read (+4 B/row written)\n",
                 v.name.c_str(), med, v.ms.front(), (double)n_rows * 128.0 / (med * 1e-3) / 1e9);
        rep += line;
    }
    snprintf(report, cap, "%s", rep.c_str());
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return QAMD_OK;
}
