// Standalone experiment (hipcc --offload-arch=gfx950 -O3 stream_variants.hip -o /tmp/sv && /tmp/sv): pure streaming reads of
// 7.68 GB with different tile sizes per wave, workgroup sizes, cache-policy bits and persistent / non-persistent grids -
// is there an access shape that reads faster than the 7.0 TB/s the scans' shape reaches?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int POLICY> __device__ __forceinline__ u32x4 ld(const u32x4 *p) {
    if (POLICY == 0) return *p;                                   // default
    if (POLICY == 1) return __builtin_nontemporal_load(p);        // nt
    u32x4 v;
    if (POLICY == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    if (POLICY >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return v;
}

// one tile of LOADS x 1 KiB per wave, non-persistent
template <int LOADS, int POLICY, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_tile(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *scratch) {
    const uint64_t wave = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint64_t base = wave * (64 * LOADS) + (threadIdx.x & 63);
    uint32_t acc = 0;
    if (base + (LOADS - 1) * 64 < n16) {
        u32x4 v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; j++) v[j] = __builtin_nontemporal_load(p + base + j * 64);
        if (POLICY == 0) {
#pragma unroll
            for (int j = 0; j < LOADS; j++) v[j] = p[base + j * 64];
        }
#pragma unroll
        for (int j = 0; j < LOADS; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x9E3779B9u) scratch[blockIdx.x & 16383] = acc;
}
// persistent: grid = CUs * wgs_per_cu, each wave walks tiles with a grid stride
template <int LOADS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_persist(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *scratch) {
    const uint64_t waves = (uint64_t)gridDim.x * (BLOCK / 64);
    uint32_t acc = 0;
    for (uint64_t wave = ((uint64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;; wave += waves) {
        const uint64_t base = wave * (64 * LOADS) + (threadIdx.x & 63);
        if (base + (LOADS - 1) * 64 >= n16) break;
        u32x4 v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; j++) v[j] = __builtin_nontemporal_load(p + base + j * 64);
#pragma unroll
        for (int j = 0; j < LOADS; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x9E3779B9u) scratch[blockIdx.x & 16383] = acc;
}
// policy variants through inline assembly: 8 loads per wave
template <int POLICY>
__global__ __launch_bounds__(512) void k_policy(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *scratch) {
    const uint64_t wave = ((uint64_t)blockIdx.x * 512 + threadIdx.x) >> 6;
    const uint64_t base = wave * (64 * 8) + (threadIdx.x & 63);
    uint32_t acc = 0;
    if (base + 7 * 64 < n16) {
        u32x4 v[8];
        const u32x4 *q = p + base;
        if (POLICY == 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(q + j * 64) : "memory");
        } else if (POLICY == 3) {
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[j]) : "v"(q + j * 64) : "memory");
        } else if (POLICY == 4) {
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v[j]) : "v"(q + j * 64) : "memory");
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v[j]) : "v"(q + j * 64) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x9E3779B9u) scratch[blockIdx.x & 16383] = acc;
}

int main() {
    const uint64_t bytes = 7680000000ull, n16 = bytes / 16;
    void *buf, *scratch;
    hipMalloc(&buf, bytes);
    hipMalloc(&scratch, 1 << 16);
    hipMemset(buf, 1, bytes);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        std::vector<float> ms;
        for (int r = 0; r < 12; r++) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            if (r >= 3) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-56s median %.4f ms  %.0f GB/s   min %.4f ms\n", name, ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9, ms[0]);
        fflush(stdout);
    };
    const u32x4 *p = (const u32x4 *)buf;
    uint32_t *sc = (uint32_t *)scratch;
#define TILE(L, B)                                                                                           \
    run("nt, tile " #L " KiB per wave, block " #B, [&] {                                                     \
        const unsigned grid = (unsigned)((n16 / (64 * L) + (B / 64) - 1) / (B / 64));                        \
        hipLaunchKernelGGL((k_tile<L, 1, B>), dim3(grid), dim3(B), 0, 0, p, n16, sc);                        \
    })
    TILE(4, 512); TILE(8, 512); TILE(12, 512); TILE(16, 512); TILE(24, 512); TILE(32, 512);
    TILE(16, 256); TILE(16, 1024); TILE(8, 256); TILE(8, 1024); TILE(32, 256);
#define PERS(L, B, W)                                                                                        \
    run("nt, persistent, tile " #L " KiB, block " #B ", " #W " workgroups per CU", [&] {                      \
        hipLaunchKernelGGL((k_persist<L, B>), dim3(cus * W), dim3(B), 0, 0, p, n16, sc);                     \
    })
    PERS(16, 512, 1); PERS(16, 512, 2); PERS(16, 512, 4); PERS(8, 256, 8); PERS(16, 1024, 2); PERS(32, 512, 2);
#define POL(P, NAME)                                                                                         \
    run(NAME ", tile 8 KiB, block 512", [&] {                                                                \
        const unsigned grid = (unsigned)((n16 / (64 * 8) + 7) / 8);                                          \
        hipLaunchKernelGGL((k_policy<P>), dim3(grid), dim3(512), 0, 0, p, n16, sc);                          \
    })
    POL(2, "sc1"); POL(3, "sc0 sc1"); POL(4, "sc1 nt"); POL(5, "sc0 nt");
    return 0;
}
