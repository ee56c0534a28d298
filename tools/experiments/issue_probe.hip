// Stand-alone probe (not built into the library): what a CU of gfx950 pays per trip of a loop shaped like
// pq_scan_skew_kernel's - per eight gathers: 8 ds_read_u8 (code bytes out of a ring, static addresses), 8 v_mad_u32_u24
// (table address), 8 ds_read_b32 (conflict-free gathers from a [code][chunk] table), 8 v_add_f32 and NX more vector-ALU
// instructions - software-pipelined like the real one: nothing waits for a read issued in the same trip.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_probe tools/experiments/issue_probe.hip && /tmp/issue_probe
// One workgroup per CU, W waves each.  Variants switch single ingredients off to see which of them add and which hide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t kRing = 48 * 1024, kLut = 96 * 1024, kPitch = 384;  // table rows of 96 floats

// CODE: 0 no code reads (codes stay what they are), 1 ds_read_u8, 2 ds_read_b32 at the same place (& 255 folded in the mad)
// GATH: gathers on/off     MAD / ADD: the two vector-ALU instructions per gather on/off     NX: extra vector-ALU per trip
template <int CODE, int GATH, int MAD, int ADD, int NX>
__global__ __launch_bounds__(1024) void probe(uint32_t iters, uint32_t *sink, unsigned long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (kRing + kLut) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t *>(lds)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    uint32_t ring[8], base[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        // ring: this wave's 3 KiB, dword stride between quads odd (25) as in the kernel: conflict-free byte reads
        ring[e] = (wave * 3072u + (lane >> 2) * 100u + (lane & 3u) + 4u * e) % kRing;
        base[e] = kRing + 4u * ((lane + 4u * e) & 31u) + 128u * (e & 1) + 256u * (e >> 2 & 0);  // bank = column: distinct per half-wave
    }
    uint32_t code[2][8], val[2][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        code[0][e] = code[1][e] = (lane * 7u + e * 13u) & 255u;
        val[0][e] = val[1][e] = 0;
    }
    float acc = 0.f, done = 0.f;
    uint32_t x = lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it += 2) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            // A: next trip's code bytes
            if (CODE == 1) {
#pragma unroll
                for (int e = 0; e < 8; e++) asm volatile("ds_read_u8 %0, %1" : "=&v"(code[p ^ 1][e]) : "v"(ring[e]));
            } else if (CODE == 2) {
#pragma unroll
                for (int e = 0; e < 8; e++) asm volatile("ds_read_b32 %0, %1" : "=&v"(code[p ^ 1][e]) : "v"(ring[e] & ~3u));
            }
            __builtin_amdgcn_sched_barrier(0);
            // B: this trip's addresses and gathers (codes asked for one trip ago: at most the 8 reads above are newer)
            if (CODE && GATH) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
            else if (CODE) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            uint32_t ad[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const uint32_t c = CODE == 2 ? (code[p][e] & 255u) : code[p][e];
                ad[e] = MAD ? __umul24(c, kPitch) + base[e] : base[e] + (code[p][e] & 0u);
            }
            if (GATH) {
#pragma unroll
                for (int e = 0; e < 8; e++) asm volatile("ds_read_b32 %0, %1" : "=&v"(val[p][e]) : "v"(ad[e]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) val[p][e] = ad[e];
            }
            __builtin_amdgcn_sched_barrier(0);
            // C: the previous trip's table entries (16 newer reads: lgkmcnt saturates at 15)
            if (GATH && CODE) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
            else if (GATH) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (ADD) acc += __uint_as_float(val[p ^ 1][e]);
                else x ^= val[p ^ 1][e] & (e == 0 ? ~0u : 0u);
                if (2 * e < NX) {  // selects like the kernel's row-end bookkeeping, between the adds as there
                    const bool fin = (lane >> 2 & 7u) == (uint32_t)e;
                    done = fin ? acc : done;
                    acc = fin ? 0.0f : acc;
                }
            }
            asm volatile("" : "+v"(acc), "+v"(done));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = x ^ __float_as_uint(acc) ^ __float_as_uint(done);
#pragma unroll
    for (int e = 0; e < 8; e++) s ^= val[0][e] ^ val[1][e] ^ code[0][e] ^ code[1][e];
    if (s == 0x12345678u) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}


// SPLIT: the waves of a SIMD take ROLES - (wave >> 2) even: only LDS reads (2 x 16 per trip), odd: only vector-ALU (2 x NVV
// per trip) - the same totals per SIMD as 16 reads + NVV ALU per wave.  max(LDS, ALU) = the hardware overlaps the two
// kinds across waves; their sum = it does not.
template <int NVV>
__global__ __launch_bounds__(1024) void probe_split(uint32_t iters, uint32_t *sink, unsigned long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (kRing + kLut) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t *>(lds)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    uint32_t base[8];
#pragma unroll
    for (int e = 0; e < 8; e++) base[e] = kRing + 4u * ((lane + 4u * e) & 31u) + 128u * (e & 1) + 384u * ((lane * 7u + e) & 255u);
    uint32_t val[4][8];
    float acc = 0.f, done = 0.f;
    uint32_t x = lane, y = lane * 3u;
    if (((wave >> 2) & 1u) == 0) {
        for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
            for (int p = 0; p < 4; p++) {
#pragma unroll
                for (int e = 0; e < 8; e++) asm volatile("ds_read_b32 %0, %1" : "=&v"(val[p][e]) : "v"(base[e]));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 8; e++) x ^= val[0][e] ^ val[1][e] ^ val[2][e] ^ val[3][e];
    } else {
        for (uint32_t it = 0; it < iters; it++) {
#pragma unroll
            for (int e = 0; e < 2 * NVV; e++) {
                if (e & 1) acc += __uint_as_float(x);
                else x = __umul24(x, 0x101u) + y;
            }
            asm volatile("" : "+v"(acc), "+v"(x));
        }
    }
    uint32_t s = x ^ __float_as_uint(acc) ^ __float_as_uint(done);
    if (s == 0x12345678u) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = 0;
}

// FINE: the same 16 reads + 16 + NX ALU per trip and wave, but issued alternately (read, ALU, read, ALU, ...), each from
// a different pipeline stage, instead of in phases of eight.
template <int NX>
__global__ __launch_bounds__(1024) void probe_fine(uint32_t iters, uint32_t *sink, unsigned long long *cycles) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < (kRing + kLut) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t *>(lds)[i] = (i * 2654435761u) >> 7;
    __syncthreads();
    uint32_t ring[8], base[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        ring[e] = (wave * 3072u + (lane >> 2) * 100u + (lane & 3u) + 4u * e) % kRing;
        base[e] = kRing + 4u * ((lane + 4u * e) & 31u) + 128u * (e & 1);
    }
    uint32_t code[2][8], val[2][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        code[0][e] = code[1][e] = (lane * 7u + e * 13u) & 255u;
        val[0][e] = val[1][e] = 0;
    }
    float acc = 0.f, done = 0.f;
    for (uint32_t it = 0; it < iters; it += 2) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // everything asked for in the previous trip is here
#pragma unroll
            for (int e = 0; e < 8; e++) {
                asm volatile("ds_read_u8 %0, %1" : "=&v"(code[p ^ 1][e]) : "v"(ring[e]));
                const uint32_t ad = __umul24(code[p][e], kPitch) + base[e];
                asm volatile("" ::"v"(ad));
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("ds_read_b32 %0, %1" : "=&v"(val[p][e]) : "v"(ad));
                acc += __uint_as_float(val[p ^ 1][e]);
                if (2 * e < NX) {
                    const bool fin = (lane >> 2 & 7u) == (uint32_t)e;
                    done = fin ? acc : done;
                    acc = fin ? 0.0f : acc;
                }
                asm volatile("" : "+v"(acc), "+v"(done));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t s = __float_as_uint(acc) ^ __float_as_uint(done);
#pragma unroll
    for (int e = 0; e < 8; e++) s ^= val[0][e] ^ val[1][e] ^ code[0][e] ^ code[1][e];
    if (s == 0x12345678u) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = 0;
}

template <typename K>
static void run_k(K k, int waves, const char *what) {
    static uint32_t *sink = nullptr;
    static unsigned long long *cyc = nullptr;
    if (!sink) {
        CK(hipMalloc(&sink, 64));
        CK(hipMalloc(&cyc, 64));
    }
    const uint32_t iters = 6000;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 160 * 1024, 0, iters, sink, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 160 * 1024, 0, iters, sink, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s W=%2d : %7.1f ns per trip of all waves\n", what, waves, ms * 1e6 / iters);
}

template <int CODE, int GATH, int MAD, int ADD, int NX>
static void run(int waves, const char *what) {
    static uint32_t *sink = nullptr;
    static unsigned long long *cyc = nullptr;
    if (!sink) {
        CK(hipMalloc(&sink, 64));
        CK(hipMalloc(&cyc, 64));
    }
    const uint32_t iters = 6000;
    auto k = probe<CODE, GATH, MAD, ADD, NX>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 160 * 1024, 0, iters, sink, cyc);  // 160 KiB: one workgroup per CU
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 160 * 1024, 0, iters, sink, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c = 0;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double per_trip = (double)c / iters;
    const int nl = (CODE ? 8 : 0) + (GATH ? 8 : 0), nv = (MAD ? 8 : 0) + (ADD ? 8 : 0) + NX;
    // wall time of a trip of ALL the CU's waves = per_trip (each wave runs its trips concurrently with the others)
    (void)per_trip;  // the oldest wave wins every arbitration: wave 0's own cycles say nothing; the launch's wall time does
    const double ns = ms * 1e6 / iters;
    printf("%-44s W=%2d LDS %2d VALU %2d : %7.1f ns per trip of all waves   LDS array %5.1f%%   VALU (2 clk) %5.1f%%   (at 2.4 GHz)\n", what,
           waves, nl, nv, ns, 100.0 * 2.0 * nl * waves / (ns * 2.4), 100.0 * 2.0 * nv * (waves / 4.0) / (ns * 2.4));
}

int main() {
    for (int waves : {16, 12, 8, 4}) {
        run<1, 1, 1, 1, 6>(waves, "the loop: u8 + mad + gather + add + 6");
        run<1, 1, 1, 1, 0>(waves, "  without the 6 selects");
        run<1, 1, 0, 0, 0>(waves, "  LDS only (u8 + gather)");
        run<2, 1, 0, 0, 0>(waves, "  LDS only (b32 + gather)");
        run<0, 1, 0, 0, 0>(waves, "  gathers only");
        run<1, 0, 0, 0, 0>(waves, "  u8 reads only");
        run<2, 0, 0, 0, 0>(waves, "  b32 reads only");
        run<0, 0, 1, 1, 6>(waves, "  VALU only (22)");
        run<0, 0, 1, 1, 0>(waves, "  VALU only (16)");
        run<0, 1, 1, 1, 6>(waves, "  no code reads");
        run<1, 0, 1, 1, 6>(waves, "  no gathers");
        run<2, 1, 1, 1, 6>(waves, "  codes by ds_read_b32 (+ and)");
        if (waves >= 8) {
            run_k(probe_split<16>, waves, "roles: half the waves 32 reads, half 32 ALU (= 16 + 16 per wave)");
            run_k(probe_split<22>, waves, "roles: half the waves 32 reads, half 44 ALU (= 16 + 22 per wave)");
            run_k(probe_split<0>, waves, "roles: half the waves 32 reads, the others nothing");
        }
        run_k(probe_fine<6>, waves, "fine: read, ALU, read, ALU ... (16 + 22)");
        run_k(probe_fine<0>, waves, "fine: read, ALU, read, ALU ... (16 + 16)");
    }
    return 0;
}
