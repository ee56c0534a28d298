// Stand-alone copy of pq_scan_skew_kernel (csrc/pq.hip) at M = 96, whole rows, plain score output, with single
// ingredients switched off by a template mask - to see which of them the kernel's time is made of - and the shader clock
// it runs at (s_memtime against s_memrealtime).  Not built into the library; results are NOT scores when anything is off.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pq_skew_probe tools/experiments/pq_skew_probe.hip && /tmp/pq_skew_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kCentroids = 256;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
    u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint32_t code_address(uint32_t code, uint32_t pitch, uint32_t base) { return __umul24(code, pitch) + base; }

enum : int { NO_GATHER = 1, NO_CODE = 2, NO_LOAD = 4, NO_REFILL = 8, NO_STORE = 16, NO_SELECT = 32, KEEP_LOADS = 64 };

template <int NV, int WAVES, int OFF, int D, int RUN, int ST>
__global__ __launch_bounds__(64 * (WAVES + (ST == 7 ? 1 : 0))) void skew(const uint4 *__restrict__ rows4, const float *__restrict__ lut_t_g, uint32_t n_rows,
                                                   float *__restrict__ out, unsigned long long *stamps) {
    constexpr int M = 16 * NV, S = 4 * NV, SR = S;
    constexpr int kWaves = WAVES, kThreads = 64 * (kWaves + (ST == 7 ? 1 : 0));
    constexpr uint32_t kMail0 = kWaves * 32u * M + (uint32_t)M * kCentroids * 4u, kFlags0 = kMail0 + kWaves * 512u;  // ST == 7: [wave][2][64] scores, produced[wave], consumed[wave]
    constexpr uint32_t kSlot = 16u * M, kStage0 = 0, kLut0 = kWaves * 2u * kSlot;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    {
        float4 *dst = reinterpret_cast<float4 *>(lds_raw + kLut0);
        const float4 *src = reinterpret_cast<const float4 *>(lut_t_g);
        for (uint32_t i = threadIdx.x; i < (uint32_t)M * (kCentroids / 4); i += kThreads) dst[i] = src[i];
        if (ST == 7 && threadIdx.x < 2 * kWaves) reinterpret_cast<uint32_t *>(lds_raw + kFlags0)[threadIdx.x] = 0;
        __syncthreads();
    }
    if (ST == 7 && (threadIdx.x >> 6) == kWaves) {
        // the STORE wave: nothing but score stores in its vmcnt queue.  Compute wave w hands over a run's 64 scores through
        // its two-slot mailbox; produced[w] / consumed[w] count runs.
        const uint32_t lane = threadIdx.x & 63, k = lane & 3, q = lane >> 2;
        const uint32_t n_waves = gridDim.x * kWaves, n_blocks = (n_rows + 15) / 16;
        const uint32_t runs = n_blocks / (n_waves * 4);  // per compute wave (the probe drops the tail)
        volatile uint32_t *produced = reinterpret_cast<volatile uint32_t *>(lds_raw + kFlags0), *consumed = produced + kWaves;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, n_rows * 4u, 0x00020000);
        uint32_t left = runs * kWaves;
        while (left) {
            for (uint32_t w = 0; w < (uint32_t)kWaves; w++) {
                const uint32_t c = consumed[w];
                if (c >= runs || produced[w] <= c) continue;
                const uint32_t v = reinterpret_cast<volatile uint32_t *>(lds_raw + kMail0 + w * 512u + (c & 1u) * 256u)[lane];
                const uint32_t gw = blockIdx.x * kWaves + w;
                const uint32_t row = ((gw * 4u + k) + c * n_waves * 4u) * 16u + q;
                __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, row < n_rows ? row * 4u : 0xFFFFFFFFu, 0, 2);
                consumed[w] = c + 1;
                left--;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = lane & 3, q = lane >> 2, r = 8u - (q & 7u);
    const uint32_t gw = blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * kWaves;
    const uint32_t n_blocks = (n_rows + 15) / 16;
    if (gw >= n_blocks) return;
    const uint32_t J = RUN == 1 ? (n_blocks - gw + n_waves - 1) / n_waves : (n_blocks / (n_waves * RUN)) * RUN;  // (RUN > 1: the tail is dropped - timing only)
    const uint32_t stage = kStage0 + (threadIdx.x >> 6) * 2u * kSlot;
    const uint32_t off_cur = kLut0 + 4u * k - 16u * r, off_new = off_cur + 4u * M;
    const uint32_t rd8 = stage + q * M + k + 4u * (8u - r);
    uint32_t rd_even[8], rd_odd[8], lut_base[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const bool lag = (int)r > u;
        const uint32_t at = stage + q * M + k + (lag ? 4u * (u + S - r) : 4u * (u - r));
        rd_even[u] = at + (lag ? kSlot : 0u);
        rd_odd[u] = at + (lag ? 0u : kSlot);
        lut_base[u] = lag ? off_new : off_cur;
    }
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, n_rows * 4u, 0x00020000);
    constexpr int kWide = 16 * M / 1024;
    constexpr bool kHalf = (16 * M) % 1024 != 0;
    struct Held {
        uint4 wide[kWide > 0 ? kWide : 1];
        uint2 half;
    };
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(gw), n_waves_u = __builtin_amdgcn_readfirstlane(n_waves);
    const uint32_t J_u = __builtin_amdgcn_readfirstlane(J);
    const uint8_t *rows_b = reinterpret_cast<const uint8_t *>(rows4);
    auto request = [&](Held &h, uint32_t j) {
        if (OFF & NO_LOAD) {
            asm volatile("" : "+v"(h.wide[0].x), "+v"(h.half.x));
            return;
        }
        const uint32_t jc = j < J_u ? j : J_u - 1;
        const uint32_t blk = RUN == 1 ? wave_u + jc * n_waves_u : wave_u * RUN + (jc % RUN) + (jc / RUN) * n_waves_u * RUN;
        const uint8_t *p = rows_b + (size_t)blk * 16u * M;
#pragma unroll
        for (int i = 0; i < kWide; i++) h.wide[i] = ld_nt(reinterpret_cast<const uint4 *>(p + 1024 * i) + lane);
        if (kHalf) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p + 1024u * kWide + 8u * lane));
            h.half = make_uint2(t.x, t.y);
        }
    };
    auto refill = [&](const Held &h, uint32_t slot) {
        if ((OFF & NO_REFILL) && (OFF & KEEP_LOADS)) {  // no ring write, but the loads are waited for where the write would be
            asm volatile("" ::"v"(h.wide[0].x), "v"(h.wide[0].w), "v"(h.half.x), "v"(h.half.y));
            return;
        }
        if (OFF & NO_REFILL) return;
        uint8_t *d = lds_raw + stage + slot * kSlot;
#pragma unroll
        for (int i = 0; i < kWide; i++) *reinterpret_cast<uint4 *>(d + 1024u * i + 16u * lane) = h.wide[i];
        if (kHalf) *reinterpret_cast<uint2 *>(d + 1024u * kWide + 8u * lane) = h.half;
    };
    Held buf[D];
#pragma unroll
    for (int j = 0; j < D; j++) {
        buf[j].wide[0] = make_uint4(lane, lane * 3, lane * 5, lane * 7);
        buf[j].half = make_uint2(lane * 11, lane * 13);
        if (kWide > 1) buf[j].wide[kWide > 1 ? 1 : 0] = make_uint4(lane, lane * 3, lane * 5, lane * 7);
        request(buf[j], j);
    }
    refill(buf[0], 0);
    request(buf[0], D);
    constexpr int GR = S / 8, NG = D * GR;
    auto ring_addr = [&](int jj, int u) {
        return u < 8 ? ((jj & 1) ? rd_odd[u] : rd_even[u]) : rd8 + (uint32_t)(4 * (u - 8)) + ((jj & 1) ? kSlot : 0u);
    };
    uint32_t codes[2][8];
    float vals[2][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        codes[0][e] = lds_raw[ring_addr(0, e)];
        codes[1][e] = (lane * 7 + e) & 255;
        vals[1][e] = 0.0f;
        vals[0][e] = 0.0f;
    }
    float acc = 0.0f, done = 0.0f, keep[RUN >= 4 ? RUN / 4 : 1] = {};
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    for (uint32_t j0 = 0; j0 < J + 2; j0 += D) {
#pragma unroll
        for (int G = 0; G < NG; G++) {
            const int jj = G / GR, g = G % GR;
            if (g == 0) {
                refill(buf[(jj + 1) % D], (jj + 1) & 1);
                request(buf[(jj + 1) % D], j0 + jj + 1 + D);
            }
            if (!(OFF & NO_CODE)) {  // A(G + 1)
                const int Gn = (G + 1) % NG, jjn = Gn / GR, gn = Gn % GR;
#pragma unroll
                for (int e = 0; e < 8; e++) codes[(G + 1) & 1][e] = lds_raw[ring_addr(jjn, 8 * gn + e)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; e++) {  // B(G)
                const int u = 8 * g + e;
                const uint32_t base = u < 8 ? lut_base[u] : off_cur;
                const uint32_t ad = code_address(codes[G & 1][e], 4u * M, base) + 16 * u;
                if (OFF & NO_GATHER) vals[G & 1][e] = __uint_as_float(ad);
                else vals[G & 1][e] = *reinterpret_cast<const float *>(lds_raw + ad);
            }
            __builtin_amdgcn_sched_barrier(0);
            {  // C(G - 1)
                const int Gp = (G + NG - 1) % NG, jjp = Gp / GR, gp = Gp % GR;
                const uint32_t jp = G == 0 ? j0 - 1u : j0 + (uint32_t)jjp;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    acc += vals[(G + 1) & 1][e];
                    const int u = 8 * gp + e, w = u % SR;
                    if (w < 8 && !(OFF & NO_SELECT)) {
                        const bool fin = (int)r == w + 1;
                        done = fin ? acc : done;
                        acc = fin ? 0.0f : acc;
                    }
                    if (w == 7) {
                        if (OFF & NO_SELECT) {
                            done = acc;
                            acc = 0.0f;
                        }
                        const float a = done + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(done), 0x4E, 0xF, 0xF, false));
                        const float sc = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF, false));
                        const uint32_t jb = jp - 1u;
                        if (ST) {  // RUN = 4 NK, D % RUN == 0: lane k keeps the scores of the run's blocks k, k + 4, ...; NK 256-byte stores per run
                            constexpr int NK = RUN / 4;
                            const int c = G == 0 ? RUN - 2 : ((jjp + RUN - 1) % RUN);  // = jb % RUN (a constant once the loop is unrolled)
                            keep[c >> 2] = (int)k == (c & 3) ? sc : keep[c >> 2];
                            if (ST == 7 && c == RUN - 1) {
                                const uint32_t wv = threadIdx.x >> 6, rj = jb >> 2;  // this wave's run number (jb = -1: the first trip, nothing yet)
                                if (jb < J) {
                                    volatile uint32_t *produced = reinterpret_cast<volatile uint32_t *>(lds_raw + kFlags0), *consumed = produced + kWaves;
                                    while (consumed[wv] + 2u <= rj) __builtin_amdgcn_s_sleep(1);  // the slot still holds run rj - 2
                                    reinterpret_cast<volatile uint32_t *>(lds_raw + kMail0 + wv * 512u + (rj & 1u) * 256u)[lane] = __float_as_uint(keep[0]);
                                    produced[wv] = rj + 1u;
                                }
                            } else if (c == RUN - 1) {
#pragma unroll
                                for (int i = 0; i < NK; i++) {
                                    const uint32_t jbk = jb - (uint32_t)(RUN - 1) + 4u * i + k;
                                    const uint32_t row = ((gw * RUN + 4u * i + k) + (jb / RUN) * n_waves * RUN) * 16u + q;
                                    const bool live = jbk < J && row < n_rows;
                                    constexpr int aux = ST == 2 ? 2 : ST == 3 ? 3 : ST == 4 ? 17 : ST == 5 ? 19 : ST == 6 ? 16 : 0;
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(keep[i]), out_rsrc, live ? row * 4u : 0xFFFFFFFFu, 0, aux);
                                }
                            }
                        } else {
                        const uint32_t row = (RUN == 1 ? gw + jb * n_waves : gw * RUN + (jb % RUN) + (jb / RUN) * n_waves * RUN) * 16u + q;
                        const bool live = jb < J && row < n_rows;
                        if (OFF & NO_STORE) {
                            if (sc == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), out_rsrc, live ? row * 4u : 0xFFFFFFFFu, 0, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), out_rsrc, live ? row * 4u : 0xFFFFFFFFu, 0, 0);
                        }
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        stamps[0] = __builtin_amdgcn_s_memtime() - t0;
        stamps[1] = __builtin_amdgcn_s_memrealtime() - r0;
        stamps[2] = J;
    }
}

template <int WAVES, int OFF, int D = 4, int RUN = 1, int ST = 0>
static double run(const uint4 *rows, const float *lut, uint32_t n, float *out, unsigned long long *stamps, const char *what) {
    auto k = skew<6, WAVES, OFF, D, RUN, ST>;
    const size_t lds = (size_t)WAVES * 32u * 96 + (size_t)96 * kCentroids * 4 + (ST == 7 ? WAVES * 512u + 256u : 0u);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(256), dim3(64 * (WAVES + (ST == 7 ? 1 : 0))), lds, 0, rows, lut, n, out, stamps);
    CK(hipEventRecord(e0));
    const int reps = 100;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(256), dim3(64 * (WAVES + (ST == 7 ? 1 : 0))), lds, 0, rows, lut, n, out, stamps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    unsigned long long st[3];
    CK(hipMemcpy(st, stamps, 24, hipMemcpyDeviceToHost));
    const double ghz = (double)st[0] / ((double)st[1] * 10.0);  // s_memrealtime: 100 MHz
    (void)ghz;
    (void)what;
    return ms;
}

int main() {
    const uint32_t n = 10000000;
    std::vector<uint8_t> h((size_t)n * 96 + 65536);
    uint64_t s = 88172645463325252ull;
    for (auto &b : h) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        b = (uint8_t)(s >> 24);
    }
    std::vector<float> lut((size_t)96 * 256);
    for (size_t i = 0; i < lut.size(); i++) lut[i] = (float)((i * 2654435761u) >> 16 & 1023) * 0.01f;
    uint4 *rows;
    float *lut_d, *out;
    unsigned long long *stamps;
    CK(hipMalloc(&rows, h.size()));
    CK(hipMalloc(&lut_d, lut.size() * 4));
    CK(hipMalloc(&out, (size_t)n * 4));
    CK(hipMalloc(&stamps, 64));
    CK(hipMemcpy(rows, h.data(), h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(lut_d, lut.data(), lut.size() * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 10; i++) run<16, 0>(rows, lut_d, n, out, stamps, "warm");  // clocks ramp with busy time: ~0.25 s first
    const int NC = 8, ROUNDS = 3;
    double t[NC][ROUNDS];
    const char *names[NC];
    for (int r = 0; r < ROUNDS; r++) {
        t[0][r] = run<16, NO_GATHER | NO_CODE | NO_SELECT, 4, 4, 2>(rows, lut_d, n, out, stamps, ""); names[0] = "stream: loads, ring writes, adds, stores             <16, NO_GATHER | NO_CODE | NO_SELECT, 4, 4, 2>";
        t[1][r] = run<16, NO_GATHER | NO_CODE | NO_SELECT | NO_REFILL | KEEP_LOADS, 4, 4, 2>(rows, lut_d, n, out, stamps, ""); names[1] = "stream without the ring writes (loads waited for)    <16, NO_GATHER | NO_CODE | NO_SELECT | NO_REFILL | KEEP_LOADS, 4, 4, 2>";
        t[2][r] = run<16, NO_GATHER | NO_CODE | NO_SELECT | NO_STORE, 4, 4, 0>(rows, lut_d, n, out, stamps, ""); names[2] = "stream without the stores                            <16, NO_GATHER | NO_CODE | NO_SELECT | NO_STORE, 4, 4, 0>";
        t[3][r] = run<16, NO_GATHER | NO_CODE | NO_SELECT | NO_STORE | NO_REFILL | KEEP_LOADS, 4, 4, 0>(rows, lut_d, n, out, stamps, ""); names[3] = "stream without ring writes and stores                <16, NO_GATHER | NO_CODE | NO_SELECT | NO_STORE | NO_REFILL | KEEP_LOADS, 4, 4, 0>";
        t[4][r] = run<16, NO_GATHER | NO_CODE | NO_SELECT, 4, 1, 0>(rows, lut_d, n, out, stamps, ""); names[4] = "stream, single blocks, 64 B stores                   <16, NO_GATHER | NO_CODE | NO_SELECT, 4, 1, 0>";
        t[5][r] = run<16, NO_GATHER | NO_CODE | NO_SELECT | NO_REFILL | KEEP_LOADS, 4, 1, 0>(rows, lut_d, n, out, stamps, ""); names[5] = "the same without the ring writes                     <16, NO_GATHER | NO_CODE | NO_SELECT | NO_REFILL | KEEP_LOADS, 4, 1, 0>";
        t[6][r] = run<16, 0, 4, 4, 2>(rows, lut_d, n, out, stamps, ""); names[6] = "the kernel                                           <16, 0, 4, 4, 2>";
        t[7][r] = run<16, NO_REFILL | KEEP_LOADS, 4, 4, 2>(rows, lut_d, n, out, stamps, ""); names[7] = "the kernel without the ring writes (stale codes)     <16, NO_REFILL | KEEP_LOADS, 4, 4, 2>";
    }
    for (int c = 0; c < NC; c++) {
        double mn = 1e9, mx = 0, sum = 0;
        for (int r = 0; r < ROUNDS; r++) mn = t[c][r] < mn ? t[c][r] : mn, mx = t[c][r] > mx ? t[c][r] : mx, sum += t[c][r];
        printf("%-120s min %.4f  mean %.4f  max %.4f ms\n", names[c], mn, sum / ROUNDS, mx);
    }
    return 0;
}
