// Multi-query scoring of the scalar-u8 store on the matrix cores (gfx950 MFMA, int8).
//
// What it replaces: the reference scores queries one at a time — the caller loops over
// queries and, inside, over rows (demos/src/ann_benchmark.rs:245-260), keeping a 30-entry heap
// per query (demos/src/ann_benchmark_data.rs:151-167).  With Q queries in flight the same
// work is a dense u8 x u8 -> i32 contraction  S[q][row] = sum_d Qc[q][d] * V[row][d]
// (codes <= 127, so int8 is exact), i.e. GEMM-shaped: every byte of the store read from HBM is
// used Q times, and at Q ~ 1024 the bound moves from HBM to the MFMA pipe (SURVEY 8f rank 1,
// BASELINE config 4).  The f32 epilogue is the single-query one,
// ((multiplier*s)+q.offset)+vector_offset (encoded_vectors_u8.rs:347), so every score is
// bit-identical to qamd_u8_score_all for that query (integer sum rounded once).
//
// Both operands are K-contiguous in memory (queries [Q][pitch], store rows [N][D]) and the MFMA
// K index is only a summation index, so lane (r, h) simply takes bytes [32s + 16h, +16) of
// query/row r in k-step s: fragments are plain 16-byte pieces, no transposition anywhere.
//
// Six kernels share the arithmetic, the integer pre-filter and the epilogue; launch_gemm picks by what
// is re-read and from where (DESIGN.md 3.3b has the table and the measurements):
//   * u8_gemm_rs_kernel<MODE, LOW, MI, NT> -- row-streaming: up to 128 queries (and several 128-query
//     tiles up to ~700): the query tile resident in LDS, every wave streams its own rows HBM ->
//     registers (coalesced, nt) -> wave-private LDS transpose -> MFMA.  HBM-bound, no barriers.
//   * u8_gemm_qs16_kernel<MODE, LOW, JT, IT> / u8_gemm_qs_kernel<MODE, LOW, MJ> -- query-streaming: many queries on
//     rows of up to 1536 B: 128 (96) store rows resident in LDS, the batch streamed from L2 in MFMA fragment
//     order straight into operand registers.  Rows leave HBM once; the reuse needs no co-scheduling of
//     workgroups.  qs16 (round 3, the default) runs on v_mfma_i32_16x16x64_i8, which this part clocks a fifth
//     higher under load than the 32x32x32 instruction of the round-2 form (QAMD_QS16=0 selects that one).
//   * u8_gemm_qr16_kernel<MODE, LOW, NSTEPS> -- 129 .. 256 queries on rows of 256 / 384 / 512 / 768 / 1024 B (round 3): a wave's 32
//     queries in registers for all k-steps, the rows through a double-buffered 64-row LDS slab filled by LDS-DMA under
//     the MFMAs; no vector-memory wait in the K loop, one barrier per block.
//   * u8_gemm_pp_kernel<MODE, LOW, MI, MJ> -- ping-pong (round 1): both operands through an LDS-DMA
//     ring, two wave groups half a phase apart; now for what the two above do not take.
//   * u8_gemm_kernel<MODE, TQ, TR, WQ, WR, BK> -- the first version (128-byte K slabs through
//     registers -> ds_write -> LDS at a 144-byte pitch, one barrier per slab, float-compare filter,
//     per-query global atomics); now only for short rows no tile applies to, a zero / non-finite
//     multiplier, and the developer switch QAMD_GEMM_CFG (r / q / p force one of the three above).
//
// Top-k per query is fused as in topk.hip: a pivot per query from S sampled rows (scored by the
// same kernel on a gathered sub-store; S grows with the store, see qamd_u8_topk_batch), a filter
// pass appending candidates, one workgroup per query sorting its list; queries whose list
// over/underflows are redone through the exact single-query path.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "topk.hpp"
#include "topk_device.hpp"
#include "u8_internal.hpp"
#include "batch_common.hpp"

#pragma clang fp contract(off)

using namespace qamd;

namespace {

constexpr int TQ = 256;  // largest query tile: padding granularity of a query batch
// Developer timeline (tools/gemm_timeline.py): when set, lane 0 of every wave of the first 4096
// workgroups stores s_memtime at the phase boundaries, 16 slots per wave.
#ifdef QAMD_DEV  // libquantization_amd_dev.so only (make dev): the in-kernel timeline hook of tools/gemm_timeline.py
__device__ unsigned long long *g_gemm_stamps = nullptr;
#define QAMD_GEMM_STAMPS() g_gemm_stamps
#else            // product build: no hook, the stamp branches fold away
#define QAMD_GEMM_STAMPS() (static_cast<unsigned long long *>(nullptr))
#endif
#ifdef QAMD_GEMM_ABLATION  // developer builds only (make EXTRA=-DQAMD_GEMM_ABLATION): the shipped library has no such switch
__device__ unsigned int g_gemm_dbg = 0;  // TIMING EXPERIMENTS ONLY (results are wrong when set): bit0 skip A DMA, bit1 skip B DMA
#endif
constexpr uint32_t kStampBlocks = 4096;

// S[q][row] for the tile; MODE 0: write scores out[q * out_pitch + row]; MODE 1 / 2: filter for the
// largest / smallest scores.
// Tile TQ queries x TR rows per workgroup, WQ x WR waves, each wave (MI*32) x (MJ*32) outputs.
//   <128,128,2,2>: 4 waves, 72 KiB LDS  — small batches (padding to 128 queries)
//   <256,256,2,4>: 8 waves, 144 KiB LDS — the guide's 256^2 shape: a 128^2 tile needs 32 KiB of
//   operands per 512 MFMA cycles, more than one CU's share of L2 bandwidth (~55 B/clk), and
//   measured 13 % of the int8 MFMA peak; at 256^2 the slab is 64 KiB per 2048 MFMA cycles.
template <int MODE, int TQ_, int TR_, int WQ, int WR, int BK_>
__global__ __launch_bounds__(64 * WQ * WR) void u8_gemm_kernel(const uint8_t *__restrict__ codes,
                                                              const float *__restrict__ v_offsets,
                                                              const uint8_t *__restrict__ qcodes, uint32_t q_pitch,
                                                              const float *__restrict__ q_offsets, float multiplier,
                                                              uint32_t n_rows, uint32_t n_queries, uint32_t ad,
                                                              uint32_t q_tiles, float *__restrict__ out,
                                                              uint64_t out_pitch, BatchFilter filt) {
    constexpr int T = 64 * WQ * WR;
    constexpr int MI = TQ_ / WQ / 32, MJ = TR_ / WR / 32;
    constexpr int BK = BK_, PITCH = BK_ + 16;  // slab bytes per row; +16 B pad: conflict-free ds_read_b128
    constexpr int CH = BK / 16;                // 16-byte chunks per slab row
    constexpr int ROWS_PER_PASS = T / CH;      // CH lanes x 16 B cover one slab row
    constexpr int LA = TQ_ / ROWS_PER_PASS, LB = TR_ / ROWS_PER_PASS;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    // layout: buffer b: A at b*(TQ+TR)*PITCH, B right after A
    auto ldsA = [&](int buf) { return lds_raw + (size_t)buf * (TQ_ + TR_) * PITCH; };
    auto ldsB = [&](int buf) { return lds_raw + (size_t)buf * (TQ_ + TR_) * PITCH + (size_t)TQ_ * PITCH; };
    // XCD-aware tile map: blocks b, b+8, b+16, ... (one XCD under round-robin dispatch) walk the
    // query tiles of ONE row tile; speed only, any placement is correct.
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u, w = b >> 3;
    const uint32_t q_tile = w % q_tiles;
    const uint32_t r_tile = (w / q_tiles) * 8u + xcd;
    const uint64_t row0 = (uint64_t)r_tile * TR_;
    if (row0 >= n_rows) return;
    const uint32_t q0 = q_tile * TQ_;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wq = wave / WR, wr = wave % WR;
    const int r = lane & 31, h = lane >> 5;

    unsigned long long *stamps = QAMD_GEMM_STAMPS();
    if (stamps) stamps = (blockIdx.x < kStampBlocks && lane == 0) ? stamps + ((uint64_t)blockIdx.x * (T / 64) + wave) * 16 : nullptr;
    auto stamp = [&](int slot) {
        if (stamps) stamps[slot] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    const int s_chunk = t % CH, s_row = t / CH;
    const uint8_t *gA = qcodes + (uint64_t)(q0 + s_row) * q_pitch + s_chunk * 16;
    const uint8_t *gB = codes + (row0 + s_row) * ad + s_chunk * 16;
    const uint32_t n_slabs = (ad + BK - 1) / BK;

    uint4 ra[LA], rb[LB];
    auto load_slab = [&](uint32_t s) {
        const uint32_t k = s * BK + s_chunk * 16;
        const bool in = k < ad;  // ad is a multiple of 16: a chunk is entirely in or out
#pragma unroll
        for (int i = 0; i < LA; i++)
            ra[i] = in ? *reinterpret_cast<const uint4 *>(gA + (uint64_t)(ROWS_PER_PASS * i) * q_pitch + s * BK)
                       : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < LB; i++)
            rb[i] = in ? *reinterpret_cast<const uint4 *>(gB + (uint64_t)(ROWS_PER_PASS * i) * ad + s * BK)
                       : make_uint4(0, 0, 0, 0);
    };
    auto store_slab = [&](int buf) {
        uint8_t *a = ldsA(buf), *bb = ldsB(buf);
#pragma unroll
        for (int i = 0; i < LA; i++)
            *reinterpret_cast<uint4 *>(a + (s_row + ROWS_PER_PASS * i) * PITCH + s_chunk * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; i++)
            *reinterpret_cast<uint4 *>(bb + (s_row + ROWS_PER_PASS * i) * PITCH + s_chunk * 16) = rb[i];
    };

    v16i acc[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < MJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0;

    load_slab(0);
    store_slab(0);
    __syncthreads();
    stamp(1);
    for (uint32_t s = 0; s < n_slabs; s++) {
        const int cur = s & 1;
        if (s + 1 < n_slabs) load_slab(s + 1);
        const uint8_t *A = ldsA(cur) + (wq * (MI * 32) + r) * PITCH + h * 16;
        const uint8_t *B = ldsB(cur) + (wr * (MJ * 32) + r) * PITCH + h * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ks++) {
            v4i a[MI], bf[MJ];
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = *reinterpret_cast<const v4i *>(A + i * 32 * PITCH + ks * 32);
#pragma unroll
            for (int j = 0; j < MJ; j++) bf[j] = *reinterpret_cast<const v4i *>(B + j * 32 * PITCH + ks * 32);
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < MJ; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (s < 8) stamp(2 + s);  // MFMAs of slab s issued
        if (s + 1 < n_slabs) store_slab(cur ^ 1);
        __syncthreads();
    }
    stamp(10);

    // Epilogue.  C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8*(reg >> 2)
    // + 4*(lane >> 5): query index on the registers, store row on the lanes (coalesced writes).
    // The tile's per-query constants (offset, pivot key) go through LDS once: read per element
    // from global memory they were 256 loads per lane and dominated the whole kernel.
    // The filter is a FLOAT compare against the pivot score (a superset of "key <= pivot key":
    // only the rare passing element pays for the key, the atomic and the store) — the first
    // version built the ordered key of all 128 results per lane and spent more instructions in
    // this epilogue (5000) than in the MFMA loop.
    float *q_off_s = reinterpret_cast<float *>(lds_raw);  // [TQ_]   (operand buffers are dead now)
    float *pivot_s = reinterpret_cast<float *>(lds_raw) + TQ_;  // [TQ_]
    for (int i = t; i < TQ_; i += T) {
        q_off_s[i] = q_offsets[q0 + i];
        if (MODE != 0) pivot_s[i] = filt.pivot_scores[q0 + i];
    }
    __syncthreads();
    stamp(11);
    constexpr bool LARGEST = MODE == 1;
    const float never = LARGEST ? -__builtin_huge_valf() : __builtin_huge_valf();
#pragma unroll
    for (int j = 0; j < MJ; j++) {
        const uint64_t row = row0 + wr * (MJ * 32) + j * 32 + r;
        const bool row_ok = row < n_rows;
        // a padding row gets an offset that can never pass the filter
        const float v_off = (MODE == 0 || row_ok) ? v_offsets[row] : never;  // v_offsets is padded like codes[]
#pragma unroll
        for (int i = 0; i < MI; i++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {  // registers 4g .. 4g+3 are four consecutive queries
                const uint32_t ql = wq * (MI * 32) + i * 32 + 8 * g + 4 * h;
                const float4 qo4 = *reinterpret_cast<const float4 *>(q_off_s + ql);
                const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                float sc[4];
#pragma unroll
                for (int e = 0; e < 4; e++) sc[e] = (multiplier * (float)acc[i][j][4 * g + e] + qo[e]) + v_off;
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t q = q0 + ql + e;
                        if (row_ok && q < n_queries) out[(uint64_t)q * out_pitch + row] = sc[e];
                    }
                } else {
                    const float4 pv4 = *reinterpret_cast<const float4 *>(pivot_s + ql);
                    const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
                    // margin d >= 0  <=>  the score is at least as good as the pivot (the sign of an
                    // f32 difference is exact).  One max over the four margins and one compare decide
                    // for the whole group; NaN scores never pass (v_max drops them).
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) d[e] = LARGEST ? sc[e] - pv[e] : pv[e] - sc[e];
                    const float dmax = fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]));
                    if (dmax >= 0.0f) {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if (d[e] >= 0.0f) {
                                const uint32_t q = q0 + ql + e;
                                const uint32_t key = topk_ordered_bits(sc[e], LARGEST);
                                const uint32_t pos = atomicAdd(filt.counters + (uint64_t)q * kCounterStride, 1u);
                                if (pos < kBatchCap)
                                    filt.candidates[(uint64_t)q * kBatchCap + pos] =
                                        ((unsigned long long)key << 32) | (uint32_t)row;
                            }
                        }
                    }
                }
            }
        }
    }
    stamp(12);
}

// ------------------------------------------------------------------------------------------
// Ping-pong kernel (batches of more than 128 queries).  Same arithmetic and epilogue as
// u8_gemm_kernel; what changes is how the matrix pipe is kept fed:
//   * persistent workgroups (one per CU, 8 waves): a workgroup keeps one query tile (256 queries)
//     and walks row tiles xcd + 8*(row_lane + row_lanes*i); the q_tiles workgroups of one
//     (XCD, row_lane) walk the same row tiles, so a row tile comes from HBM once per XCD.
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), no staging registers and
//     no ds_write pass; K is a continuous stream of 64-byte K-tiles (across output tiles too, so
//     the next tile's first K-tiles are in flight during the epilogue) through a ring of RING
//     LDS slots, RING-1 K-tiles ahead, retired with COUNTED vmcnt waits (4 x 32 KiB for the
//     256 x 256 tile, 3 x 40 KiB for the 128 x 512 tile of small batches).
//   * LDS image is lane-linear per DMA instruction (16 rows x 64 B); the 16-byte chunk index is
//     XOR-swizzled with (row>>2)&3 on the SOURCE address and on the fragment read, which makes
//     every ds_read_b128 lane group hit 16 distinct 16-byte slots.
//   * the two wave groups (waves 0-3: queries 0-127, waves 4-7: queries 128-255; one wave of each
//     per SIMD) run half a phase apart: while one group issues its 8 MFMAs (256 cycles) the other
//     reads its fragments and issues DMA.  Raw s_barrier at every half-phase boundary.
// Time is counted in slots (one barrier each).  Group g runs phase p of K-tile u in slots
// 4u+2p+g (reads + DMA issue) and 4u+2p+g+1 (MFMAs).  Hazards, by slot number:
//   RAW  K-tile u+1 is first read in slot 4u+4; every wave retires its DMA of K-tile u+1 with a
//        counted vmcnt at the end of its phase-0 MFMA slot of K-tile u (slots 4u+1 / 4u+2).
//   WAR  B of K-tile u+RING-1 (ring slot of K-tile u-1) is issued in slots 4u / 4u+1; the last reads
//        of K-tile u-1's B were issued in slot 4u-3 and waited for in slot 4u-2.  A of K-tile
//        u+RING-1 is issued in slots 4u+2 / 4u+3; the last reads of K-tile u-1's A were issued in
//        slot 4u-1 and waited for in slot 4u.  (Nothing in the argument depends on RING beyond
//        "the slot being refilled is the one K-tile u-1 used".)
// Measured alternatives (in-kernel timeline, tools/gemm_timeline.py): DMA issued inside the MFMA
// slot: K loop +10 % (behind the 4th MFMA) / whole call +1..3 % (behind the 8th); one 16-MFMA slot per K-tile and group (half the barriers): K loop +5..15 %
// (and the 128-query tile, 8 MFMAs per K-tile in one slot: 2.14-2.59 ms vs 1.87-2.08 for 4..128 queries).
// With DMA, barriers and fragment reads all removed the K loop still takes 1.2-1.4x the nominal
// 32 cycles per MFMA in s_memtime ticks: the chip runs this kernel at about 1.8-2.0 GHz.
constexpr int PP_KT = 64;  // K-tile bytes per row
// Batch size from which the query-streaming kernel is preferred, by 128-byte K-blocks per row (measured, whole
// topk_batch(30) calls at 7.68 GB of rows; below it several 128-query tiles of the row-streaming kernel, or the
// ping-pong kernel where only 64-query tiles fit).  Round 3, with the 16x16x64 form of the kernel for rows of up to
// 1024 bytes (profiles/r03_qs_experiments.txt), ms, row-streaming / query-streaming:
//   rows of 256 B,  30M:   257 q  3.96 / 4.45    385 q  5.10 / 4.73    704 q  7.81 / 6.98    960 q  9.80 / 8.07
//   rows of 384 B,  20M:   257 q  3.64 / 3.69    385 q  4.60 / 3.97    704 q  7.07 / 5.98    960 q  8.96 / 7.16
//   rows of 512 B,  15M:   192 q  2.34 / 2.73    257 q  3.44 / 3.30    385 q  4.35 / 3.70    704 q  6.77 / 5.53
//   rows of 768 B,  10M:   192 q  2.19 / 2.35    257 q  3.24 / 2.88    385 q  4.10 / 3.39    704 q  6.36 / 5.07
//   rows of 1024 B, 7.5M:  192 q  2.22 / 2.21    257 q  3.41 / 2.74    385 q  4.76 / 3.25    704 q  8.61 / 4.89
//   rows <= 1536 B: 12.5M x 1536: 256 q  pp 5.84 qs 6.34; 384 q  pp 10.3 qs 8.8; 640 q  pp 16.1 qs 13.8   (round 2)
// i.e. from the third 128-query tile on (the fourth for rows of up to 384 bytes); the round-2 thresholds (960 / 704)
// dated from before that round's block-change and epilogue work and this round's matrix instruction.  With chunks of 32
// queries for batches of up to 256 (a chunk for every wave) the second tile goes the same way on rows past 768 bytes:
//   rows of 768 B,  10M:   129 q  2.18 / 2.06    192 q  2.2-2.5 / 2.20    256 q  2.31 / 2.49     (kept on row-streaming)
//   rows of 1024 B, 7.5M:  129 q  2.22 / 2.02    192 q  2.26 / 2.11       256 q  2.80 / 2.37
//   rows of 1536 B, 12.5M: 129 q  5.44 / 4.88    192 q  5.53 / 5.23       256 q  6.87 / 5.90
inline uint64_t qs_min_queries(uint32_t nkb) { return nkb <= 3 ? 385 : nkb <= 6 ? 257 : 129; }
// Workgroup shapes (8 waves as 2 query groups x 4 row groups; a wave owns MI x MJ 32x32 tiles):
//   <4,2>: 256 queries x 256 rows, ring of 4 x 32 KiB  -- more than 128 queries, MFMA-bound
//   <2,4>: 128 queries x 512 rows, ring of 3 x 40 KiB  -- up to 128 queries: the store is streamed
//          once, so the shape that moves the most row bytes per slot wins (HBM-bound)
template <int MI, int MJ> struct PpShape {
    static constexpr int TQH = MI * 32, TQW = 2 * TQH;  // queries per wave group / workgroup
    static constexpr int RW = MJ * 32, TR = 4 * RW;     // rows per wave / workgroup
    static constexpr uint32_t GA = TQW / 128, GB = TR / 128;
    static constexpr int BYTES_B = TR * PP_KT, BYTES_A = TQW * PP_KT, SLOT = BYTES_B + BYTES_A;
    static constexpr size_t CONSTS = 3 * 256 * sizeof(float) + 64;  // q_off, pivot, B_q, 8 counters
    static constexpr int RING = (size_t)4 * SLOT + CONSTS <= 160 * 1024 ? 4 : 3;
    static constexpr size_t LDS = (size_t)RING * SLOT + CONSTS;
};

#define PP_BARRIER()                          \
    do {                                      \
        asm volatile("" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);    \
        __builtin_amdgcn_s_barrier();         \
        __builtin_amdgcn_sched_barrier(0);    \
        asm volatile("" ::: "memory");        \
    } while (0)

__device__ __forceinline__ void pp_glds16(const uint8_t *src, uint8_t *lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}
// The same instruction behind the compiler's back (wave-uniform base + 32-bit lane offset, nontemporal).  A kernel whose
// LDS reads must not wait for the DMA needs it: with the builtin the compiler cannot tell the slab being filled from the
// slab being read and puts s_waitcnt vmcnt(0) in front of every ds_read.  The caller orders the reads itself
// (s_waitcnt vmcnt(0) + barrier before a slab is read).  m0 is saved and restored around it.
__device__ __forceinline__ void glds16_nt_unordered(const uint8_t *base, uint32_t lane_offset, uint32_t lds_addr) {
    uint32_t saved_m0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(saved_m0)
                 : "v"(lane_offset), "s"(base), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void pp_wait_vm(uint32_t n) {  // n is wave-uniform; rounded DOWN to a supported count
    if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A wave owns MI x MJ 32x32 accumulator tiles (PpShape): <4,2> is the 256 x 256 workgroup tile
// described above; <2,4> = 128 queries x 512 rows serves batches of up to 128 queries, where the
// store is streamed exactly once and the shape that moves the most row bytes per slot wins
// (measured: K loop at 5.5 TB/s of row bytes; 1.76-1.85 ms per 10M x 768 for 4..64 queries against
// 1.86-2.0 with a 128 x 256 tile; splitting its phases by row fragments instead of query fragments
// to balance the read slots changed nothing; the nt cache policy on its row DMA cost 30 %: a
// 64-byte K-tile is half a cache line, and the other half must still be in L2 one K-tile later).
template <int MODE, bool LOW, int MI, int MJ>
__global__ __launch_bounds__(512) void u8_gemm_pp_kernel(const uint8_t *__restrict__ codes,
                                                        const float *__restrict__ v_offsets,
                                                        const uint8_t *__restrict__ qcodes, uint32_t q_pitch,
                                                        const float *__restrict__ q_offsets, float multiplier,
                                                        uint32_t n_rows, uint32_t n_queries, uint32_t ad,
                                                        uint32_t q_tiles, uint32_t row_lanes,
                                                        float *__restrict__ out, uint64_t out_pitch,
                                                        BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int g = wave >> 2, wr = wave & 3, r = lane & 31, h = lane >> 5;
    const uint32_t b = blockIdx.x, xcd = b & 7u, j = b >> 3;
    if (j >= row_lanes * q_tiles) return;
    const uint32_t q_tile = j % q_tiles, row_lane = j / q_tiles;
    using Shape = PpShape<MI, MJ>;
    constexpr int TQH = Shape::TQH;  // queries per wave group
    constexpr int TQW = Shape::TQW;  // queries per workgroup
    constexpr int HALF = MI / 2;     // query fragments per phase
    constexpr int RW = Shape::RW;    // rows per wave
    constexpr int TR = Shape::TR;    // rows per workgroup tile
    constexpr uint32_t GA = Shape::GA, GB = Shape::GB;  // DMA instructions per wave and K-tile (queries / rows)
    constexpr int BYTES_B = Shape::BYTES_B, SLOT = Shape::SLOT, RING = Shape::RING, AHEAD = RING - 1;
    const uint32_t q0 = q_tile * TQW;
    const uint32_t n_rtiles = (n_rows + TR - 1) / TR;
    const uint32_t first = xcd + 8 * row_lane, step = 8 * row_lanes;
    if (first >= n_rtiles) return;
    const uint32_t my_tiles = __builtin_amdgcn_readfirstlane((n_rtiles - first + step - 1) / step);
    const uint32_t nkt = __builtin_amdgcn_readfirstlane((ad + PP_KT - 1) / PP_KT);
    const uint32_t total = my_tiles * nkt;  // K-tiles in this workgroup's stream

    unsigned long long *stamps = QAMD_GEMM_STAMPS();
    if (stamps) stamps = (blockIdx.x < kStampBlocks && lane == 0) ? stamps + ((uint64_t)blockIdx.x * 8 + wave) * 16 : nullptr;
    auto stamp = [&](int slot) {
        if (stamps) stamps[slot] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);
    const bool dbg = QAMD_GEMM_STAMPS() != nullptr;  // wave-uniform
    unsigned long long dbg_flagged = 0;         // (query group, row) lanes sent to the exact epilogue

    // The workgroup's query tile never changes: its per-query constants are staged once.
    float *q_off_s = reinterpret_cast<float *>(lds_raw + (size_t)RING * SLOT);  // [256]
    float *pivot_s = q_off_s + 256;                                                   // [256]
    int *bq_s = reinterpret_cast<int *>(pivot_s + 256);                               // [256] integer query bounds
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(bq_s + 256) + wave;             // this wave's append counter
    // (arrays sized for the 256-query tile; the 128-query tile uses their first halves)
    if (MODE != 0 && lane == 0) *wcount_s = 0;
    constexpr bool LARGEST = MODE == 1;
    if (t < TQW) {
        const float qo = q_offsets[q0 + t];
        q_off_s[t] = qo;
        if (MODE != 0) {
            const float pv = filt.pivot_scores[q0 + t];
            pivot_s[t] = pv;
            int bq = pp_bound<LOW>(pv - qo, fabsf(pv) + fabsf(qo), multiplier, 1);  // LOW: s <= T  <=>  s - (T+1) < 0
            if (__builtin_isinf(pv))  // padding query (nothing may pass) or a degenerate pivot (everything does)
                bq = ((pv > 0.0f) == LARGEST) == LOW ? -(int)kPpLim : (int)kPpLim;
            bq_s[t] = bq;
        }
    }
    __syncthreads();

    // DMA addressing: instruction `idx` of an operand covers rows idx*128 + wave*16 + lane/4;
    // lane writes LDS (row, position lane%4) and fetches chunk (lane%4) ^ ((row>>2)&3).
    const uint32_t dma_row = wave * 16 + (lane >> 2);
    const uint32_t dma_chunk = (((uint32_t)lane & 3u) ^ ((uint32_t)lane >> 4)) * 16;
    // per-lane source pointers kept across the loop and advanced by wave-uniform amounts (one
    // 64-bit add per DMA instruction in the read slot instead of a 64-bit multiply-add chain)
    const uint8_t *src_a0 = qcodes + ((uint64_t)q0 + dma_row) * q_pitch + dma_chunk;
    const uint8_t *src_b = codes + ((uint64_t)first * TR + dma_row) * ad + dma_chunk;
    const uint64_t tile_stride_b = (uint64_t)step * TR * ad, half_b = (uint64_t)128 * ad, half_a = (uint64_t)128 * q_pitch;
    uint32_t pf_kt = 0, pf_koff = 0, pf_u = 0, pf_slot = 0;  // pf_slot = pf_u % RING
#ifdef QAMD_GEMM_ABLATION
    const uint32_t dbgf = g_gemm_dbg;
#else
    constexpr uint32_t dbgf = 0;
#endif
    auto issue_B = [&](uint32_t idx_lo, uint32_t idx_hi) {
        const uint8_t *src = src_b + pf_koff;
        uint8_t *dst = lds_raw + pf_slot * SLOT + wave * 1024;
        if (dbgf & 2u) return;
#pragma unroll
        for (uint32_t idx = idx_lo; idx < idx_hi; idx++) pp_glds16(src + idx * half_b, dst + idx * 128 * PP_KT);
    };
    auto issue_A = [&]() {  // second half of a K-tile's DMA: advances the prefetch position
        const uint8_t *src = src_a0 + pf_koff;
        uint8_t *dst = lds_raw + pf_slot * SLOT + BYTES_B + wave * 1024;
        if (!(dbgf & 1u)) {
#pragma unroll
            for (uint32_t idx = 0; idx < GA; idx++) pp_glds16(src + idx * half_a, dst + idx * 128 * PP_KT);
        }
        pf_u++;
        pf_slot = pf_slot + 1 == (uint32_t)RING ? 0u : pf_slot + 1;
        pf_koff += PP_KT;
        if (++pf_kt == nkt) {
            pf_kt = 0;
            pf_koff = 0;
            src_b += tile_stride_b;
        }
    };
    // fragment read offsets: chunk 2*ks + h of row r (+ multiples of 16 rows), swizzled
    const uint32_t swz = ((uint32_t)r >> 2) & 3u;
    const uint32_t off0 = (((uint32_t)h) ^ swz) * 16, off1 = ((2u + (uint32_t)h) ^ swz) * 16;
    const uint32_t fragA = BYTES_B + (g * TQH + r) * PP_KT, fragB = (wr * RW + r) * PP_KT;

    // v_offset of this lane's MJ rows (jj = 0 .. MJ-1) for the tile about to start.  Loaded one tile
    // ahead by inline asm so that the compiler attaches no wait to it (next to LDS-DMA it would
    // drain everything with vmcnt(0)); the loads are older than the DMA that follows, so the
    // counted waits of the K loop retire them (K-tile 1's wait at the latest: the launcher sends
    // stores with fewer than three K-tiles per row to u8_gemm_kernel).  The values are only
    // touched (copied, used) after the K loop; check the .s when editing this (a register copy
    // placed before the data has landed would copy garbage).
    float vo_next[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // [MJ] used
    auto load_voff = [&](uint32_t tile_idx) {
        const float *p0 = v_offsets + (uint64_t)tile_idx * TR + wr * RW + r;  // padded like codes[]
        if (MJ == 2)
            asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %2, off offset:128"
                         : "=&v"(vo_next[0]), "=&v"(vo_next[1])
                         : "v"(p0)
                         : "memory");
        else
            asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %4, off offset:128\n\t"
                         "global_load_dword %2, %4, off offset:256\n\tglobal_load_dword %3, %4, off offset:384"
                         : "=&v"(vo_next[0]), "=&v"(vo_next[1]), "=&v"(vo_next[2]), "=&v"(vo_next[3])
                         : "v"(p0)
                         : "memory");
    };
    if (MODE != 0) load_voff(first);

    // prologue: K-tiles 0 .. AHEAD-1 in flight, K-tile 0 retired
    for (int k = 0; k < AHEAD; k++)
        if (pf_u < total) {
            issue_B(0, GB);
            issue_A();
        }
    {
        uint32_t n0 = 0;
        for (int v = 1; v < AHEAD; v++) n0 += (uint32_t)v < total ? GB + GA : 0u;
        pp_wait_vm(n0);
    }
    PP_BARRIER();
    stamp(1);
    if (g == 1) PP_BARRIER();  // group 1 runs one slot behind

    const float never = LARGEST ? -__builtin_huge_valf() : __builtin_huge_valf();
    uint32_t tile = first, u = 0, rd_slot = 0;  // rd_slot = u % RING
    for (uint32_t ti = 0; ti < my_tiles; ti++, tile += step) {
    v16i acc[MI][MJ];
    float vo_cur[MJ];
    int br[MJ];
#pragma unroll
    for (int jj = 0; jj < MJ; jj++) {
        vo_cur[jj] = 0.0f;
        br[jj] = 0;
    }
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int jj = 0; jj < MJ; jj++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][jj][e] = 0;
    } else {
        if (MJ == 2) asm volatile("" : "+v"(vo_next[0]), "+v"(vo_next[1]));  // ordered after every wait above
        else asm volatile("" : "+v"(vo_next[0]), "+v"(vo_next[1]), "+v"(vo_next[2]), "+v"(vo_next[3]));
#pragma unroll
        for (int jj = 0; jj < MJ; jj++) vo_cur[jj] = vo_next[jj];
        if (ti + 1 < my_tiles) load_voff(tile + step);
        const uint64_t row_a = (uint64_t)tile * TR + wr * RW + r;
#pragma unroll
        for (int jj = 0; jj < MJ; jj++)
            br[jj] = row_a + 32 * jj < n_rows ? pp_bound<LOW>(-vo_cur[jj], fabsf(vo_cur[jj]), multiplier, 0)
                                              : (LOW ? -(int)kPpLim : (int)kPpLim);
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const v4i bq4 = *reinterpret_cast<const v4i *>(bq_s + g * TQH + i * 32 + 8 * gq + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int jj = 0; jj < MJ; jj++) acc[i][jj][4 * gq + e] = -(bq4[e] + br[jj]);
            }
    }
    if (ti < 4) stamp(2 + 3 * (int)ti);  // tile set up
    for (uint32_t kt = 0; kt < nkt; kt++, u++) {
        const uint8_t *slot = lds_raw + rd_slot * SLOT;
        rd_slot = rd_slot + 1 == (uint32_t)RING ? 0u : rd_slot + 1;
        const uint8_t *pA = slot + fragA, *pB = slot + fragB;
        const bool more = u + AHEAD < total;
        uint32_t nw = more ? GB : 0u;  // DMA instructions younger than K-tile u+1's at the phase-0 wait
        for (int v = 2; v < AHEAD; v++) nw += u + v < total ? GB + GA : 0u;
        // ---- phase 0: the first half of the query fragments and all row fragments
        v4i a[HALF][2], bf[MJ][2];
#pragma unroll
        for (int i = 0; i < MJ; i++) {
            bf[i][0] = *reinterpret_cast<const v4i *>(pB + i * 32 * PP_KT + off0);
            bf[i][1] = *reinterpret_cast<const v4i *>(pB + i * 32 * PP_KT + off1);
        }
#pragma unroll
        for (int i = 0; i < HALF; i++) {
            a[i][0] = *reinterpret_cast<const v4i *>(pA + i * 32 * PP_KT + off0);
            a[i][1] = *reinterpret_cast<const v4i *>(pA + i * 32 * PP_KT + off1);
        }
        if (more) issue_B(0, GB);
        PP_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < HALF; i++)
#pragma unroll
                for (int jj = 0; jj < MJ; jj++)
                    acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i][ks], bf[jj][ks], acc[i][jj], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        pp_wait_vm(nw);  // retires this wave's DMA of K-tile u+1
        PP_BARRIER();
        // ---- phase 1: the other query fragments against the row fragments already in registers
#pragma unroll
        for (int i = 0; i < HALF; i++) {
            a[i][0] = *reinterpret_cast<const v4i *>(pA + (HALF + i) * 32 * PP_KT + off0);
            a[i][1] = *reinterpret_cast<const v4i *>(pA + (HALF + i) * 32 * PP_KT + off1);
        }
        if (more) issue_A();
        PP_BARRIER();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int i = 0; i < HALF; i++)
#pragma unroll
                for (int jj = 0; jj < MJ; jj++)
                    acc[HALF + i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i][ks], bf[jj][ks], acc[HALF + i][jj], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        PP_BARRIER();
    }

        // ---- tile finished.  Both groups run the epilogue together (group 0 waits one slot for
        // group 1's last MFMAs; group 1 falls one slot behind again afterwards).
        if (g == 0) PP_BARRIER();
        if (ti < 4) stamp(3 + 3 * (int)ti);  // K loop done
        const uint64_t row0 = (uint64_t)tile * TR;
        // opaque per tile: keeps per-query output addresses and everything else the epilogue
        // derives from the lane / wave id from being hoisted out of the K loop as loop invariants
        // (with 128 accumulators live, every hoisted invariant is a spill in that loop)
        uint32_t q0_e = q0, wave_e = (uint32_t)wave, lane_e = (uint32_t)lane;
        asm volatile("" : "+s"(q0_e), "+s"(wave_e), "+v"(lane_e));
        uint4 *wave_list = MODE != 0 ? filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * 8 + wave_e) * filt.wave_cap : nullptr;
        const uint32_t g_e = wave_e >> 2, wr_e = wave_e & 3u, r_e = lane_e & 31u, h_e = lane_e >> 5;
#pragma unroll
        for (int jj = 0; jj < MJ; jj++) {
            const uint64_t row = row0 + wr_e * RW + jj * 32 + r_e;
            const bool row_ok = row < n_rows;
            float v_off;
            if (MODE == 0) v_off = v_offsets[row];  // padded like codes[]
            else v_off = row_ok ? vo_cur[jj] : never;
            const int brj = br[jj];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                if (MODE == 0) __builtin_amdgcn_sched_barrier(0);  // one accumulator tile at a time: no load clustering
                if (MODE != 0) {
                    // two-level test: one sign test for the whole 32 x 32 accumulator tile first (about
                    // three quarters of them hold no candidate at all), then per group of four
                    int all = acc[i][jj][0];
#pragma unroll
                    for (int e = 1; e < 16; e++) all = LOW ? (all | acc[i][jj][e]) : (all & acc[i][jj][e]);
                    if (!__builtin_amdgcn_readfirstlane(__ballot(LOW ? all < 0 : all >= 0) != 0)) continue;
                }
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {  // registers 4gq .. 4gq+3 are four consecutive queries
                    const uint32_t ql = g_e * TQH + i * 32 + 8 * gq + 4 * h_e;
                    if (MODE == 0) {
                        const float4 qo4 = *reinterpret_cast<const float4 *>(q_off_s + ql);
                        const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float sc = (multiplier * (float)acc[i][jj][4 * gq + e] + qo[e]) + v_off;
                            const uint32_t q = q0_e + ql + e;
                            if (row_ok && q < n_queries) out[(uint64_t)q * out_pitch + row] = sc;
                        }
                    } else {
                        const int a0 = acc[i][jj][4 * gq], a1 = acc[i][jj][4 * gq + 1], a2 = acc[i][jj][4 * gq + 2],
                                  a3 = acc[i][jj][4 * gq + 3];
                        const bool may_pass = LOW ? ((a0 | a1 | a2 | a3) < 0) : ((a0 & a1 & a2 & a3) >= 0);
                        if (dbg) dbg_flagged += __builtin_popcountll(__ballot(may_pass));
                        if (may_pass) {  // rare: the exact f32 epilogue for these four (query, row) pairs
                            const v4i bq4 = *reinterpret_cast<const v4i *>(bq_s + ql);
                            const float4 qo4 = *reinterpret_cast<const float4 *>(q_off_s + ql);
                            const float4 pv4 = *reinterpret_cast<const float4 *>(pivot_s + ql);
                            const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                            const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
                            const int av[4] = {a0, a1, a2, a3};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int s_int = av[e] + bq4[e] + brj;  // the plain integer dot product
                                const float sc = (multiplier * (float)s_int + qo[e]) + v_off;
                                const float d = LARGEST ? sc - pv[e] : pv[e] - sc;
                                if (d >= 0.0f) {
                                    // wave-private list: an LDS counter, a fire-and-forget 16-byte store
                                    const uint32_t pos = atomicAdd(wcount_s, 1u);
                                    if (pos < filt.wave_cap)
                                        wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row,
                                                                    filt.query_base + q0_e + ql + e, 0u);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (ti < 3) stamp(4 + 3 * (int)ti);  // epilogue done (slot 13 is HW_ID)
        if (g == 1 && ti + 1 < my_tiles) PP_BARRIER();  // fall one slot behind again
    }
    stamp(15);
    if (stamps) stamps[14] = dbg_flagged;
    if (stamps) stamps[13] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8]
    if (MODE != 0 && lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = *wcount_s;
}

// ------------------------------------------------------------------------------------------
// Row-streaming kernel (batches whose query tile fits in LDS: every batch of up to 128 queries at
// <= 1152 code bytes per row, 64 at <= 2304, 32 at <= 4608).  With so few queries the GEMM is
// HBM-bound: each store row is needed by ONE wave only, so the rows do not go through LDS at all.
//   * the workgroup's whole query tile (32*MI queries x the padded row length) is staged into LDS
//     ONCE (pitch + 16 B: conflict-free ds_read_b128); no ring, no barrier after that — the 8 waves
//     run independently and hide each other's HBM latency;
//   * a wave owns 64 store rows at a time and streams them straight from HBM into its MFMA B
//     operand registers: the K index of an MFMA is only a summation index, so lane (r, h) takes
//     the 64 contiguous bytes [128 kb + 64 h, +64) of row r as the B fragments of the four MFMA
//     k-steps of K-block kb (and reads the same bytes of query r from LDS for A) — whole 128-byte
//     lines per row and K-block, one K-block ahead in a second register set (8 KiB per wave in
//     flight), continuous across row chunks so the epilogue of one chunk runs under the loads of
//     the next;
//   * bytes past the end of a row (last K-block when the row length is not a multiple of 128)
//     belong to the next row or to the store's zero padding: the query's LDS image is zero there,
//     so they add nothing (integer arithmetic);
//   * accumulators start at -(B_q + B_row) and the epilogue is the ping-pong kernel's (integer sign
//     pre-filter, exact f32 epilogue for the few that may pass, wave-private candidate lists).
// Workgroups b, b+8, ... of one XCD that share a row lane walk the same rows with different query
// tiles (q_tiles > 1: the row bytes come from HBM once and from that XCD's L2 afterwards).
template <int MI> struct RsShape {
    static constexpr int MJ = 2;              // 32-row fragments per wave
    static constexpr int TQ = 32 * MI;        // queries per workgroup
    static constexpr int CHUNK = 32 * MJ;     // rows per wave and chunk
    static constexpr int TR = 8 * CHUNK;      // rows per workgroup and step
    static constexpr int KB = 128;            // K-block bytes per row
    static constexpr size_t CONSTS = 3 * 128 * sizeof(float) + 64;
    static constexpr size_t SLOTS = 8 * 4096;  // one wave-private transposition slot per wave
    static size_t lds_bytes(uint32_t ad) { return (size_t)TQ * (round_up((uint64_t)ad, KB) + 16) + CONSTS + SLOTS; }
};

template <int MODE, bool LOW, int MI, bool NT>
__global__ __launch_bounds__(512) void u8_gemm_rs_kernel(const uint8_t *__restrict__ codes,
                                                        const float *__restrict__ v_offsets,
                                                        const uint8_t *__restrict__ qcodes, uint32_t q_pitch,
                                                        const float *__restrict__ q_offsets, float multiplier,
                                                        uint32_t n_rows, uint32_t n_queries, uint32_t ad,
                                                        uint32_t q_tiles, uint32_t row_lanes,
                                                        float *__restrict__ out, uint64_t out_pitch,
                                                        BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    using Shape = RsShape<MI>;
    constexpr int MJ = Shape::MJ, TQ = Shape::TQ, CHUNK = Shape::CHUNK, TR = Shape::TR, KB = Shape::KB;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t b = blockIdx.x, xcd = b & 7u, j = b >> 3;
    if (j >= row_lanes * q_tiles) return;
    const uint32_t q_tile = j % q_tiles, row_lane = j / q_tiles;
    const uint32_t q0 = q_tile * TQ;
    const uint32_t nkb = __builtin_amdgcn_readfirstlane((ad + KB - 1) / KB);
    const uint32_t PA = nkb * KB + 16;  // LDS pitch of a query
    const uint32_t n_rtiles = (n_rows + TR - 1) / TR;
    const uint32_t first = xcd + 8 * row_lane, step = 8 * row_lanes;
    const uint32_t my_tiles =
        __builtin_amdgcn_readfirstlane(first < n_rtiles ? (n_rtiles - first + step - 1) / step : 0u);

    float *q_off_s = reinterpret_cast<float *>(lds_raw + (size_t)TQ * PA);  // [128]
    float *pivot_s = q_off_s + 128;                                         // [128]
    int *bq_s = reinterpret_cast<int *>(pivot_s + 128);                     // [128] integer query bounds
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(bq_s + 128) + wave;   // this wave's append counter
    constexpr bool LARGEST = MODE == 1;
    if (MODE != 0 && lane == 0) *wcount_s = 0;
    if (t < TQ) {
        const float qo = q_offsets[q0 + t];
        q_off_s[t] = qo;
        if (MODE != 0) {
            const float pv = filt.pivot_scores[q0 + t];
            pivot_s[t] = pv;
            int bq = pp_bound<LOW>(pv - qo, fabsf(pv) + fabsf(qo), multiplier, 1);
            if (__builtin_isinf(pv)) bq = ((pv > 0.0f) == LARGEST) == LOW ? -(int)kPpLim : (int)kPpLim;
            bq_s[t] = bq;
        }
    }
    {  // the query tile, zero beyond the batch's pitch
        const uint32_t per_row = nkb * (KB / 16);
        for (uint32_t idx = t; idx < (uint32_t)TQ * per_row; idx += 512) {
            const uint32_t row = idx / per_row, k = (idx % per_row) * 16;
            v4i v = {0, 0, 0, 0};
            if (k < q_pitch) v = *reinterpret_cast<const v4i *>(qcodes + (uint64_t)(q0 + row) * q_pitch + k);
            *reinterpret_cast<v4i *>(lds_raw + row * PA + k) = v;
        }
    }
    __syncthreads();
    if (my_tiles == 0) {
        if (MODE != 0 && lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = 0;
        return;
    }

    // B stream.  HBM -> registers: fully coalesced, one touch per 128-byte line (lane l takes bytes
    // [16 (l % 8), +16) of chunk rows l / 8 + 8 i, i = 0..7), which is what lets the nt policy work
    // (measured on pure-load kernels, tools/tune_stream.py: 6.96 TB/s this way; 6.25 TB/s with the
    // operand layout fetched directly — each line touched by four instructions — and 3.5 TB/s if
    // THAT is marked nt).  Registers -> operand layout through a wave-private 4 KiB LDS slot (32
    // rows x 128 B, 16-byte chunk index XOR (row & 7): conflict-free both ways), one half chunk
    // at a time; LDS operations of one wave execute in order, so the slot needs no barrier.
    const uint64_t row_bytes = ad;
    const uint32_t l8 = (uint32_t)lane & 7u, lrow = (uint32_t)lane >> 3;
    const uint8_t *pf0 = codes + ((uint64_t)first * TR + wave * CHUNK + lrow) * row_bytes + 16 * l8;
    const uint64_t i_stride = (uint64_t)8 * row_bytes;
    const uint64_t wrap_advance = (uint64_t)step * TR * row_bytes - (uint64_t)nkb * KB;
    uint32_t pf_kb = 0, pf_left = my_tiles * nkb - 1;  // K-blocks after the one pf0 points at
    // NT: rows read exactly once (one query tile); with several query tiles the XCD's L2 serves the others
    v4i L[8];
    // Always issues its eight loads (past the end of the stream it re-reads the last K-block): a
    // conditional load would make the compiler drain vmcnt(0) where the two paths join.
    auto prefetch = [&]() {
        if (NT) {
#pragma unroll
            for (int i = 0; i < 8; i++) L[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(pf0 + i * i_stride));
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) L[i] = *reinterpret_cast<const v4i *>(pf0 + i * i_stride);
        }
        uint64_t adv = KB;
        if (++pf_kb == nkb) {
            pf_kb = 0;
            adv += wrap_advance;
        }
        if (pf_left == 0) {
            adv = 0;
            pf_kb = 0;
        } else {
            pf_left--;
        }
        pf0 += adv;
    };
    uint8_t *slot = lds_raw + (size_t)TQ * PA + Shape::CONSTS + (size_t)wave * 4096;
    uint8_t *wr = slot + lrow * 128 + ((l8 ^ (lrow & 7u)) * 16);  // + 1024 i
    const uint32_t rbase = (uint32_t)r * 128u + (((4u * (uint32_t)h) ^ ((uint32_t)r & 4u)) * 16u);
    const uint8_t *rd = slot + rbase;  // + 16 (x ^ (r & 3))
    const uint32_t r3 = (uint32_t)r & 3u;
    v4i F[MJ][4];
    auto transpose = [&]() {
#pragma unroll
        for (int jj = 0; jj < MJ; jj++) {
#pragma unroll
            for (int i = 0; i < 4; i++) *reinterpret_cast<v4i *>(wr + 1024 * i) = L[4 * jj + i];
#pragma unroll
            for (int x = 0; x < 4; x++) F[jj][x] = *reinterpret_cast<const v4i *>(rd + 16u * ((uint32_t)x ^ r3));
        }
    };
    const uint8_t *a_base = lds_raw + r * PA + 64 * h;

    prefetch();
    float vo_next[MJ];
    if (MODE != 0) {
#pragma unroll
        for (int jj = 0; jj < MJ; jj++) vo_next[jj] = v_offsets[(uint64_t)first * TR + wave * CHUNK + jj * 32 + r];
    }
    const float never = LARGEST ? -__builtin_huge_valf() : __builtin_huge_valf();
    uint32_t tile = first;
    for (uint32_t ti = 0; ti < my_tiles; ti++, tile += step) {
        v16i acc[MI][MJ];
        float vo_cur[MJ];
        int br[MJ];
        const uint64_t row_a = (uint64_t)tile * TR + wave * CHUNK + r;
#pragma unroll
        for (int jj = 0; jj < MJ; jj++) {
            vo_cur[jj] = 0.0f;
            br[jj] = 0;
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int jj = 0; jj < MJ; jj++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[i][jj][e] = 0;
        } else {
#pragma unroll
            for (int jj = 0; jj < MJ; jj++) vo_cur[jj] = vo_next[jj];
            {
                const uint32_t nt = ti + 1 < my_tiles ? tile + step : tile;  // unconditional load (see prefetch)
#pragma unroll
                for (int jj = 0; jj < MJ; jj++) vo_next[jj] = v_offsets[(uint64_t)nt * TR + wave * CHUNK + jj * 32 + r];
            }
#pragma unroll
            for (int jj = 0; jj < MJ; jj++)
                br[jj] = row_a + 32 * jj < n_rows ? pp_bound<LOW>(-vo_cur[jj], fabsf(vo_cur[jj]), multiplier, 0)
                                                  : (LOW ? -(int)kPpLim : (int)kPpLim);
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const v4i bq4 = *reinterpret_cast<const v4i *>(bq_s + i * 32 + 8 * gq + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; e++)
#pragma unroll
                        for (int jj = 0; jj < MJ; jj++) acc[i][jj][4 * gq + e] = -(bq4[e] + br[jj]);
                }
        }
        for (uint32_t kb = 0; kb < nkb; kb++) {
            transpose();  // waits for L, leaves it free
            prefetch();   // next K-block (of this chunk or the next) under this one's MFMAs
            const uint8_t *pa = a_base + kb * KB;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                v4i a[MI];
#pragma unroll
                for (int i = 0; i < MI; i++) a[i] = *reinterpret_cast<const v4i *>(pa + (uint32_t)i * 32u * PA + 16 * x);
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int jj = 0; jj < MJ; jj++)
                        acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], F[jj][x], acc[i][jj], 0, 0, 0);
            }
        }

        // ---- epilogue of this wave's 64 rows (see u8_gemm_pp_kernel)
        const uint64_t row0 = (uint64_t)tile * TR + wave * CHUNK;
        uint32_t q0_e = q0, wave_e = (uint32_t)wave, lane_e = (uint32_t)lane;
        asm volatile("" : "+s"(q0_e), "+s"(wave_e), "+v"(lane_e));
        uint4 *wave_list = MODE != 0 ? filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * 8 + wave_e) * filt.wave_cap : nullptr;
        const uint32_t r_e = lane_e & 31u, h_e = lane_e >> 5;
#pragma unroll
        for (int jj = 0; jj < MJ; jj++) {
            const uint64_t row = row0 + jj * 32 + r_e;
            const bool row_ok = row < n_rows;
            float v_off;
            if (MODE == 0) v_off = v_offsets[row];  // padded like codes[]
            else v_off = row_ok ? vo_cur[jj] : never;
            const int brj = br[jj];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                if (MODE == 0) __builtin_amdgcn_sched_barrier(0);
                if (MODE != 0) {
                    int all = acc[i][jj][0];
#pragma unroll
                    for (int e = 1; e < 16; e++) all = LOW ? (all | acc[i][jj][e]) : (all & acc[i][jj][e]);
                    if (!__builtin_amdgcn_readfirstlane(__ballot(LOW ? all < 0 : all >= 0) != 0)) continue;
                }
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const uint32_t ql = i * 32 + 8 * gq + 4 * h_e;
                    if (MODE == 0) {
                        const float4 qo4 = *reinterpret_cast<const float4 *>(q_off_s + ql);
                        const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float sc = (multiplier * (float)acc[i][jj][4 * gq + e] + qo[e]) + v_off;
                            const uint32_t q = q0_e + ql + e;
                            if (row_ok && q < n_queries) out[(uint64_t)q * out_pitch + row] = sc;
                        }
                    } else {
                        const int a0 = acc[i][jj][4 * gq], a1 = acc[i][jj][4 * gq + 1], a2 = acc[i][jj][4 * gq + 2],
                                  a3 = acc[i][jj][4 * gq + 3];
                        const bool may_pass = LOW ? ((a0 | a1 | a2 | a3) < 0) : ((a0 & a1 & a2 & a3) >= 0);
                        if (may_pass) {
                            const v4i bq4 = *reinterpret_cast<const v4i *>(bq_s + ql);
                            const float4 qo4 = *reinterpret_cast<const float4 *>(q_off_s + ql);
                            const float4 pv4 = *reinterpret_cast<const float4 *>(pivot_s + ql);
                            const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                            const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
                            const int av[4] = {a0, a1, a2, a3};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int s_int = av[e] + bq4[e] + brj;  // the plain integer dot product
                                const float sc = (multiplier * (float)s_int + qo[e]) + v_off;
                                const float d = LARGEST ? sc - pv[e] : pv[e] - sc;
                                if (d >= 0.0f) {
                                    const uint32_t pos = atomicAdd(wcount_s, 1u);
                                    if (pos < filt.wave_cap)
                                        wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row,
                                                                    filt.query_base + q0_e + ql + e, 0u);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (MODE != 0 && lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = *wcount_s;
}

// ------------------------------------------------------------------------------------------
// Query-streaming kernel (many queries, rows of up to 1152 code bytes).  The roles of the two
// operands are swapped against the row-streaming kernel: a workgroup keeps a block of 128 STORE ROWS
// resident in LDS and streams the whole query batch past it, then takes the next row block.
//   * every row byte leaves HBM exactly once (nt), whatever the number of queries; what is re-read
//     per row block is the QUERY batch (n_queries x row bytes, < 1 MiB per 1024 queries), and that
//     stays in every XCD's L2 for the whole launch — no co-scheduling of workgroups needed for the
//     reuse (the ping-pong and row-streaming kernels re-read ROWS through L2, which only works while
//     the workgroups sharing them stay within microseconds of each other);
//   * the batch carries a second copy of its codes in MFMA fragment order (swizzle_queries_kernel:
//     per 32 queries and 128-byte K-block four 1 KiB pieces, lane (r, h) of piece x holding bytes
//     [64h + 16x, +16) of query r), so a wave's streamed operand is eight fully coalesced 1 KiB
//     loads per K-block straight into operand registers, one K-block ahead; no LDS, no transposition;
//   * a wave takes 64 queries at a time (2 x 4 accumulator tiles of 32 x 32 against the 128 resident
//     rows): per 32-byte k-step 4 ds_read_b128 + 8 MFMAs, half the LDS traffic per MFMA of the other
//     two kernels; the 8 waves run independently between the two barriers of a row-block change.
// Integer pre-filter, exact epilogue and wave-private candidate lists as in the ping-pong kernel; the
// per-query integer bounds come precomputed from qs_bounds_kernel.
__global__ __launch_bounds__(256) void swizzle_queries_kernel(const uint8_t *__restrict__ codes, uint32_t pitch,
                                                             uint32_t q_pad, uint32_t nkb, uint4 *__restrict__ out) {
    // out[((f * nkb + kb) * 4 + x) * 64 + lane] = bytes [128 kb + 64 h + 16 x, +16) of query 32 f + r
    const uint64_t total = (uint64_t)(q_pad / 32) * nkb * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t lane = (uint32_t)(i & 63u), x = (uint32_t)(i >> 6) & 3u;
        const uint64_t fk = i >> 8;
        const uint32_t kb = (uint32_t)(fk % nkb), f = (uint32_t)(fk / nkb);
        const uint32_t r = lane & 31u, h = lane >> 5, k = kb * 128u + 64u * h + 16u * x;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < pitch) v = *reinterpret_cast<const uint4 *>(codes + (uint64_t)(32u * f + r) * pitch + k);
        out[i] = v;
    }
}

template <bool LOW>
__global__ __launch_bounds__(256) void qs_bounds_kernel(const float *__restrict__ pivots, const float *__restrict__ q_offsets,
                                                       float multiplier, int largest, uint32_t q_pad, int *__restrict__ bq) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= q_pad) return;
    const float pv = pivots[q], qo = q_offsets[q];
    int b = pp_bound<LOW>(pv - qo, fabsf(pv) + fabsf(qo), multiplier, 1);
    if (__builtin_isinf(pv)) b = ((pv > 0.0f) == (largest != 0)) == LOW ? -(int)kPpLim : (int)kPpLim;
    bq[q] = b;
}

template <int MODE, bool LOW, int MJ>  // MJ: 32-row fragments resident (4: rows of up to 1152 B; 3: up to 1536 B)
__global__ __launch_bounds__(512) void u8_gemm_qs_kernel(const uint8_t *__restrict__ codes,
                                                        const float *__restrict__ v_offsets,
                                                        const uint4 *__restrict__ qfrag, const float *__restrict__ q_offsets,
                                                        const int *__restrict__ bq_all, float multiplier, uint32_t n_rows,
                                                        uint32_t n_queries, uint32_t q_pad, uint32_t ad,
                                                        float *__restrict__ out, uint64_t out_pitch, BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr int MI = 2, KB = 128, QS_ROWS = 32 * MJ;  // resident rows per workgroup
    // MODE 0: scores out; 1 / 2: filter for the largest / smallest; 3: the best score of every (query, row block)
    // out[q * out_pitch + block] (direction filt.largest) - the pivot sample of a large batch without its Q x S score matrix
    constexpr bool FILTER = MODE == 1 || MODE == 2;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t nkb = __builtin_amdgcn_readfirstlane((ad + KB - 1) / KB);
    const uint32_t PA = nkb * KB + 16;  // LDS pitch of a row
    const uint32_t n_blocks = (n_rows + QS_ROWS - 1) / QS_ROWS;
    const uint32_t n_chunks = q_pad / 64;  // 64-query chunks (q_pad is a multiple of 256: whole chunks, zero queries at the end)
    const uint32_t live_chunks = (n_queries + 63) / 64;
    float *voff_s = reinterpret_cast<float *>(lds_raw + (size_t)QS_ROWS * PA);  // [128]
    int *br_s = reinterpret_cast<int *>(voff_s + QS_ROWS);                       // [128]
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(br_s + QS_ROWS) + wave;
    int *bq_s = reinterpret_cast<int *>(br_s + QS_ROWS) + 16;                    // [64 * live_chunks] integer query bounds
    constexpr bool LARGEST = MODE == 1;
    if (FILTER && lane == 0) *wcount_s = 0;
    if (FILTER)
        for (uint32_t i = t; i < 64 * live_chunks; i += 512) bq_s[i] = bq_all[i];
    const float never = (MODE == 3 ? filt.largest != 0 : LARGEST) ? -__builtin_huge_valf() : __builtin_huge_valf();
    (void)n_chunks;

    // streamed operand: chunk c, K-block kb -> fragments 2c and 2c + 1, 4 KiB each, contiguous per fragment
    const uint4 *q_lane = qfrag + lane;
    v4i A0[MI][4], A1[MI][4];
    auto load_q = [&](v4i(&a)[MI][4], uint32_t c, uint32_t kb) {
#pragma unroll
        for (int i = 0; i < MI; i++) {
            const uint4 *p = q_lane + ((uint64_t)(2 * c + i) * nkb + kb) * 256;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const uint4 v = p[64 * x];
                a[i][x] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            }
        }
    };
    const uint8_t *b_base = lds_raw + r * PA + 64 * h;

    // Row-block fill: the block is 128 * ad contiguous bytes of the store; thread t takes the 16-byte
    // pieces t, t + 512, ... (at most 18), requested BEFORE the barrier that frees the LDS rows (a
    // wave that finishes its queries early has its share of the next block in flight while the others
    // compute) and written after it.  Bytes [ad, nkb * 128) of an LDS row are never written: the
    // query image is zero there, so whatever they hold adds nothing (integer arithmetic).
    const uint32_t per = ad / 16;                                  // pieces per row
    const uint32_t n_pieces = __builtin_amdgcn_readfirstlane((QS_ROWS * per + 511) / 512);  // per thread (the last one may fall past the block)
    const uint32_t p_row0 = (uint32_t)t / per, p_c0 = (uint32_t)t % per, d_row = 512 / per, d_c = 512 % per;
    constexpr int MAXP = 18;
    v4i st[MAXP];
    float vo_pf = 0.0f;  // v_offset of row t of the requested block (threads 0..127)
    // every element is (re)defined on every call (pieces past the count re-read the last one): a
    // conditional definition would keep the old value alive through the whole query loop (spills).
    // A thread's last piece may lie just past the block: in the store's row padding, never written.
    // (Requesting the pieces BEFORE the wave's last epilogue, to take the last wave's HBM round trip
    // out of the block change, was tried: accumulators + pieces + epilogue temporaries do not fit in
    // 256 registers, and the spills cost more than the round trip.)
    auto fill_request = [&](uint32_t blk) {
        const uint8_t *p = codes + (uint64_t)blk * QS_ROWS * ad + (size_t)t * 16;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint32_t ii = (uint32_t)i < n_pieces ? (uint32_t)i : n_pieces - 1;  // wave-uniform
            st[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p + (size_t)ii * 8192));
        }
        if (n_pieces > 12) {  // rows longer than 768 B
#pragma unroll
            for (int i = 12; i < MAXP; i++) {
                const uint32_t ii = (uint32_t)i < n_pieces ? (uint32_t)i : n_pieces - 1;
                st[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p + (size_t)ii * 8192));
            }
        } else {  // "defined" without an instruction: keeps the old values from staying alive (see above)
#pragma unroll
            for (int i = 12; i < MAXP; i++) asm volatile("" : "=v"(st[i]));
        }
        vo_pf = v_offsets[(uint64_t)blk * QS_ROWS + (t < QS_ROWS ? t : 0)];  // padded like codes[]
    };
    auto fill_write = [&]() {
        uint32_t row = p_row0, c = p_c0;
        asm volatile("" : "+v"(row), "+v"(c));  // the 18 LDS addresses are recomputed per block, not kept (and spilled)
#pragma unroll
        for (int i = 0; i < MAXP; i++) {
            if ((uint32_t)i < n_pieces && row < (uint32_t)QS_ROWS) *reinterpret_cast<v4i *>(lds_raw + row * PA + c * 16) = st[i];
            row += d_row;
            c += d_c;
            if (c >= per) {
                c -= per;
                row++;
            }
        }
    };
    const uint32_t my_first = wave;  // this wave's first chunk of every row block
    // developer timeline (libquantization_amd_dev.so only): cycles per phase, summed over the row blocks
    unsigned long long *stamps = QAMD_GEMM_STAMPS();
    const bool timed = stamps != nullptr;
    unsigned long long tm_prev = timed ? __builtin_amdgcn_s_memtime() : 0ull, tm_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tm_first = tm_prev, rt_first = timed ? __builtin_amdgcn_s_memrealtime() : 0ull;  // 100 MHz
    auto lap = [&](int slot) {
        if (timed) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tm_acc[slot] += now - tm_prev;
            tm_prev = now;
        }
    };
    fill_request(blockIdx.x < n_blocks ? blockIdx.x : 0u);
    if (my_first < live_chunks) load_q(A0, my_first, 0);

    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t row0 = (uint64_t)blk * QS_ROWS;
        const uint32_t next_blk = blk + gridDim.x < n_blocks ? blk + gridDim.x : blk;  // past the end: re-request (always defined)
        lap(5);  // (request issue, loop overhead)
        __syncthreads();  // every wave is done with the previous block's rows
        lap(0);  // waiting for the other waves
        fill_write();
        if (t < QS_ROWS) {
            const bool ok = row0 + t < n_rows;
            const float vo = vo_pf;  // requested with the rows
            voff_s[t] = ok ? vo : never;
            if (FILTER) br_s[t] = ok ? pp_bound<LOW>(-vo, fabsf(vo), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
        }
        __syncthreads();
        lap(1);  // own pieces landing + LDS writes + second barrier

        for (uint32_t c = wave; c < live_chunks; c += 8) {
            // the chunk whose first K-block is requested under this chunk's last MFMAs (of this row
            // block, or the first one of the next: the queries do not depend on the rows)
            const uint32_t c_next = c + 8 < live_chunks ? c + 8 : my_first;
            v16i acc[MI][MJ];
            int br[MJ];
#pragma unroll
            for (int jj = 0; jj < MJ; jj++) br[jj] = FILTER ? br_s[jj * 32 + r] : 0;
            if (!FILTER) {
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int jj = 0; jj < MJ; jj++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[i][jj][e] = 0;
            } else {
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const v4i bq4 = *reinterpret_cast<const v4i *>(bq_s + 64 * c + i * 32 + 8 * gq + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; e++)
#pragma unroll
                            for (int jj = 0; jj < MJ; jj++) acc[i][jj][4 * gq + e] = -(bq4[e] + br[jj]);
                    }
            }
            auto compute = [&](const v4i(&a)[MI][4], uint32_t kb) {
                const uint8_t *pb = b_base + kb * KB;
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    v4i bf[MJ];
#pragma unroll
                    for (int jj = 0; jj < MJ; jj++) bf[jj] = *reinterpret_cast<const v4i *>(pb + (uint32_t)jj * 32u * PA + 16 * x);
#pragma unroll
                    for (int i = 0; i < MI; i++)
#pragma unroll
                        for (int jj = 0; jj < MJ; jj++)
                            acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i][x], bf[jj], acc[i][jj], 0, 0, 0);
                }
            };
            // Two K-blocks per turn, each requested a K-block ahead (sched_barrier: the eight loads are
            // issued HERE; left alone the scheduler sinks each load to just before its first use and
            // the wave stalls on every L2 round trip).  The request after the chunk's last K-block is
            // the next chunk's first one.
            lap(2);  // accumulator set-up
            // The two waves of a SIMD take turns at instruction priority, chunk by chunk: left alone the
            // older wave (0..3) wins the MFMA arbitration every time, finishes its chunks a quarter
            // earlier and idles at the barrier while the other one runs alone.
            if (((c >> 3) + ((uint32_t)wave >> 2)) & 1u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
            uint32_t kb = 0;
            for (; kb + 1 < nkb; kb += 2) {
                load_q(A1, c, kb + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(A0, kb);
                __builtin_amdgcn_sched_barrier(0);
                if (kb + 2 < nkb) load_q(A0, c, kb + 2);
                else load_q(A0, c_next, 0);
                __builtin_amdgcn_sched_barrier(0);
                compute(A1, kb + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            const bool odd = kb < nkb;
            if (odd) {  // odd K-block count: the next chunk's first block arrives in A1 and is moved after the epilogue
                load_q(A1, c_next, 0);
                __builtin_amdgcn_sched_barrier(0);
                compute(A0, kb);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            lap(3);  // K loop
            // ---- epilogue: 64 queries x 128 rows of this wave
            uint32_t c_e = c, wave_e = (uint32_t)wave, lane_e = (uint32_t)lane;
            asm volatile("" : "+s"(c_e), "+s"(wave_e), "+v"(lane_e));
            if (MODE == 3) {  // best score per query over the block's rows (rows on the lanes of each half wave)
                const bool lg = filt.largest != 0;
                const uint32_t r3 = lane_e & 31u, h3 = lane_e >> 5;
                float vo3[MJ];
#pragma unroll
                for (int jj = 0; jj < MJ; jj++) vo3[jj] = voff_s[jj * 32 + r3];  // `never` for rows past the end
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const uint32_t q = 64 * c_e + i * 32 + 8 * gq + 4 * h3;
                        const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + q);
                        const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            float best = never;
#pragma unroll
                            for (int jj = 0; jj < MJ; jj++) {
                                const float sc = (multiplier * (float)acc[i][jj][4 * gq + e] + qo[e]) + vo3[jj];
                                best = lg ? fmaxf(best, sc) : fminf(best, sc);
                            }
#pragma unroll
                            for (int d = 16; d >= 1; d >>= 1) {
                                const float o = __shfl_xor(best, d);
                                best = lg ? fmaxf(best, o) : fminf(best, o);
                            }
                            if (r3 == 0 && q + e < n_queries) out[(uint64_t)(q + e) * out_pitch + blk] = best;
                        }
                    }
            } else {
            uint4 *wave_list = FILTER ? filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * 8 + wave_e) * filt.wave_cap : nullptr;
            const uint32_t r_e = lane_e & 31u, h_e = lane_e >> 5;
#pragma unroll
            for (int jj = 0; jj < MJ; jj++) {
                const uint64_t row = row0 + jj * 32 + r_e;
                const bool row_ok = row < n_rows;
                const float v_off = voff_s[jj * 32 + r_e];
                const int brj = br[jj];
#pragma unroll
                for (int i = 0; i < MI; i++) {
                    if (!FILTER) __builtin_amdgcn_sched_barrier(0);
                    if (FILTER) {
                        // "some accumulator of the tile may pass" = the smallest is negative (LOW) / the
                        // largest is not: v_min3 / v_max3 fold two values per instruction
                        int ext = acc[i][jj][0];
#pragma unroll
                        for (int e = 1; e < 15; e += 2)
                            ext = LOW ? min(min(ext, acc[i][jj][e]), acc[i][jj][e + 1])
                                      : max(max(ext, acc[i][jj][e]), acc[i][jj][e + 1]);
                        ext = LOW ? min(ext, acc[i][jj][15]) : max(ext, acc[i][jj][15]);
                        if (!__builtin_amdgcn_readfirstlane(__ballot(LOW ? ext < 0 : ext >= 0) != 0)) continue;
                    }
#pragma unroll
                    for (int gq = 0; gq < 4; gq++) {
                        const uint32_t q = 64 * c_e + i * 32 + 8 * gq + 4 * h_e;  // first of four consecutive queries
                        if (!FILTER) {
                            const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + q);
                            const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float sc = (multiplier * (float)acc[i][jj][4 * gq + e] + qo[e]) + v_off;
                                if (row_ok && q + e < n_queries) out[(uint64_t)(q + e) * out_pitch + row] = sc;
                            }
                        } else {
                            const int a0 = acc[i][jj][4 * gq], a1 = acc[i][jj][4 * gq + 1], a2 = acc[i][jj][4 * gq + 2],
                                      a3 = acc[i][jj][4 * gq + 3];
                            const bool may_pass = LOW ? ((a0 | a1 | a2 | a3) < 0) : ((a0 & a1 & a2 & a3) >= 0);
                            if (may_pass) {
                                const v4i bq4 = *reinterpret_cast<const v4i *>(bq_all + q);
                                const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + q);
                                const float4 pv4 = *reinterpret_cast<const float4 *>(filt.pivot_scores + q);
                                const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                                const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
                                const int av[4] = {a0, a1, a2, a3};
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int s_int = av[e] + bq4[e] + brj;  // the plain integer dot product
                                    const float sc = (multiplier * (float)s_int + qo[e]) + v_off;
                                    const float d = LARGEST ? sc - pv[e] : pv[e] - sc;
                                    if (d >= 0.0f) {
                                        const uint32_t pos = atomicAdd(wcount_s, 1u);
                                        if (pos < filt.wave_cap)
                                            wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row,
                                                                        filt.query_base + q + e, 0u);
                                    }
                                }
                            }
                        }
                    }
                }
            }
            }  // MODE != 3
            if (odd) {
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int x = 0; x < 4; x++) A0[i][x] = A1[i][x];
            }
            lap(4);  // epilogue
        }
        // this wave's share of the next row block: requested as soon as its own chunks are done (the
        // accumulators are dead, their registers hold the pieces until the barrier)
        fill_request(next_blk);
    }
    if (timed && lane == 0 && blockIdx.x < kStampBlocks) {
        unsigned long long *o = stamps + ((uint64_t)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 6; i++) o[i] = tm_acc[i];
        o[6] = __builtin_amdgcn_s_memtime() - tm_first;      // shader clocks spent in the kernel ...
        o[7] = __builtin_amdgcn_s_memrealtime() - rt_first;  // ... and 10 ns ticks: the clock the kernel ran at
        o[15] = 1;
    }
    if (FILTER && lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = *wcount_s;
}


// ------------------------------------------------------------------------------------------
// The query-streaming kernel on v_mfma_i32_16x16x64_i8 (round 3).  Same structure, arithmetic and results as
// u8_gemm_qs_kernel; what changes is the matrix instruction.  Under int8 MFMA load this part is clock-limited: a bare
// loop of 32x32x32 instructions runs at 1.84-1.89 GHz (3.6-3.8 POP/s), the same loop on 16x16x64 at 2.2-2.3 GHz
// (4.35-4.40 POP/s) - a fifth more work per second out of the same pipes (tools/mfma_peak.py).  A wave's tile is
// unchanged, 64 queries x 128 rows = 4 x 8 accumulator tiles of 16 x 16 (128 registers), and so is the operand
// traffic per multiply: per 64-byte k-step 4 query fragments (streamed from L2 in fragment order, two K-blocks in
// flight) and 8 row fragments (ds_read_b128) feed 32 MFMAs.
//   * fragment order of the batch copy (swizzle_queries16_kernel): per 16 queries, 128-byte K-block and 64-byte
//     k-step one 1 KiB piece, lane (i = lane % 16, g = lane / 16) holding bytes [64 s + 16 g, +16) of query i;
//   * the resident rows lie on a pitch of whole 256-byte LDS bank rows, the 16-byte chunks of a row XOR-swizzled
//     by (row & 15): lane (i, g) reads chunk 8 kb + 4 s + g of row 16 jt + i, and the 16 lanes of every ds_read_b128
//     group ({0-3, 12-15, 20-27}, ...: two values of g) land on 16 different slots;
//   * lane (i, g) ends up with queries 4 g .. 4 g + 3 of the tile against row i of the tile: four accumulators.
// JT = 8: 128 resident rows, rows of up to 1024 bytes (128 KiB of LDS); JT = 6: 96 rows of up to 1536 bytes (144 KiB).
__global__ __launch_bounds__(256) void swizzle_queries16_kernel(const uint8_t *__restrict__ codes, uint32_t pitch,
                                                               uint32_t q_pad, uint32_t nkb, uint4 *__restrict__ out) {
    // out[((t * nkb + kb) * 2 + s) * 64 + lane] = bytes [128 kb + 64 s + 16 g, +16) of query 16 t + i
    const uint64_t total = (uint64_t)(q_pad / 16) * nkb * 128;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
        const uint32_t lane = (uint32_t)(idx & 63u), sx = (uint32_t)(idx >> 6) & 1u;
        const uint64_t tk = idx >> 7;
        const uint32_t kb = (uint32_t)(tk % nkb), t16 = (uint32_t)(tk / nkb);
        const uint32_t i = lane & 15u, g = lane >> 4, k = kb * 128u + 64u * sx + 16u * g;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < pitch) v = *reinterpret_cast<const uint4 *>(codes + (uint64_t)(16u * t16 + i) * pitch + k);
        out[idx] = v;
    }
}

// IT: 16-query tiles per wave and chunk: 4 (64 queries), or 2 for small batches - 129 .. 256 queries are 5 .. 8 chunks of 32,
// one for each of the 8 waves, where chunks of 64 would leave half of them idle.
template <int MODE, bool LOW, int JT, int IT>
__global__ __launch_bounds__(512) void u8_gemm_qs16_kernel(const uint8_t *__restrict__ codes,
                                                          const float *__restrict__ v_offsets,
                                                          const uint4 *__restrict__ qfrag, const float *__restrict__ q_offsets,
                                                          const int *__restrict__ bq_all, float multiplier, uint32_t n_rows,
                                                          uint32_t n_queries, uint32_t q_pad, uint32_t ad,
                                                          float *__restrict__ out, uint64_t out_pitch, BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr int KB = 128, QS_ROWS = 16 * JT, JH = JT / 2, CQ = 16 * IT;  // CQ queries x 128 (96) rows per wave and chunk
    constexpr bool FILTER = MODE == 1 || MODE == 2;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t i16 = (uint32_t)lane & 15u, g4 = (uint32_t)lane >> 4;
    const uint32_t nkb = __builtin_amdgcn_readfirstlane((ad + KB - 1) / KB);
    const uint32_t per = ad / 16;                                              // 16-byte chunks per row
    const uint32_t PA = __builtin_amdgcn_readfirstlane(((per + 15) / 16) * 256);  // LDS pitch: whole 256-byte bank rows
    const uint32_t n_blocks = (n_rows + QS_ROWS - 1) / QS_ROWS;
    const uint32_t live_chunks = (n_queries + CQ - 1) / CQ;
    float *voff_s = reinterpret_cast<float *>(lds_raw + (size_t)QS_ROWS * PA);  // [128]
    int *br_s = reinterpret_cast<int *>(voff_s + QS_ROWS);                       // [128]
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(br_s + QS_ROWS) + wave;
    int *bq_s = reinterpret_cast<int *>(br_s + QS_ROWS) + 16;                    // [CQ * live_chunks] integer query bounds
    constexpr bool LARGEST = MODE == 1;
    if (FILTER && lane == 0) *wcount_s = 0;
    if (FILTER)
        for (uint32_t i = t; i < CQ * live_chunks; i += 512) bq_s[i] = bq_all[i];
    const float never = (MODE == 3 ? filt.largest != 0 : LARGEST) ? -__builtin_huge_valf() : __builtin_huge_valf();

    // streamed operand: chunk c, k-step j (64 bytes of K: K-block j / 2, half j % 2) -> one 1 KiB piece per 16-query tile
    // 4c .. 4c + 3.  Three buffers of one k-step each rotate: the step in use and the next two on their way (two k-steps
    // = one K-block of lead, as in u8_gemm_qs_kernel, in 48 registers instead of 64: the 128 accumulator registers and
    // two waves per SIMD leave no more).
    const uint32_t nsteps = 2 * nkb;
    v4i Q0[IT], Q1[IT], Q2[IT];
    auto load_step = [&](v4i(&a)[IT], uint32_t c, uint32_t j) {
        const uint4 *p = qfrag + ((uint64_t)(IT * c) * nkb * 2 + j) * 64 + lane;  // (tile IT c, K-block j / 2, half j % 2)
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const uint4 v = p[(uint64_t)it * nkb * 128];
            a[it] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        }
    };
    // resident operand: lane (i16, g4) reads chunk 8 kb + 4 sx + g4 of row 16 jt + i16; the chunk's place in the row
    // is its index XOR i16 (the low four bits: K-block parity, k-step and g4), whole 256-byte groups (kb >> 1) apart
    // (k-step j of the chunk = chunk 4 j + g4 of the row: place (4 (j & 3) + g4) ^ i16 of 256-byte group j >> 2; 4 (j & 3)
    // has no bit in common with g4, so the place is ((g4 ^ i16) ^ 4 (j & 3)): one XOR with a scalar per k-step)
    const uint32_t b_row = i16 * PA, b_gi = (g4 ^ i16) * 16u;

    // Row-block fill: thread t takes the 16-byte pieces t, t + 512, ... of the block's 128 * ad contiguous bytes
    // (requested before the barrier that frees the LDS rows, written after it, as in u8_gemm_qs_kernel); piece c of
    // row r goes to place c ^ (r & 15).  Places a row does not fill (past its last chunk) are only ever multiplied
    // with the zero bytes of the query image.
    const uint32_t n_pieces = __builtin_amdgcn_readfirstlane((QS_ROWS * per + 511) / 512);
    const uint32_t p_row0 = (uint32_t)t / per, p_c0 = (uint32_t)t % per, d_row = 512 / per, d_c = 512 % per;
    constexpr int MAXP = JT == 8 ? 16 : 18;  // 128 rows x 64 pieces / 512 threads; 96 x 96 / 512
    v4i st[MAXP];
    float vo_pf = 0.0f;
    auto fill_request = [&](uint32_t blk) {
        const uint8_t *p = codes + (uint64_t)blk * QS_ROWS * ad + (size_t)t * 16;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            const uint32_t ii = (uint32_t)i < n_pieces ? (uint32_t)i : n_pieces - 1;  // wave-uniform
            st[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p + (size_t)ii * 8192));
        }
        if (n_pieces > 12) {
#pragma unroll
            for (int i = 12; i < MAXP; i++) {
                const uint32_t ii = (uint32_t)i < n_pieces ? (uint32_t)i : n_pieces - 1;
                st[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(p + (size_t)ii * 8192));
            }
        } else {
#pragma unroll
            for (int i = 12; i < MAXP; i++) asm volatile("" : "=v"(st[i]));
        }
        vo_pf = v_offsets[(uint64_t)blk * QS_ROWS + (t < QS_ROWS ? t : 0)];  // padded like codes[]
    };
    auto fill_write = [&]() {
        uint32_t row = p_row0, c = p_c0;
        asm volatile("" : "+v"(row), "+v"(c));
#pragma unroll
        for (int i = 0; i < MAXP; i++) {
            if ((uint32_t)i < n_pieces && row < (uint32_t)QS_ROWS)
                *reinterpret_cast<v4i *>(lds_raw + row * PA + ((c ^ (row & 15u)) * 16u)) = st[i];
            row += d_row;
            c += d_c;
            if (c >= per) {
                c -= per;
                row++;
            }
        }
    };
    const uint32_t my_first = wave;
    unsigned long long *stamps = QAMD_GEMM_STAMPS();
    const bool timed = stamps != nullptr;
    unsigned long long tm_prev = timed ? __builtin_amdgcn_s_memtime() : 0ull, tm_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tm_first = tm_prev, rt_first = timed ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto lap = [&](int slot) {
        if (timed) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tm_acc[slot] += now - tm_prev;
            tm_prev = now;
        }
    };
    fill_request(blockIdx.x < n_blocks ? blockIdx.x : 0u);
    if (my_first < live_chunks) {
        load_step(Q0, my_first, 0);
        load_step(Q1, my_first, 1);
    }

    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t row0 = (uint64_t)blk * QS_ROWS;
        const uint32_t next_blk = blk + gridDim.x < n_blocks ? blk + gridDim.x : blk;
        lap(5);
        __syncthreads();  // every wave is done with the previous block's rows
        lap(0);
        fill_write();
        if (t < QS_ROWS) {
            const bool ok = row0 + t < n_rows;
            const float vo = vo_pf;
            voff_s[t] = ok ? vo : never;
            if (FILTER) br_s[t] = ok ? pp_bound<LOW>(-vo, fabsf(vo), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
        }
        __syncthreads();
        lap(1);

        for (uint32_t c = wave; c < live_chunks; c += 8) {
            const uint32_t c_next = c + 8 < live_chunks ? c + 8 : my_first;
            v4i acc[IT][JT];
            {
                int br[JT];  // (re-read in the epilogue's rare path: eight registers less across the K loop)
#pragma unroll
                for (int jt = 0; jt < JT; jt++) br[jt] = FILTER ? br_s[jt * 16 + i16] : 0;
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    v4i bq4 = {0, 0, 0, 0};
                    if (FILTER) bq4 = *reinterpret_cast<const v4i *>(bq_s + CQ * c + 16 * it + 4 * g4);
#pragma unroll
                    for (int jt = 0; jt < JT; jt++)
#pragma unroll
                        for (int e = 0; e < 4; e++) acc[it][jt][e] = FILTER ? -(bq4[e] + br[jt]) : 0;
                }
            }
            // k-step j with its streamed fragments in `a`: 8 row fragments (two halves of 4 tiles), 32 MFMAs.  (Reading the
            // row fragments half a k-step ahead in a pinned order - 4 reads, 16 MFMAs, 4 reads, 16 MFMAs - was measured:
            // no gain, 7 registers more; the other wave of the SIMD already covers the LDS round trip.)
            auto compute = [&](const v4i(&a)[IT], uint32_t j) {
                // this lane's address in tile 0; the tiles are 16 * PA apart (a scalar).  Opaque to the optimiser: left
                // alone it keeps all (k-step, tile) addresses in registers across the loop - and spills them
                uint32_t lane_addr = b_row + (b_gi ^ ((j & 3u) * 64u)) + (j >> 2) * 256u;
                asm volatile("" : "+v"(lane_addr));
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    v4i bf[JH];
#pragma unroll
                    for (int j4 = 0; j4 < JH; j4++)
                        bf[j4] = *reinterpret_cast<const v4i *>(lds_raw + lane_addr + (uint32_t)(JH * hf + j4) * 16u * PA);
#pragma unroll
                    for (int it = 0; it < IT; it++)
#pragma unroll
                        for (int j4 = 0; j4 < JH; j4++)
                            acc[it][JH * hf + j4] =
                                __builtin_amdgcn_mfma_i32_16x16x64_i8(a[it], bf[j4], acc[it][JH * hf + j4], 0, 0, 0);
                }
            };
            // k-step j >= nsteps of a chunk is k-step j - nsteps of the next one (its first two are requested under this
            // chunk's last MFMAs; the queries do not depend on the rows, so across row blocks as well)
            auto request = [&](v4i(&a)[IT], uint32_t j) {
                if (j < nsteps) load_step(a, c, j);
                else load_step(a, c_next, j - nsteps);
            };
            lap(2);
            // (the two waves of a SIMD take turns at priority chunk by chunk, as in u8_gemm_qs_kernel; without it the older
            // wave wins every arbitration and idles a quarter of the block at the barrier; flipping every turn of three
            // k-steps instead balances no better: both measured)
            if (((c >> 3) + ((uint32_t)wave >> 2)) & 1u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
            // at entry Q0 = k-step 0 and Q1 = k-step 1 are on their way; sched_barrier: the four loads are issued HERE
            uint32_t j = 0;
            for (; j + 2 < nsteps; j += 3) {
                request(Q2, j + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(Q0, j);
                __builtin_amdgcn_sched_barrier(0);
                request(Q0, j + 3);
                __builtin_amdgcn_sched_barrier(0);
                compute(Q1, j + 1);
                __builtin_amdgcn_sched_barrier(0);
                request(Q1, j + 4);
                __builtin_amdgcn_sched_barrier(0);
                compute(Q2, j + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint32_t left = nsteps - j;  // 0, 1 or 2 k-steps (nsteps is even: 2 when nsteps % 3 == 2, 1 when == 1)
            if (left >= 1) {
                request(Q2, j + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(Q0, j);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (left == 2) {
                request(Q0, j + 3);
                __builtin_amdgcn_sched_barrier(0);
                compute(Q1, j + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            lap(3);
            // ---- epilogue: lane (i16, g4) holds, per (it, jt), queries 64 c + 16 it + 4 g4 + e against row 16 jt + i16
            uint32_t c_e = c, wave_e = (uint32_t)wave, lane_e = (uint32_t)lane;
            asm volatile("" : "+s"(c_e), "+s"(wave_e), "+v"(lane_e));
            const uint32_t i_e = lane_e & 15u, g_e = lane_e >> 4;
            if (MODE == 3) {  // best score per query over the block's rows
                const bool lg = filt.largest != 0;
                float vo3[JT];
#pragma unroll
                for (int jt = 0; jt < JT; jt++) vo3[jt] = voff_s[jt * 16 + i_e];  // `never` for rows past the end
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const uint32_t q = CQ * c_e + 16 * it + 4 * g_e;
                    const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + q);
                    const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float best = never;
#pragma unroll
                        for (int jt = 0; jt < JT; jt++) {
                            const float sc = (multiplier * (float)acc[it][jt][e] + qo[e]) + vo3[jt];
                            best = lg ? fmaxf(best, sc) : fminf(best, sc);
                        }
#pragma unroll
                        for (int d = 8; d >= 1; d >>= 1) {
                            const float o = __shfl_xor(best, d);
                            best = lg ? fmaxf(best, o) : fminf(best, o);
                        }
                        if (i_e == 0 && q + e < n_queries) out[(uint64_t)(q + e) * out_pitch + blk] = best;
                    }
                }
            } else {
                uint4 *wave_list = FILTER ? filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * 8 + wave_e) * filt.wave_cap : nullptr;
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const uint32_t q = CQ * c_e + 16 * it + 4 * g_e;  // first of this lane's four consecutive queries
                    if (FILTER) {
                        // "some accumulator of these 8 tiles may pass" = the smallest is negative (LOW) / the largest is not
                        int ext = acc[it][0][0];
#pragma unroll
                        for (int jt = 0; jt < JT; jt++) {
                            ext = LOW ? min(min(ext, acc[it][jt][0]), acc[it][jt][1]) : max(max(ext, acc[it][jt][0]), acc[it][jt][1]);
                            ext = LOW ? min(min(ext, acc[it][jt][2]), acc[it][jt][3]) : max(max(ext, acc[it][jt][2]), acc[it][jt][3]);
                        }
                        if (!__builtin_amdgcn_readfirstlane(__ballot(LOW ? ext < 0 : ext >= 0) != 0)) continue;
                    } else {
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int jt = 0; jt < JT; jt++) {
                        __builtin_amdgcn_sched_barrier(0);  // one tile at a time: no hoisting of the next tiles' address arithmetic
                        const uint64_t row = row0 + jt * 16 + i_e;
                        const bool row_ok = row < n_rows;
                        if (!FILTER) {
                            const float v_off = voff_s[jt * 16 + i_e];
                            const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + q);
                            const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float sc = (multiplier * (float)acc[it][jt][e] + qo[e]) + v_off;
                                if (row_ok && q + e < n_queries) out[(uint64_t)(q + e) * out_pitch + row] = sc;
                            }
                        } else {
                            const int a0 = acc[it][jt][0], a1 = acc[it][jt][1], a2 = acc[it][jt][2], a3 = acc[it][jt][3];
                            const bool may_pass = LOW ? ((a0 | a1 | a2 | a3) < 0) : ((a0 & a1 & a2 & a3) >= 0);
                            if (may_pass) {
                                const float v_off = voff_s[jt * 16 + i_e];
                                const v4i bq4 = *reinterpret_cast<const v4i *>(bq_all + q);
                                const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + q);
                                const float4 pv4 = *reinterpret_cast<const float4 *>(filt.pivot_scores + q);
                                const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                                const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
                                const int av[4] = {a0, a1, a2, a3};
                                const int brj = br_s[jt * 16 + i_e];
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int s_int = av[e] + bq4[e] + brj;  // the plain integer dot product
                                    const float sc = (multiplier * (float)s_int + qo[e]) + v_off;
                                    const float d = LARGEST ? sc - pv[e] : pv[e] - sc;
                                    if (d >= 0.0f) {
                                        const uint32_t pos = atomicAdd(wcount_s, 1u);
                                        if (pos < filt.wave_cap)
                                            wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row,
                                                                        filt.query_base + q + e, 0u);
                                    }
                                }
                            }
                        }
                    }
                }
            }
            // the next chunk's k-steps 0 and 1 belong in Q0 and Q1: after one left-over step they sit in Q1, Q2; after two in Q2, Q0
            if (left == 1) {
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    Q0[it] = Q1[it];
                    Q1[it] = Q2[it];
                }
            } else if (left == 2) {
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const v4i k1 = Q0[it];
                    Q0[it] = Q2[it];
                    Q1[it] = k1;
                }
            }
            lap(4);
        }
        fill_request(next_blk);
    }
    if (timed && lane == 0 && blockIdx.x < kStampBlocks) {
        unsigned long long *o = stamps + ((uint64_t)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 6; i++) o[i] = tm_acc[i];
        o[6] = __builtin_amdgcn_s_memtime() - tm_first;
        o[7] = __builtin_amdgcn_s_memrealtime() - rt_first;
        o[15] = 1;
    }
    if (FILTER && lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = *wcount_s;
}


// ------------------------------------------------------------------------------------------
// Queries in registers, rows through a double-buffered LDS slab (round 3; v_mfma_i32_16x16x64_i8).  For batches of a few
// hundred queries the query-streaming kernel is bound by its synchronous row-block change (3.5 us per 128 rows against
// 2-4 us of MFMAs), and that change cannot be overlapped there: `vmcnt` retires in order, every wave waits for its
// streamed query fragments once per k-step, and such a wait also waits for an older row DMA.  Here the K loop has NO
// vector-memory operation: a wave keeps its 32 queries' fragments for ALL k-steps in registers (NSTEPS x 2 x 4: 96 at
// 768-byte rows - two 16 x 16 x 64 tiles need only 32 accumulator registers against 64 rows), loaded once per launch, so
// the next 64-row slab can be asked for by LDS-DMA at the START of a block and arrive under this block's MFMAs; the
// only wait is the one in front of the block-end barrier.  A launch serves 256 queries (8 waves x 32); a batch is cut
// into passes, each of which streams the store once at close to the HBM rate (64 rows x 768 B per CU and ~1.9 us of
// MFMAs).  Row layout, fragment order and epilogue as in u8_gemm_qs16_kernel; rows of 256 / 384 / 512 / 768 / 1024 bytes.
template <int MODE, bool LOW, int NSTEPS>
__global__ __launch_bounds__(512) void u8_gemm_qr16_kernel(const uint8_t *__restrict__ codes,
                                                          const float *__restrict__ v_offsets,
                                                          const uint4 *__restrict__ qfrag, const float *__restrict__ q_offsets,
                                                          const int *__restrict__ bq_all, float multiplier, uint32_t n_rows,
                                                          uint32_t n_queries, uint32_t ad,
                                                          float *__restrict__ out, uint64_t out_pitch, BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr int IT = 2, JT = 4, QR_ROWS = 16 * JT;  // 32 queries x 64 rows per wave and block
    constexpr bool FILTER = MODE == 1 || MODE == 2;
    constexpr bool LARGEST = MODE == 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t i16 = (uint32_t)lane & 15u, g4 = (uint32_t)lane >> 4;
    const uint32_t nkb = __builtin_amdgcn_readfirstlane(ad / 128);  // rows of whole 128 bytes, NSTEPS == 2 nkb
    const uint32_t per = ad / 16;                                    // chunks per row
    const uint32_t pp = ((per + 15) / 16) * 16, PA = pp * 16;        // places per row in LDS: whole 256-byte bank rows (384-byte rows: 512)
    const uint32_t SLAB = QR_ROWS * PA;
    const uint32_t n_blocks = (n_rows + QR_ROWS - 1) / QR_ROWS;
    float *voff_s = reinterpret_cast<float *>(lds_raw + 2 * (size_t)SLAB);  // [2][64]
    int *br_s = reinterpret_cast<int *>(voff_s + 2 * QR_ROWS);              // [2][64]
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(br_s + 2 * QR_ROWS) + wave;
    if (FILTER && lane == 0) *wcount_s = 0;
    const float never = (MODE == 3 ? filt.largest != 0 : LARGEST) ? -__builtin_huge_valf() : __builtin_huge_valf();
    const bool live = (uint32_t)wave * 32u < n_queries;  // this wave's 32 queries exist (wave-uniform)

    // row DMA: LDS position p (16-byte units) of a slab = row p / per, place p % per, filled with the row's chunk
    // place ^ (row & 15); instruction k covers positions [64 k, 64 k + 64), wave w issues k = w, w + 8, ... (per in all)
    const uint32_t d_pos0 = (uint32_t)wave * 64u + (uint32_t)lane;
    const uint32_t d_row0 = d_pos0 / pp, d_place0 = d_pos0 % pp;
    const uint32_t d_dr = __builtin_amdgcn_readfirstlane(512u / pp), d_dc = __builtin_amdgcn_readfirstlane(512u % pp);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_raw;
    float vo_pf = 0.0f;
    // One DMA instruction occupies the CU's address path for its 1 KiB and holds the issuing wave meanwhile (48 of them at
    // the start of a block cost every wave ~1300 cycles): the pieces are issued ONE AT A TIME between k-steps, under MFMAs.
    const uint8_t *d_src = codes;
    uint32_t d_dst = 0, d_row = 0, d_place = 0, d_i = 0;
    auto dma_begin = [&](uint32_t blk, uint32_t par) {
        d_src = codes + (uint64_t)blk * QR_ROWS * ad;
        d_dst = lds_base + par * SLAB + (uint32_t)wave * 1024u;
        d_row = d_row0, d_place = d_place0, d_i = 0;
        vo_pf = v_offsets[(uint64_t)blk * QR_ROWS + (t < QR_ROWS ? t : 0)];  // padded like codes[]
    };
    auto dma_piece = [&]() {  // piece d_i of this wave (pp <= 64 instructions of 1 KiB per slab: at most 8 per wave)
        if ((uint32_t)wave + 8u * d_i < pp) {
            // a place past the row's last chunk (pitch > row: 384-byte rows) is never multiplied with a non-zero query
            // byte; it is filled with the row's last chunk (any readable bytes would do)
            const uint32_t c = d_place ^ (d_row & 15u);
            glds16_nt_unordered(d_src, d_row * ad + (c < per ? c : per - 1) * 16u, __builtin_amdgcn_readfirstlane(d_dst + d_i * 8192u));
        }
        d_row += d_dr;
        d_place += d_dc;
        if (d_place >= pp) {
            d_place -= pp;
            d_row++;
        }
        d_i++;
    };
    auto dma_request = [&](uint32_t blk, uint32_t par) {  // all pieces at once (the first block; waves without queries)
        dma_begin(blk, par);
        for (int i = 0; i < 8; i++) dma_piece();
    };
    auto tables_write = [&](uint64_t row0, uint32_t par) {
        if (t < QR_ROWS) {
            const bool ok = row0 + t < n_rows;
            const float vo = vo_pf;
            voff_s[par * QR_ROWS + t] = ok ? vo : never;
            if (FILTER) br_s[par * QR_ROWS + t] = ok ? pp_bound<LOW>(-vo, fabsf(vo), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
        }
    };
    const uint32_t first_blk = blockIdx.x < n_blocks ? blockIdx.x : 0u;
    dma_request(first_blk, 0u);
    // this wave's queries: tiles 2 wave and 2 wave + 1, every k-step, for the whole launch
    v4i Qr[NSTEPS][IT];
    {
        const uint4 *p = qfrag + ((uint64_t)(IT * wave) * nkb * 2) * 64 + lane;
#pragma unroll
        for (int j = 0; j < NSTEPS; j++)
#pragma unroll
            for (int it = 0; it < IT; it++) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (live && (uint32_t)j < 2 * nkb) v = p[((uint64_t)it * nkb * 2 + j) * 64];
                Qr[j][it] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
            }
    }
    // per-query constants of this lane: queries 32 wave + 16 it + 4 g4 + e
    v4i bq4[IT];
    float qo[IT][4];
#pragma unroll
    for (int it = 0; it < IT; it++) {
        const uint32_t q = 32u * wave + 16u * it + 4u * g4;
        bq4[it] = v4i{0, 0, 0, 0};
        if (FILTER && live) bq4[it] = *reinterpret_cast<const v4i *>(bq_all + q);
        float4 q4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (live) q4 = *reinterpret_cast<const float4 *>(q_offsets + q);
        qo[it][0] = q4.x, qo[it][1] = q4.y, qo[it][2] = q4.z, qo[it][3] = q4.w;
    }
    const uint32_t b_row = i16 * PA, b_gi = (g4 ^ i16) * 16u;  // see u8_gemm_qs16_kernel
    unsigned long long *stamps = QAMD_GEMM_STAMPS();
    const bool timed = stamps != nullptr;
    unsigned long long tm_prev = timed ? __builtin_amdgcn_s_memtime() : 0ull, tm_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tm_first = tm_prev, rt_first = timed ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto lap = [&](int slot) {
        if (timed) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tm_acc[slot] += now - tm_prev;
            tm_prev = now;
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tables_write((uint64_t)first_blk * QR_ROWS, 0u);
    __syncthreads();
    uint32_t par = 0;
    uint4 *wave_list = FILTER ? filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * 8 + wave) * filt.wave_cap : nullptr;

    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const uint64_t row0 = (uint64_t)blk * QR_ROWS;
        const uint32_t next_blk = blk + gridDim.x < n_blocks ? blk + gridDim.x : blk;  // past the end: re-request (harmless)
        // the next slab arrives under this block's MFMAs: nothing below waits for vector memory
        if (live) dma_begin(next_blk, par ^ 1u);
        else dma_request(next_blk, par ^ 1u);
        lap(5);
        if (live) {
            const float *voff_cur = voff_s + par * QR_ROWS;
            const int *br_cur = br_s + par * QR_ROWS;
            v4i acc[IT][JT];
            {
                int br[JT];
#pragma unroll
                for (int jt = 0; jt < JT; jt++) br[jt] = FILTER ? br_cur[jt * 16 + i16] : 0;
#pragma unroll
                for (int it = 0; it < IT; it++)
#pragma unroll
                    for (int jt = 0; jt < JT; jt++)
#pragma unroll
                        for (int e = 0; e < 4; e++) acc[it][jt][e] = FILTER ? -(bq4[it][e] + br[jt]) : 0;
            }
            lap(2);
            // the two waves of a SIMD (w, w + 4) take turns at priority block by block (see u8_gemm_qs_kernel)
            if ((((blk - blockIdx.x) / gridDim.x) + ((uint32_t)wave >> 2)) & 1u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
            const uint32_t slab_addr = par * SLAB;
#pragma unroll
            for (int j = 0; j < NSTEPS; j++) {  // NSTEPS == 2 nkb exactly (fully unrolled: Qr[] must stay in registers)
                if (j < 8) dma_piece();
                uint32_t lane_addr = slab_addr + b_row + (b_gi ^ ((uint32_t)(j & 3) * 64u)) + (uint32_t)(j >> 2) * 256u;
                asm volatile("" : "+v"(lane_addr));
                v4i bf[JT];
#pragma unroll
                for (int jt = 0; jt < JT; jt++) bf[jt] = *reinterpret_cast<const v4i *>(lds_raw + lane_addr + (uint32_t)jt * 16u * PA);
#pragma unroll
                for (int it = 0; it < IT; it++)
#pragma unroll
                    for (int jt = 0; jt < JT; jt++)
                        acc[it][jt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qr[j][it], bf[jt], acc[it][jt], 0, 0, 0);
            }
            if (NSTEPS < 8) {  // short rows: fewer k-steps than pieces
#pragma unroll
                for (int i = NSTEPS; i < 8; i++) dma_piece();
            }
            __builtin_amdgcn_s_setprio(0);
            lap(3);
            // ---- epilogue: lane (i16, g4) holds, per (it, jt), queries 32 wave + 16 it + 4 g4 + e against row 16 jt + i16
            if (MODE == 3) {
                const bool lg = filt.largest != 0;
                float vo3[JT];
#pragma unroll
                for (int jt = 0; jt < JT; jt++) vo3[jt] = voff_cur[jt * 16 + i16];
#pragma unroll
                for (int it = 0; it < IT; it++) {
                    const uint32_t q = 32u * wave + 16u * it + 4u * g4;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        float best = never;
#pragma unroll
                        for (int jt = 0; jt < JT; jt++) {
                            const float sc = (multiplier * (float)acc[it][jt][e] + qo[it][e]) + vo3[jt];
                            best = lg ? fmaxf(best, sc) : fminf(best, sc);
                        }
#pragma unroll
                        for (int d = 8; d >= 1; d >>= 1) {
                            const float o = __shfl_xor(best, d);
                            best = lg ? fmaxf(best, o) : fminf(best, o);
                        }
                        if (i16 == 0 && q + e < n_queries) out[(uint64_t)(q + e) * out_pitch + blk] = best;
                    }
                }
            } else {
                bool any = true;
                if (FILTER) {  // "some accumulator of the 8 tiles may pass" = the smallest is negative (LOW) / the largest is not
                    int ext = acc[0][0][0];
#pragma unroll
                    for (int it = 0; it < IT; it++)
#pragma unroll
                        for (int jt = 0; jt < JT; jt++) {
                            ext = LOW ? min(min(ext, acc[it][jt][0]), acc[it][jt][1]) : max(max(ext, acc[it][jt][0]), acc[it][jt][1]);
                            ext = LOW ? min(min(ext, acc[it][jt][2]), acc[it][jt][3]) : max(max(ext, acc[it][jt][2]), acc[it][jt][3]);
                        }
                    any = __builtin_amdgcn_readfirstlane(__ballot(LOW ? ext < 0 : ext >= 0) != 0);
                }
                if (any) {
#pragma unroll
                    for (int it = 0; it < IT; it++) {
                        const uint32_t q = 32u * wave + 16u * it + 4u * g4;
#pragma unroll
                        for (int jt = 0; jt < JT; jt++) {
                            __builtin_amdgcn_sched_barrier(0);
                            const uint64_t row = row0 + jt * 16 + i16;
                            const bool row_ok = row < n_rows;
                            if (!FILTER) {
                                const float v_off = voff_cur[jt * 16 + i16];
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const float sc = (multiplier * (float)acc[it][jt][e] + qo[it][e]) + v_off;
                                    if (row_ok && q + e < n_queries) out[(uint64_t)(q + e) * out_pitch + row] = sc;
                                }
                            } else {
                                const int a0 = acc[it][jt][0], a1 = acc[it][jt][1], a2 = acc[it][jt][2], a3 = acc[it][jt][3];
                                const bool may_pass = LOW ? ((a0 | a1 | a2 | a3) < 0) : ((a0 & a1 & a2 & a3) >= 0);
                                if (may_pass) {
                                    const float v_off = voff_cur[jt * 16 + i16];
                                    const float4 pv4 = *reinterpret_cast<const float4 *>(filt.pivot_scores + q);
                                    const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
                                    const int av[4] = {a0, a1, a2, a3};
                                    const int brj = br_cur[jt * 16 + i16];
#pragma unroll
                                    for (int e = 0; e < 4; e++) {
                                        const int s_int = av[e] + bq4[it][e] + brj;  // the plain integer dot product
                                        const float sc = (multiplier * (float)s_int + qo[it][e]) + v_off;
                                        const float d = LARGEST ? sc - pv[e] : pv[e] - sc;
                                        if (d >= 0.0f) {
                                            const uint32_t pos = atomicAdd(wcount_s, 1u);
                                            if (pos < filt.wave_cap)
                                                wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row,
                                                                            filt.query_base + q + e, 0u);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            lap(4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next slab (and its row offsets) has landed
        tables_write((uint64_t)next_blk * QR_ROWS, par ^ 1u);
        lap(1);
        __syncthreads();  // ... and so has everybody's; every wave is done with this block's slab
        lap(0);
        par ^= 1u;
    }
    if (timed && lane == 0 && blockIdx.x < kStampBlocks) {
        unsigned long long *o = stamps + ((uint64_t)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 6; i++) o[i] = tm_acc[i];
        o[6] = __builtin_amdgcn_s_memtime() - tm_first;
        o[7] = __builtin_amdgcn_s_memrealtime() - rt_first;
        o[15] = 1;
    }
    if (FILTER && lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = *wcount_s;
}


// ------------------------------------------------------------------------------------------
// Queries RESIDENT, rows streamed, 16 x 16 x 64 (late round 4): the form that took the binary batches off their floor
// (bin_gemm_rs4_kernel), for u8.  The query-streaming kernel pays per 128-row block a synchronous block change and an
// epilogue during which its matrix pipes idle (K loop 72 % of a block at 1024 queries); here a workgroup keeps a GROUP of
// the batch's query tiles in LDS in fragment order (up to 12 tiles of 768-byte queries: 144 KiB) for the whole launch, and
// every wave streams its OWN rows - 32 per trip (two 16-row tiles), straight from memory into registers in fragment order
// (lane (i, g): bytes [64 j + 16 g, + 16) of row i for k-step j, exactly the B operand; no LDS hop, no re-layout), the
// next trip's rows requested a whole trip ahead into a second register set - and multiplies them with every tile of its
// group: per pair of query tiles and k-step two fragment reads (conflict-free: a tile x k-step is one lane-linear 1 KiB
// piece) feed four MFMAs.  No barrier after the set-up.
// A batch larger than one group's LDS is cut into G groups that run AT THE SAME TIME on different CUs of the same XCD:
// workgroup b sits on XCD b % 8 (round-robin placement - speed only), its slot b / 8 there gives group slot % G and row
// stream slot / G, and the G workgroups of a stream walk the same rows in the same order, so a row leaves HBM once and
// the other groups find it in that XCD's L2 (or the memory-side cache).  Arithmetic, bounds and candidate lists are
// u8_gemm_qr16_kernel's.
struct RqGeometry {
    uint32_t groups, pairs_lo /* tile pairs of every group */, pairs_extra /* the first so many groups take one more */,
        streams_per_xcd, n_tiles /* of the batch, even */;
};
constexpr uint32_t rq_tile_cap(uint32_t nsteps) { return ((160u * 1024u - 2048u) / (nsteps * 1024u + 64u)) & ~1u; }
template <int MODE, bool LOW, int NSTEPS>
__global__ __launch_bounds__(512) void u8_gemm_rq16_kernel(const uint8_t *__restrict__ codes, const float *__restrict__ v_offsets,
                                                          const uint4 *__restrict__ qfrag, const float *__restrict__ q_offsets,
                                                          const int *__restrict__ bq_all, float multiplier, uint32_t n_rows,
                                                          RqGeometry geo, BatchFilter filt) {
    static_assert(MODE == 1 || MODE == 2, "the filter pass of topk_batch");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr int RT = 2, QP = 2, AD = 64 * NSTEPS, WAVES = 8;
    constexpr bool LARGEST = MODE == 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t i16 = (uint32_t)lane & 15u, g4 = (uint32_t)lane >> 4;
    const uint32_t xcd_slot = blockIdx.x / 8u, grp = xcd_slot % geo.groups, stream_local = xcd_slot / geo.groups;
    if (stream_local >= geo.streams_per_xcd) return;  // (32 CUs per XCD do not divide by every group count)
    const uint32_t stream = (blockIdx.x % 8u) * geo.streams_per_xcd + stream_local, n_streams = 8u * geo.streams_per_xcd;
    // group g: tile pairs [g lo + min(g, extra), + lo + (g < extra))
    const uint32_t tile0 = 2u * (grp * geo.pairs_lo + min(grp, geo.pairs_extra));
    const uint32_t my_tiles = 2u * (geo.pairs_lo + (grp < geo.pairs_extra ? 1u : 0u)), max_tiles = 2u * (geo.pairs_lo + (geo.pairs_extra ? 1u : 0u));
    uint4 *img = reinterpret_cast<uint4 *>(lds_raw);                                  // [my_tiles][NSTEPS][64] x 16 B
    int *nbq_s = reinterpret_cast<int *>(lds_raw + (size_t)max_tiles * NSTEPS * 1024);  // [16 my_tiles]: MINUS the query bounds
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(nbq_s + 16 * max_tiles) + wave;
    if (lane == 0) *wcount_s = 0;
    {
        const uint4 *src = qfrag + (size_t)tile0 * NSTEPS * 64;
        for (uint32_t idx = t; idx < my_tiles * NSTEPS * 64; idx += 64 * WAVES) img[idx] = src[idx];
        for (uint32_t i = t; i < 16 * my_tiles; i += 64 * WAVES) nbq_s[i] = -bq_all[16 * tile0 + i];
    }
    __syncthreads();
    uint4 *wave_list = filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * WAVES + wave) * filt.wave_cap;
    if (my_tiles) {
        const uint32_t n_chunks = (n_rows + 16 * RT - 1) / (16 * RT), stride = n_streams * WAVES;
        const uint32_t first = stream * WAVES + wave;
        auto clamp_chunk = [&](uint32_t c) { return c < n_chunks ? c : n_chunks - 1u; };  // past the end: a harmless re-read
        // the rows of a trip, as B operands: lane (i, g) holds bytes [64 j + 16 g, + 16) of row 16 rt + i.  Plain loads: the other
        // groups of this stream want the lines from L2.
        auto load_rows = [&](v4i (&rows)[RT][NSTEPS], float (&vo)[RT], uint32_t chunk) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const uint64_t row = (uint64_t)chunk * (16 * RT) + rt * 16 + i16;  // (the codes are padded by a 256-row tile)
                const uint8_t *p = codes + row * AD + 16u * g4;
#pragma unroll
                for (int j = 0; j < NSTEPS; j++) {
                    const uint4 x = *reinterpret_cast<const uint4 *>(p + 64 * j);
                    rows[rt][j] = v4i{(int)x.x, (int)x.y, (int)x.z, (int)x.w};
                }
                vo[rt] = v_offsets[row];
            }
        };
        v4i rows_a[RT][NSTEPS], rows_b[RT][NSTEPS];
        float vo_a[RT], vo_b[RT];
        load_rows(rows_a, vo_a, clamp_chunk(first));
        load_rows(rows_b, vo_b, clamp_chunk(first + stride));
        auto trip = [&](uint32_t chunk, v4i (&rows)[RT][NSTEPS], float (&vo)[RT]) {
            const uint64_t row0 = (uint64_t)chunk * (16 * RT);
            float v_off[RT];
            int br[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                v_off[rt] = vo[rt];
                const bool ok = row0 + rt * 16 + i16 < n_rows;
                br[rt] = ok ? pp_bound<LOW>(-v_off[rt], fabsf(v_off[rt]), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
            }
            for (uint32_t qt = 0; qt < my_tiles; qt += QP) {
                v4i acc[QP][RT];
                const uint4 *a_base = img + (size_t)qt * NSTEPS * 64 + lane;
#pragma unroll
                for (int qp = 0; qp < QP; qp++) {
                    const v4i nb = *reinterpret_cast<const v4i *>(nbq_s + 16 * (qt + qp) + 4 * g4);  // -(bound of queries 4 g .. 4 g + 3)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
#pragma unroll
                        for (int e = 0; e < 4; e++) acc[qp][rt][e] = nb[e] - br[rt];
                }
#pragma unroll
                for (int j = 0; j < NSTEPS; j++) {
#pragma unroll
                    for (int qp = 0; qp < QP; qp++) {
                        const uint4 a = a_base[(qp * NSTEPS + j) * 64];
                        const v4i a4 = {(int)a.x, (int)a.y, (int)a.z, (int)a.w};
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) acc[qp][rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, rows[rt][j], acc[qp][rt], 0, 0, 0);
                    }
                }
                // ---- epilogue: lane (i, g) holds queries 16 (tile0 + qt + qp) + 4 g + e against row 16 rt + i of the trip;
                // "some accumulator may pass" = the smallest is negative (LOW) / the largest is not
                int ext = acc[0][0][0];
#pragma unroll
                for (int qp = 0; qp < QP; qp++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        ext = LOW ? min(min(ext, acc[qp][rt][0]), acc[qp][rt][1]) : max(max(ext, acc[qp][rt][0]), acc[qp][rt][1]);
                        ext = LOW ? min(min(ext, acc[qp][rt][2]), acc[qp][rt][3]) : max(max(ext, acc[qp][rt][2]), acc[qp][rt][3]);
                    }
                if (__builtin_amdgcn_readfirstlane(__ballot(LOW ? ext < 0 : ext >= 0) != 0)) {
#pragma unroll
                    for (int qp = 0; qp < QP; qp++)
#pragma unroll
                        for (int rt = 0; rt < RT; rt++) {
                            const int a0 = acc[qp][rt][0], a1 = acc[qp][rt][1], a2 = acc[qp][rt][2], a3 = acc[qp][rt][3];
                            const bool may_pass = LOW ? ((a0 | a1 | a2 | a3) < 0) : ((a0 & a1 & a2 & a3) >= 0);
                            if (may_pass) {
                                const uint32_t ql = 16u * (qt + qp) + 4u * g4, q = 16u * tile0 + ql;  // in the group / in the launch's batch
                                const uint64_t row = row0 + rt * 16 + i16;
                                const float4 pv4 = *reinterpret_cast<const float4 *>(filt.pivot_scores + q);
                                const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + q);
                                const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w}, qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                                const int av[4] = {a0, a1, a2, a3};
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    const int s_int = av[e] - nbq_s[ql + e] + br[rt];  // the plain integer dot product
                                    const float sc = (multiplier * (float)s_int + qo[e]) + v_off[rt];
                                    const float d = LARGEST ? sc - pv[e] : pv[e] - sc;
                                    if (d >= 0.0f) {
                                        const uint32_t pos = atomicAdd(wcount_s, 1u);
                                        if (pos < filt.wave_cap)
                                            wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row, filt.query_base + q + e, 0u);
                                    }
                                }
                            }
                        }
                }
            }
            load_rows(rows, vo, clamp_chunk(chunk + 2 * stride));  // this set is free: the trip after the next one, a whole trip ahead
        };
        for (uint32_t chunk = first; chunk < n_chunks; chunk += 2 * stride) {
            trip(chunk, rows_a, vo_a);
            if (chunk + stride >= n_chunks) break;
            trip(chunk + stride, rows_b, vo_b);
        }
    }
    if (lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * WAVES + wave] = *wcount_s;
}


// The same kernel with the loops the other way round, for rows of 768 bytes (12 k-steps), where u8_gemm_rq16_kernel's two
// register sets of rows (192 of 256 registers) leave the compiler nothing to pipeline the fragment reads with (it emits
// read, wait, two MFMAs, read, wait, ...: half the matrix pipe's time).  Here the K loop is OUTSIDE: a wave keeps the
// accumulators of ALL of its group's tiles (NT x 2 x 4 registers: 96 at 12 tiles) and takes its 32 rows one 64-byte k-step
// slice at a time - 8 registers, multiplied with every tile and then dead - out of a ring of D = 6 slices that is refilled
// half a trip ahead, across trips.  Per k-step NT fragment reads feed 2 NT MFMAs back to back; one bound test and one
// ballot per trip.  NT is a template parameter (8, 10, 12: the accumulators must be registers); a group with fewer tiles
// pads with zero tiles whose bounds never pass.
template <int MODE, bool LOW, int NSTEPS, int NT>
__global__ __launch_bounds__(512) void u8_gemm_rk16_kernel(const uint8_t *__restrict__ codes, const float *__restrict__ v_offsets,
                                                          const uint4 *__restrict__ qfrag, const float *__restrict__ q_offsets,
                                                          const int *__restrict__ bq_all, float multiplier, uint32_t n_rows,
                                                          RqGeometry geo, BatchFilter filt) {
    static_assert(MODE == 1 || MODE == 2, "the filter pass of topk_batch");
    // (measured, tools/experiments/rk_variants.sh: fragment reads 2 / 4 / 8 tiles ahead, slices 6 / 12 k-steps ahead, 8 / 12 waves
    // per workgroup at 8 tiles: the same time to the percent, or slower where 12 slices spill.  With the row loads taken out of
    // the K loop - wrong rows, timing only - a single group's K loop takes 45 % fewer cycles, and a ring twice as deep does not
    // get any of that back: at one group the loop runs at what the rows' arrival allows, 5.3 TB/s of 64-byte half lines)
    // (... and so do four row tiles per trip at 8 tiles - half the fragment reads per MFMA: 1024 queries 6.68 against 6.68 ms)
    constexpr int RT = 2, AD = 64 * NSTEPS, WAVES = 8, D = NSTEPS % 6 == 0 ? 6 : 4;
    static_assert(NSTEPS % D == 0 && NT % 2 == 0, "ring of k-step slices");
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr bool LARGEST = MODE == 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t i16 = (uint32_t)lane & 15u, g4 = (uint32_t)lane >> 4;
    const uint32_t xcd_slot = blockIdx.x / 8u, grp = xcd_slot % geo.groups, stream_local = xcd_slot / geo.groups;
    if (stream_local >= geo.streams_per_xcd) return;
    const uint32_t stream = (blockIdx.x % 8u) * geo.streams_per_xcd + stream_local, n_streams = 8u * geo.streams_per_xcd;
    const uint32_t tile0 = 2u * (grp * geo.pairs_lo + min(grp, geo.pairs_extra));
    const uint32_t my_tiles = 2u * (geo.pairs_lo + (grp < geo.pairs_extra ? 1u : 0u));  // <= NT
    uint4 *img = reinterpret_cast<uint4 *>(lds_raw);                           // [NSTEPS][NT][64] x 16 B: a k-step's fragments side by side
    int *nbq_s = reinterpret_cast<int *>(lds_raw + (size_t)NT * NSTEPS * 1024);  // [16 NT]: MINUS the query bounds
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(nbq_s + 16 * NT) + wave;
    if (lane == 0) *wcount_s = 0;
    {
        const uint4 *src = qfrag + (size_t)tile0 * NSTEPS * 64;
        for (uint32_t idx = t; idx < (uint32_t)NT * NSTEPS * 64; idx += 64 * WAVES) {  // idx = (j, q, lane) <- the batch copy's (q, j, lane)
            const uint32_t l = idx & 63u, q = (idx >> 6) % NT, j = (idx >> 6) / NT;
            img[idx] = q < my_tiles ? src[((size_t)q * NSTEPS + j) * 64 + l] : make_uint4(0, 0, 0, 0);
        }
        for (uint32_t i = t; i < 16u * NT; i += 64 * WAVES)  // a padding tile: accumulators that can never pass (|bound of a row| <= 2^29)
            nbq_s[i] = i < 16 * my_tiles ? -bq_all[16 * tile0 + i] : (LOW ? (1 << 30) : -(1 << 30));
    }
    __syncthreads();
    uint4 *wave_list = filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * WAVES + wave) * filt.wave_cap;
    const uint32_t n_chunks = (n_rows + 16 * RT - 1) / (16 * RT), stride = n_streams * WAVES;
    const uint32_t first = stream * WAVES + wave;
    unsigned long long *stamps = QAMD_GEMM_STAMPS();  // (developer build: phase times in shader cycles, tools/gemm_timeline.py)
    const bool timed = stamps != nullptr;
    unsigned long long tm_prev = timed ? __builtin_amdgcn_s_memtime() : 0ull, tm_acc[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long tm_first = tm_prev, rt_first = timed ? __builtin_amdgcn_s_memrealtime() : 0ull;
    auto lap = [&](int slot) {
        if (timed) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            tm_acc[slot] += now - tm_prev;
            tm_prev = now;
        }
    };
    if (first < n_chunks) {
        auto clamp_chunk = [&](uint32_t c) { return c < n_chunks ? c : n_chunks - 1u; };  // past the end: a harmless re-read
        // slice j of a chunk, as B operands: lane (i, g) holds bytes [64 j + 16 g, + 16) of row 16 rt + i (plain loads: the
        // stream's other groups want the lines from L2)
        auto load_slice = [&](v4i (&w)[RT], uint32_t chunk, int j) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const uint64_t row = (uint64_t)chunk * (16 * RT) + rt * 16 + i16;  // (the codes are padded by a 256-row tile)
                // (plain, not nontemporal, also for a single group: a load takes 64 bytes of a row's 128-byte line and the next k-step
                // the other half - nt re-fetches it: 192 queries 1.57 -> 1.70 ms)
                const uint4 x = *reinterpret_cast<const uint4 *>(codes + row * AD + 64 * j + 16u * g4);
                w[rt] = v4i{(int)x.x, (int)x.y, (int)x.z, (int)x.w};
            }
        };
        v4i win[D][RT];
        float vo_cur[RT], vo_nxt[RT];
#pragma unroll
        for (int d = 0; d < D; d++) load_slice(win[d], first, d);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) vo_cur[rt] = v_offsets[(uint64_t)first * (16 * RT) + rt * 16 + i16];
        for (uint32_t chunk = first; chunk < n_chunks; chunk += stride) {
            const uint32_t next = clamp_chunk(chunk + stride);
            const uint64_t row0 = (uint64_t)chunk * (16 * RT);
            float v_off[RT];
            int br[RT];
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                v_off[rt] = vo_cur[rt];
                const bool ok = row0 + rt * 16 + i16 < n_rows;
                br[rt] = ok ? pp_bound<LOW>(-v_off[rt], fabsf(v_off[rt]), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
                vo_nxt[rt] = v_offsets[(uint64_t)next * (16 * RT) + rt * 16 + i16];
            }
            lap(5);
            v4i acc[NT][RT];
#pragma unroll
            for (int q = 0; q < NT; q++) {
                const v4i nb = *reinterpret_cast<const v4i *>(nbq_s + 16 * q + 4 * g4);  // -(bound of queries 4 g .. 4 g + 3 of tile q)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[q][rt][e] = nb[e] - br[rt];
            }
            lap(2);
            // The fragment reads run PF tiles ahead of the MFMAs that use them, across k-steps, in a pinned order (left alone the
            // compiler emits read, read, wait, two MFMAs, wait, two MFMAs: the LDS round trip lies open in front of every pair):
            // unit (j, q) = the read of unit + PF, then the two MFMAs of tile q at k-step j.
            constexpr int PF = 4, R = PF + 1, UNITS = NSTEPS * NT;
            v4i a[R];
            // (opaque: left alone the compiler keeps an address register per (k-step, tile) across the loop - and spills)
            uint32_t a_addr = (uint32_t)lane * 16u;
            asm volatile("" : "+v"(a_addr));
            auto read_unit = [&](int u) {  // fragment of tile u % NT at k-step u / NT: img is [k-step][tile][lane], so unit u is piece u
                const uint4 x = *reinterpret_cast<const uint4 *>(lds_raw + a_addr + (uint32_t)(u % 48) * 1024u);
                a[u % R] = v4i{(int)x.x, (int)x.y, (int)x.z, (int)x.w};
            };
#pragma unroll
            for (int u = 0; u < PF; u++) read_unit(u);
#pragma unroll
            for (int u = 0; u < UNITS; u++) {
                const int j = u / NT, q = u % NT;
                if ((u + PF) % 48 == 0 && u + PF < UNITS) {  // the immediate offset of a read reaches 64 KiB: a new base every 48 pieces
                    a_addr += 48u * 1024u;
                    asm volatile("" : "+v"(a_addr));
                }
                if (u + PF < UNITS) read_unit(u + PF);
#pragma unroll
                for (int rt = 0; rt < RT; rt++) acc[q][rt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u % R], win[j % D][rt], acc[q][rt], 0, 0, 0);
                if (q == NT - 1)  // the slice's slot is free: slice j + D - of this chunk, or of the wave's next one
                    load_slice(win[j % D], j + D < NSTEPS ? chunk : next, (j + D) % NSTEPS);
                __builtin_amdgcn_sched_barrier(0);
            }
            lap(3);
            // ---- epilogue: lane (i, g) holds queries 16 (tile0 + q) + 4 g + e against row 16 rt + i of the chunk
            int ext = acc[0][0][0];
#pragma unroll
            for (int q = 0; q < NT; q++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    ext = LOW ? min(min(ext, acc[q][rt][0]), acc[q][rt][1]) : max(max(ext, acc[q][rt][0]), acc[q][rt][1]);
                    ext = LOW ? min(min(ext, acc[q][rt][2]), acc[q][rt][3]) : max(max(ext, acc[q][rt][2]), acc[q][rt][3]);
                }
            if (__builtin_amdgcn_readfirstlane(__ballot(LOW ? ext < 0 : ext >= 0) != 0)) {
                uint32_t g_e = g4;  // (opaque: no per-tile index registers kept across the K loop)
                asm volatile("" : "+v"(g_e));
#pragma unroll
                for (int q = 0; q < NT; q++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const int a0 = acc[q][rt][0], a1 = acc[q][rt][1], a2 = acc[q][rt][2], a3 = acc[q][rt][3];
                        const bool may_pass = LOW ? ((a0 | a1 | a2 | a3) < 0) : ((a0 & a1 & a2 & a3) >= 0);
                        if (may_pass) {
                            const uint32_t ql = 16u * q + 4u * g_e, qq = 16u * tile0 + ql;  // in the group / in the launch's batch
                            const uint64_t row = row0 + rt * 16 + i16;
                            const float4 pv4 = *reinterpret_cast<const float4 *>(filt.pivot_scores + qq);
                            const float4 qo4 = *reinterpret_cast<const float4 *>(q_offsets + qq);
                            const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w}, qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                            const int av[4] = {a0, a1, a2, a3};
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const int s_int = av[e] - nbq_s[ql + e] + br[rt];  // the plain integer dot product
                                const float sc = (multiplier * (float)s_int + qo[e]) + v_off[rt];
                                const float d = LARGEST ? sc - pv[e] : pv[e] - sc;
                                if (d >= 0.0f) {
                                    const uint32_t pos = atomicAdd(wcount_s, 1u);
                                    if (pos < filt.wave_cap)
                                        wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row, filt.query_base + qq + e, 0u);
                                }
                            }
                        }
                    }
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++) vo_cur[rt] = vo_nxt[rt];
            lap(4);
        }
    }
    if (timed && lane == 0 && blockIdx.x < kStampBlocks) {
        unsigned long long *o = stamps + ((uint64_t)blockIdx.x * 8 + wave) * 16;
        for (int i = 0; i < 6; i++) o[i] = tm_acc[i];
        o[6] = __builtin_amdgcn_s_memtime() - tm_first;
        o[7] = __builtin_amdgcn_s_memrealtime() - rt_first;
        o[15] = 1;
    }
    if (lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * WAVES + wave] = *wcount_s;
}

}  // namespace

struct qamd_u8_query_batch {
    int device = 0;
    uint64_t actual_dim = 0;
    uint64_t n_queries = 0;
    uint64_t q_pad = 0;  // round_up(n_queries, 256)
    uint64_t pitch = 0;  // round_up(actual_dim, 64): whole 64-byte K-tiles, zero padded
    DevBuf codes;        // [q_pad][pitch], zero rows past n_queries
    DevBuf offsets;      // [q_pad] f32
    DevBuf frag;         // the codes again in MFMA fragment order (swizzle_queries_kernel), for u8_gemm_qs_kernel
    uint32_t frag_nkb = 0;  // 128-byte K-blocks per query in `frag`
    bool frag16 = false;    // `frag` is in the order of u8_gemm_qs16_kernel (16-query tiles), else of u8_gemm_qs_kernel
};

namespace {

qamd_status check_batch(const qamd_u8 *h, const qamd_u8_query_batch *b) {
    if (!h || !b) return fail(QAMD_ERR_ARGUMENTS, "null handle or query batch");
    if (b->actual_dim != h->meta.actual_dim)
        return fail(QAMD_ERR_ARGUMENTS, "queries have %llu codes, store rows have %llu",
                    (unsigned long long)b->actual_dim, (unsigned long long)h->meta.actual_dim);
    return QAMD_OK;
}

// Launch the GEMM over rows [0, n_rows) of (codes, offsets) for every query of the batch.
template <int MODE, int TQ_, int TR_, int WQ, int WR, int BK_>
qamd_status launch_gemm_cfg(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                            const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                            const BatchFilter &filt, hipStream_t s) {
    const uint32_t q_tiles = (uint32_t)((b->n_queries + TQ_ - 1) / TQ_);  // q_pad (multiple of 256) covers them
    const uint64_t r_tiles = round_up((n_rows + TR_ - 1) / TR_, 8);  // whole groups of 8 row tiles (one per XCD)
    const uint64_t blocks = r_tiles * q_tiles;
    if (blocks > 0x7FFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "batch too large for one launch");
    constexpr size_t lds_bytes = (size_t)2 * (TQ_ + TR_) * (BK_ + 16);
    QAMD_LDS_OPT_IN((&u8_gemm_kernel<MODE, TQ_, TR_, WQ, WR, BK_>), (int)lds_bytes);
    hipLaunchKernelGGL((u8_gemm_kernel<MODE, TQ_, TR_, WQ, WR, BK_>), dim3((unsigned)blocks), dim3(64 * WQ * WR), lds_bytes,
                       s, codes, v_offsets, b->codes.as<uint8_t>(), (uint32_t)b->pitch, b->offsets.as<float>(), h->meta.multiplier,
                       (uint32_t)n_rows, (uint32_t)b->n_queries, (uint32_t)h->meta.actual_dim, q_tiles, out, out_pitch,
                       filt);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

inline uint32_t pp_launches(uint64_t n_queries) {
    const uint64_t per = (uint64_t)std::max(1, device_info().cu_count / 8) * (n_queries <= 128 ? 128 : 256);
    return (uint32_t)((n_queries + per - 1) / per);
}

// Ping-pong kernel launch: one persistent workgroup per CU; at most 32 query tiles per launch
// (8192 queries), larger batches go in slices of 8192.
template <int MODE, bool LOW, int MI, int MJ>
qamd_status launch_gemm_pp_cfg(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                           const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                           const BatchFilter &filt, hipStream_t s) {
    QAMD_LDS_OPT_IN((&u8_gemm_pp_kernel<MODE, LOW, MI, MJ>), (int)(PpShape<MI, MJ>::LDS));
    constexpr uint64_t TQW = 64 * MI;  // queries per workgroup tile
    constexpr size_t lds_bytes = PpShape<MI, MJ>::LDS;
    const uint32_t cus_per_xcd = (uint32_t)std::max(1, device_info().cu_count / 8);
    const uint64_t all_q_tiles = (b->n_queries + TQW - 1) / TQW;
    for (uint64_t qt0 = 0; qt0 < all_q_tiles; qt0 += cus_per_xcd) {
        const uint32_t q_tiles = (uint32_t)std::min<uint64_t>(cus_per_xcd, all_q_tiles - qt0);
        const uint32_t row_lanes = cus_per_xcd / q_tiles;
        const uint64_t q_base = qt0 * TQW;
        BatchFilter f = filt;
        if (MODE != 0) {
            f.pivot_scores += q_base;
            f.query_base = (uint32_t)q_base;
            f.wave_base = (uint32_t)(qt0 / cus_per_xcd) * pp_waves_per_launch();
        }
        hipLaunchKernelGGL((u8_gemm_pp_kernel<MODE, LOW, MI, MJ>), dim3(8 * row_lanes * q_tiles), dim3(512), lds_bytes, s,
                           codes, v_offsets, b->codes.as<uint8_t>() + q_base * b->pitch, (uint32_t)b->pitch,
                           b->offsets.as<float>() + q_base, h->meta.multiplier, (uint32_t)n_rows,
                           (uint32_t)(b->n_queries - q_base), (uint32_t)h->meta.actual_dim, q_tiles, row_lanes,
                           MODE == 0 ? out + q_base * out_pitch : out, out_pitch, f);
        QAMD_HIP(hipGetLastError());
    }
    return QAMD_OK;
}

template <int MODE>
qamd_status launch_gemm_pp(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                           const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                           const BatchFilter &filt, hipStream_t s) {
    const bool small = b->n_queries <= 128;  // 128-query tile: the store is streamed once, HBM-bound
    if (MODE == 0)
        return small ? launch_gemm_pp_cfg<0, false, 2, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s)
                     : launch_gemm_pp_cfg<0, false, 4, 2>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
    // "may pass" is s <= bound when the score falls with s (multiplier < 0) xor smallest-first
    constexpr int M = MODE == 0 ? 1 : MODE;
    const bool low = (h->meta.multiplier < 0.0f) != (MODE == 2);
    if (small)
        return low ? launch_gemm_pp_cfg<M, true, 2, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s)
                   : launch_gemm_pp_cfg<M, false, 2, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
    return low ? launch_gemm_pp_cfg<M, true, 4, 2>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s)
               : launch_gemm_pp_cfg<M, false, 4, 2>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
}

// Row-streaming kernel launch: one persistent workgroup per CU, query tile resident in LDS.
// rs_frags() = query fragments per workgroup (0: the tile does not fit, use another kernel).
inline int rs_frags(uint64_t n_queries, uint64_t ad) {
    const size_t lds_max = 160 * 1024;
    for (int mi : {1, 2, 4})  // the smallest tile that holds the whole batch, else the largest that fits
        if (n_queries <= (uint64_t)32 * mi && (size_t)32 * mi * (round_up(ad, 128) + 16) + 2048 + 32768 <= lds_max) return mi;
    for (int mi : {4, 2, 1})
        if ((size_t)32 * mi * (round_up(ad, 128) + 16) + 2048 + 32768 <= lds_max) return mi;
    return 0;
}

template <int MODE, bool LOW, int MI, bool NT>
qamd_status launch_gemm_rs_cfg(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                               const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                               const BatchFilter &filt, hipStream_t s) {
    QAMD_LDS_OPT_IN((&u8_gemm_rs_kernel<MODE, LOW, MI, NT>), 160 * 1024);
    constexpr uint64_t TQW = 32 * MI;
    const size_t lds_bytes = RsShape<MI>::lds_bytes((uint32_t)h->meta.actual_dim);
    const uint32_t cus_per_xcd = (uint32_t)std::max(1, device_info().cu_count / 8);
    const uint64_t all_q_tiles = (b->n_queries + TQW - 1) / TQW;
    for (uint64_t qt0 = 0; qt0 < all_q_tiles; qt0 += cus_per_xcd) {
        const uint32_t q_tiles = (uint32_t)std::min<uint64_t>(cus_per_xcd, all_q_tiles - qt0);
        const uint32_t row_lanes = cus_per_xcd / q_tiles;
        const uint64_t q_base = qt0 * TQW;
        BatchFilter f = filt;
        if (MODE != 0) {
            f.pivot_scores += q_base;
            f.query_base = (uint32_t)q_base;
            f.wave_base = (uint32_t)(qt0 / cus_per_xcd) * pp_waves_per_launch();
        }
        hipLaunchKernelGGL((u8_gemm_rs_kernel<MODE, LOW, MI, NT>), dim3(8 * row_lanes * q_tiles), dim3(512), lds_bytes, s,
                           codes, v_offsets, b->codes.as<uint8_t>() + q_base * b->pitch, (uint32_t)b->pitch,
                           b->offsets.as<float>() + q_base, h->meta.multiplier, (uint32_t)n_rows,
                           (uint32_t)(b->n_queries - q_base), (uint32_t)h->meta.actual_dim, q_tiles, row_lanes,
                           MODE == 0 ? out + q_base * out_pitch : out, out_pitch, f);
        QAMD_HIP(hipGetLastError());
    }
    return QAMD_OK;
}

template <int MODE>
qamd_status launch_gemm_rs(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                           const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                           const BatchFilter &filt, hipStream_t s) {
    const int mi = rs_frags(b->n_queries, h->meta.actual_dim);
    constexpr int M = MODE == 0 ? 1 : MODE;
    const bool low = MODE != 0 && (h->meta.multiplier < 0.0f) != (MODE == 2);
    const bool nt = b->n_queries <= (uint64_t)32 * mi;  // one query tile: every row byte is read exactly once
#define QAMD_RS2(MI_, NT_)                                                                                        \
    (MODE == 0 ? launch_gemm_rs_cfg<0, false, MI_, NT_>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s)  \
     : low     ? launch_gemm_rs_cfg<M, true, MI_, NT_>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s)   \
               : launch_gemm_rs_cfg<M, false, MI_, NT_>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s))
#define QAMD_RS(MI_) (nt ? QAMD_RS2(MI_, true) : QAMD_RS2(MI_, false))
    if (mi == 4) return QAMD_RS(4);
    if (mi == 2) return QAMD_RS(2);
    return QAMD_RS(1);
#undef QAMD_RS2
#undef QAMD_RS
}

// Queries per launch slice of the kernel that serves this batch (wave-list bookkeeping).
bool qr_selected(const qamd_u8 *h, const qamd_u8_query_batch *b, bool filter_mode);
bool rq_selected(const qamd_u8 *h, const qamd_u8_query_batch *b);
inline uint32_t gemm_launches(const qamd_u8 *h, const qamd_u8_query_batch *b, bool rs, bool qs) {
    if (qs && qr_selected(h, b, true)) return (uint32_t)((b->n_queries + 255) / 256);  // passes of the queries-in-registers form
    if (qs) return (uint32_t)((b->n_queries + 2048 - 1) / 2048);
    if (!rs) return pp_launches(b->n_queries);
    const uint64_t per = (uint64_t)std::max(1, device_info().cu_count / 8) * 32 * rs_frags(b->n_queries, h->meta.actual_dim);
    return (uint32_t)((b->n_queries + per - 1) / per);
}

// Which kernel serves a batch: the ping-pong kernel (rows of at least three 64-byte K-tiles, a
// usable multiplier for its integer pre-filter), else u8_gemm_kernel.
bool pp_selected(const qamd_u8 *h, const qamd_u8_query_batch *b, bool filter_mode) {
    static const char *cfg = dev_env("QAMD_GEMM_CFG");  // developer A/B switch: 0/3/4/5 = u8_gemm_kernel shapes
    if (cfg && cfg[0] != 'p') return false;
    const float m = h->meta.multiplier;
    return h->meta.actual_dim > 128 && h->meta.actual_dim <= 32768 && (!filter_mode || (std::isfinite(m) && m != 0.0f));
}

// Query-streaming kernel launch: one persistent workgroup per CU, slices of kQsSlice queries (the
// slice's fragment-order codes, 1.5 MiB at 768-byte rows, stay in every XCD's L2).
constexpr uint64_t kQsSlice = 2048;
template <int MODE, bool LOW, int MJ>
qamd_status launch_gemm_qs_cfg(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                               const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                               const BatchFilter &filt, const int *bq, hipStream_t s) {
    QAMD_LDS_OPT_IN((&u8_gemm_qs_kernel<MODE, LOW, MJ>), 160 * 1024);
    const uint32_t nkb = b->frag_nkb;
    constexpr int QS_ROWS = 32 * MJ;
    const size_t lds_bytes = (size_t)QS_ROWS * (nkb * 128 + 16) + 2 * QS_ROWS * 4 + 64 + kQsSlice * 4;
    const uint32_t grid = (uint32_t)std::max(1, device_info().cu_count / 8) * 8;
    for (uint64_t q_base = 0; q_base < b->n_queries; q_base += kQsSlice) {
        const uint32_t nq = (uint32_t)std::min<uint64_t>(kQsSlice, b->n_queries - q_base);
        BatchFilter f = filt;
        if (MODE == 1 || MODE == 2) {
            f.pivot_scores += q_base;
            f.query_base = (uint32_t)q_base;
            f.wave_base = (uint32_t)(q_base / kQsSlice) * pp_waves_per_launch();
        }
        hipLaunchKernelGGL((u8_gemm_qs_kernel<MODE, LOW, MJ>), dim3(grid), dim3(512), lds_bytes, s, codes, v_offsets,
                           b->frag.as<uint4>() + (q_base / 32) * nkb * 256, b->offsets.as<float>() + q_base,
                           (MODE == 1 || MODE == 2) ? bq + q_base : nullptr, h->meta.multiplier, (uint32_t)n_rows, nq,
                           (uint32_t)round_up((uint64_t)nq, 64), (uint32_t)h->meta.actual_dim,
                           (MODE == 0 || MODE == 3) ? out + q_base * out_pitch : out, out_pitch, f);
        QAMD_HIP(hipGetLastError());
    }
    return QAMD_OK;
}

// The same on v_mfma_i32_16x16x64_i8 (u8_gemm_qs16_kernel): rows of up to 1024 bytes, the batch's fragment copy in its order.
constexpr uint64_t kQs16SmallBatch = 256;
inline bool qs16_wanted(uint32_t nkb) {
    static const char *e = dev_env("QAMD_QS16");  // developer A/B: 0 = the 32x32x32 kernel for every row length
    return nkb >= 1 && nkb <= 12 && !(e && e[0] == '0');
}

template <int MODE, bool LOW, int JT, int IT>
qamd_status launch_gemm_qs16_cfg(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                                 const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                                 const BatchFilter &filt, const int *bq, hipStream_t s) {
    QAMD_LDS_OPT_IN((&u8_gemm_qs16_kernel<MODE, LOW, JT, IT>), 160 * 1024);
    const uint32_t nkb = b->frag_nkb;
    constexpr int QS_ROWS = 16 * JT;
    const uint32_t per = (uint32_t)(h->meta.actual_dim / 16);
    const size_t lds_bytes = (size_t)QS_ROWS * (((per + 15) / 16) * 256) + 2 * QS_ROWS * 4 + 64 + kQsSlice * 4;
    const uint32_t grid = (uint32_t)std::max(1, device_info().cu_count / 8) * 8;
    for (uint64_t q_base = 0; q_base < b->n_queries; q_base += kQsSlice) {
        const uint32_t nq = (uint32_t)std::min<uint64_t>(kQsSlice, b->n_queries - q_base);
        BatchFilter f = filt;
        if (MODE == 1 || MODE == 2) {
            f.pivot_scores += q_base;
            f.query_base = (uint32_t)q_base;
            f.wave_base = (uint32_t)(q_base / kQsSlice) * pp_waves_per_launch();
        }
        hipLaunchKernelGGL((u8_gemm_qs16_kernel<MODE, LOW, JT, IT>), dim3(grid), dim3(512), lds_bytes, s, codes, v_offsets,
                           b->frag.as<uint4>() + (q_base / 16) * nkb * 128, b->offsets.as<float>() + q_base,
                           (MODE == 1 || MODE == 2) ? bq + q_base : nullptr, h->meta.multiplier, (uint32_t)n_rows, nq,
                           (uint32_t)round_up((uint64_t)nq, 64), (uint32_t)h->meta.actual_dim,
                           (MODE == 0 || MODE == 3) ? out + q_base * out_pitch : out, out_pitch, f);
        QAMD_HIP(hipGetLastError());
    }
    return QAMD_OK;
}

// Queries in registers, rows through a double-buffered LDS slab (u8_gemm_qr16_kernel): batches cut into passes of 256
// queries.  QAMD_GEMM_CFG=g forces it where it can run (developer A/B); QAMD_QR_MIN / QAMD_QR_MAX move its range.
// Measured (profiles/r03_qs_experiments.txt §7), whole topk_batch(30) ms, row-streaming passes / this kernel:
//   10M x 768:    129 q 2.17 / 1.65   192 q 2.21 / 1.75   256 q 2.26 / 1.91     (two passes: 512 q 3.81 against 3.66 query-streaming)
//   7.5M x 1024:   65 q 1.52 / 1.23   129 q 2.18 / 1.55   256 q 2.75 / 1.84
//   15M x 512:    129 q 2.28 / 1.76   256 q 2.41 / 2.08       30M x 256:  129 q 2.68 / 2.44   256 q 2.84 / 2.55
// -> one pass only: 129 .. 256 queries (from 65 on 1024-byte rows, where the row-streaming kernel needs two 64-query tiles).
constexpr uint64_t kQrQueries = 256;
inline bool qr_possible(const qamd_u8 *h, const qamd_u8_query_batch *b, bool filter_mode) {
    const float m = h->meta.multiplier;
    if (filter_mode && !(std::isfinite(m) && m != 0.0f)) return false;
    const uint64_t ad = h->meta.actual_dim;
    return b->frag.ptr && b->frag16 && (ad == 256 || ad == 384 || ad == 512 || ad == 768 || ad == 1024);
}
bool qr_selected(const qamd_u8 *h, const qamd_u8_query_batch *b, bool filter_mode) {
    static const char *cfg = dev_env("QAMD_GEMM_CFG");
    static const char *lo = dev_env("QAMD_QR_MIN"), *hi = dev_env("QAMD_QR_MAX");
    if (cfg && cfg[0] != 'g') return false;
    if (!qr_possible(h, b, filter_mode)) return false;
    if (cfg) return true;
    // a store of fewer than ~4 slabs per workgroup (a small Qdrant segment) leaves this persistent grid a ragged tail too:
    // the same guard as qs_selected (the row-streaming tiles split such a store evenly)
    if (!lo && !hi && h->count < 131072 && h->meta.actual_dim <= 1152) return false;
    const uint64_t q_min = lo ? (uint64_t)atoll(lo) : (h->meta.actual_dim == 1024 ? 65 : 129);
    const uint64_t q_max = hi ? (uint64_t)atoll(hi) : kQrQueries;
    return b->n_queries >= q_min && b->n_queries <= q_max;
}

template <int MODE, bool LOW>
qamd_status launch_gemm_qr_cfg(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                               const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                               const BatchFilter &filt, const int *bq, hipStream_t s) {
    const uint32_t nkb = b->frag_nkb;
    const uint64_t ad = h->meta.actual_dim;
    const size_t lds_bytes = 2 * (size_t)64 * round_up(ad, 256) + 4 * 64 * 4 + 64;
    const uint32_t grid = (uint32_t)std::max(1, device_info().cu_count / 8) * 8;
    for (uint64_t q_base = 0; q_base < b->n_queries; q_base += kQrQueries) {
        const uint32_t nq = (uint32_t)std::min<uint64_t>(kQrQueries, b->n_queries - q_base);
        BatchFilter f = filt;
        if (MODE == 1 || MODE == 2) {
            f.pivot_scores += q_base;
            f.query_base = (uint32_t)q_base;
            f.wave_base = (uint32_t)(q_base / kQrQueries) * pp_waves_per_launch();
        }
#define QAMD_QR(NS_)                                                                                                          \
    do {                                                                                                                     \
        QAMD_LDS_OPT_IN((&u8_gemm_qr16_kernel<MODE, LOW, NS_>), 160 * 1024); \
        hipLaunchKernelGGL((u8_gemm_qr16_kernel<MODE, LOW, NS_>), dim3(grid), dim3(512), lds_bytes, s, codes, v_offsets,     \
                           b->frag.as<uint4>() + (q_base / 16) * nkb * 128, b->offsets.as<float>() + q_base,                 \
                           (MODE == 1 || MODE == 2) ? bq + q_base : nullptr, h->meta.multiplier, (uint32_t)n_rows, nq,       \
                           (uint32_t)ad, (MODE == 0 || MODE == 3) ? out + q_base * out_pitch : out, out_pitch, f);           \
    } while (0)
        if (nkb == 2) QAMD_QR(4);
        else if (nkb == 3) QAMD_QR(6);
        else if (nkb == 4) QAMD_QR(8);
        else if (nkb == 6) QAMD_QR(12);
        else QAMD_QR(16);
#undef QAMD_QR
        QAMD_HIP(hipGetLastError());
    }
    return QAMD_OK;
}

template <int MODE>
qamd_status launch_gemm_qr(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                           const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                           const BatchFilter &filt, hipStream_t s) {
    if (MODE == 0) return launch_gemm_qr_cfg<0, false>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, nullptr, s);
    if (MODE == 3) return launch_gemm_qr_cfg<3, false>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, nullptr, s);
    constexpr int M = (MODE == 1 || MODE == 2) ? MODE : 1;
    const bool low = (h->meta.multiplier < 0.0f) != (MODE == 2);
    int *bq = filt.query_bounds;  // per-query integer bounds of the pre-filter, behind the pivots in stream order
    if (low)
        hipLaunchKernelGGL(qs_bounds_kernel<true>, dim3((unsigned)(b->q_pad / 256)), dim3(256), 0, s, filt.pivot_scores,
                           b->offsets.as<float>(), h->meta.multiplier, filt.largest, (uint32_t)b->q_pad, bq);
    else
        hipLaunchKernelGGL(qs_bounds_kernel<false>, dim3((unsigned)(b->q_pad / 256)), dim3(256), 0, s, filt.pivot_scores,
                           b->offsets.as<float>(), h->meta.multiplier, filt.largest, (uint32_t)b->q_pad, bq);
    QAMD_HIP(hipGetLastError());
    return low ? launch_gemm_qr_cfg<M, true>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, bq, s)
               : launch_gemm_qr_cfg<M, false>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, bq, s);
}

// Queries resident, rows streamed (u8_gemm_rq16_kernel): the filter pass of topk_batch for batches of kRqMinQueries and more
// on rows of 256 / 384 / 512 / 768 bytes, in groups of query tiles that run side by side on the CUs of an XCD.
// QAMD_GEMM_CFG=s forces it where it can run, QAMD_RQ=0 switches it off (developer A/B, tools/lib build).
constexpr uint64_t kRqMinQueries = 129;
inline RqGeometry rq_geometry(uint64_t n_queries, uint32_t nsteps) {
    // the fewest groups whose tile pairs fit a CU's LDS, the pairs spread evenly; 32 / G row streams per XCD (QAMD_RQ_GROUPS:
    // developer A/B).  More groups than needed only add L2 -> CU traffic: at 10M x 768, 1024 queries, 6 groups 8.3 ms, 8 groups 8.7.
    static const char *eg = dev_env("QAMD_RQ_GROUPS");
    RqGeometry g{};
    g.n_tiles = (uint32_t)(round_up(n_queries, 32) / 16);
    const uint32_t pairs = g.n_tiles / 2, cap_pairs = rq_tile_cap(nsteps) / 2;
    g.groups = eg ? (uint32_t)atoi(eg) : (pairs + cap_pairs - 1) / cap_pairs;
    if (g.groups == 0 || g.groups > 8 || g.groups > pairs || (pairs + g.groups - 1) / g.groups > cap_pairs) {
        g.groups = 0;
        return g;
    }
    g.pairs_lo = pairs / g.groups;
    g.pairs_extra = pairs % g.groups;
    g.streams_per_xcd = 32u / g.groups;
    return g;
}
bool rq_selected(const qamd_u8 *h, const qamd_u8_query_batch *b) {
    static const char *cfg = dev_env("QAMD_GEMM_CFG"), *off = dev_env("QAMD_RQ"), *lo = dev_env("QAMD_RQ_MIN"), *hi = dev_env("QAMD_RQ_MAX");
    if ((cfg && cfg[0] != 's') || (off && off[0] == '0')) return false;
    const float m = h->meta.multiplier;
    const uint64_t ad = h->meta.actual_dim;
    if (!(std::isfinite(m) && m != 0.0f) || !b->frag.ptr || !b->frag16) return false;
    if (!(ad == 256 || ad == 384 || ad == 512 || ad == 768)) return false;
    if (device_info().cu_count != 256 || b->q_pad < round_up(b->n_queries, 32)) return false;  // (8 XCDs of 32 CUs: the group / stream map)
    const RqGeometry g = rq_geometry(b->n_queries, (uint32_t)(ad / 64));
    if (g.groups == 0) return false;  // (more than eight LDS images)
    if (cfg) return true;
    if (h->count < 131072) return false;  // (a small store: the row-streaming tiles split it evenly)
    if (lo || hi) return b->n_queries >= (lo ? (uint64_t)atoll(lo) : kRqMinQueries) && b->n_queries <= (hi ? (uint64_t)atoll(hi) : ~0ull);
    if (b->n_queries < kRqMinQueries) return false;
    // Measured, whole topk_batch(30) ms, before / this kernel (tools/experiments/u8_rq_sweep.sh, profiles/r04_u8_rq.txt):
    //   15M x 512:  129 q 1.95 / 1.49  288 q 3.42 / 2.19 | 289 q 3.40 / 2.64  576 q 5.32 / 4.14 | 768 q 5.84 / 5.48  1152 q 7.91 / 7.61
    //   30M x 256:  129 q 2.75 / 1.60  608 q 7.04 / 4.77 | 609 q 7.05 / 5.10  1216 q 10.6 / 8.89 | 1800 q 14.0 / 12.5  2400 q 19.7 / 16.6
    // 768-byte rows, the K-outer form (u8_gemm_rk16_kernel), before / with it:  129 q 1.64 / 1.49   192 q 1.78 / 1.54 | 193-256 q (two
    // groups) 1.92-2.03 / 2.33-2.38: the queries-in-registers kernel keeps those | 257 q 2.94 / 2.50  288 q 2.99 / 2.51  384 q 3.26 / 2.62 |
    // three groups: 400 q 3.52 / 3.65  512 q 3.83 / 3.91 (no), 576 q 4.80 / 4.00 (past the query-streaming kernel's step at 513) |
    // four: 640 q 5.07 / 4.71  768 q 5.50 / 5.08 | five and more (30 of an XCD's 32 CUs, or half-empty groups): 832 q 5.85 / 6.25,
    // 1024 q in eight groups of eight tiles 6.44 / 6.44
    if (ad == 768)
        return g.groups == 1 || (g.groups == 2 && b->n_queries > kQrQueries) || (g.groups == 3 && b->n_queries > 512) || g.groups == 4;
    return g.groups <= 4;
}
template <int MODE, bool LOW>
qamd_status launch_gemm_rq_cfg(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes, const float *v_offsets,
                               uint64_t n_rows, const BatchFilter &filt, const int *bq, hipStream_t s) {
    const uint32_t nsteps = (uint32_t)(h->meta.actual_dim / 64);
    const RqGeometry g = rq_geometry(b->n_queries, nsteps);
    const size_t max_tiles = 2 * (size_t)(g.pairs_lo + (g.pairs_extra ? 1 : 0));
    const size_t lds_bytes = max_tiles * nsteps * 1024 + max_tiles * 64 + 64;
    BatchFilter f = filt;
    f.query_base = 0;
    f.wave_base = 0;
    static const char *ek = dev_env("QAMD_RQ_K");  // developer A/B: 0 = the tile-outer form on 768-byte rows as well
    if (nsteps == 12 && !(ek && ek[0] == '0')) {    // the K-outer form: all tiles' accumulators in registers
#define QAMD_RK(NT_)                                                                                                       \
    do {                                                                                                                  \
        QAMD_LDS_OPT_IN((&u8_gemm_rk16_kernel<MODE, LOW, 12, NT_>), 160 * 1024);                                           \
        hipLaunchKernelGGL((u8_gemm_rk16_kernel<MODE, LOW, 12, NT_>), dim3(256), dim3(512),                                \
                           (size_t)NT_ * 12 * 1024 + (size_t)NT_ * 64 + 64, s, codes, v_offsets, b->frag.as<uint4>(),     \
                           b->offsets.as<float>(), bq, h->meta.multiplier, (uint32_t)n_rows, g, f);                       \
    } while (0)
        if (max_tiles <= 8) QAMD_RK(8);
        else if (max_tiles <= 10) QAMD_RK(10);
        else QAMD_RK(12);
#undef QAMD_RK
        QAMD_HIP(hipGetLastError());
        return QAMD_OK;
    }
#define QAMD_RQ(NS_)                                                                                                       \
    do {                                                                                                                  \
        QAMD_LDS_OPT_IN((&u8_gemm_rq16_kernel<MODE, LOW, NS_>), 160 * 1024);                                               \
        hipLaunchKernelGGL((u8_gemm_rq16_kernel<MODE, LOW, NS_>), dim3(256), dim3(512), lds_bytes, s, codes, v_offsets,    \
                           b->frag.as<uint4>(), b->offsets.as<float>(), bq, h->meta.multiplier, (uint32_t)n_rows, g, f);  \
    } while (0)
    switch (nsteps) {
        case 4: QAMD_RQ(4); break;
        case 6: QAMD_RQ(6); break;
        case 8: QAMD_RQ(8); break;
        default: QAMD_RQ(12); break;
    }
#undef QAMD_RQ
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

template <int MODE>
qamd_status launch_gemm_qs(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                           const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                           const BatchFilter &filt, hipStream_t s) {
    if ((MODE == 1 || MODE == 2) && n_rows == h->count && rq_selected(h, b)) {  // the filter pass over the store itself
        constexpr int M = (MODE == 1 || MODE == 2) ? MODE : 1;
        const bool low = (h->meta.multiplier < 0.0f) != (MODE == 2);
        int *bq = filt.query_bounds;
        if (low)
            hipLaunchKernelGGL(qs_bounds_kernel<true>, dim3((unsigned)(b->q_pad / 256)), dim3(256), 0, s, filt.pivot_scores,
                               b->offsets.as<float>(), h->meta.multiplier, filt.largest, (uint32_t)b->q_pad, bq);
        else
            hipLaunchKernelGGL(qs_bounds_kernel<false>, dim3((unsigned)(b->q_pad / 256)), dim3(256), 0, s, filt.pivot_scores,
                               b->offsets.as<float>(), h->meta.multiplier, filt.largest, (uint32_t)b->q_pad, bq);
        QAMD_HIP(hipGetLastError());
        return low ? launch_gemm_rq_cfg<M, true>(h, b, codes, v_offsets, n_rows, filt, bq, s)
                   : launch_gemm_rq_cfg<M, false>(h, b, codes, v_offsets, n_rows, filt, bq, s);
    }
    if (qr_selected(h, b, MODE != 0)) return launch_gemm_qr<MODE>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
    const bool wide = b->frag_nkb <= 9;  // 128 resident rows fit (rows of up to 1152 B), else 96
    const bool tall = b->frag_nkb <= 8;  // 16x16x64 form: 128 resident rows of up to 1024 bytes, else 96
    const bool small = b->n_queries <= kQs16SmallBatch;  // chunks of 32 queries: a chunk for every wave
#define QAMD_QS16(M_, LOW_, BQ_)                                                                                                  \
    (tall ? (small ? launch_gemm_qs16_cfg<M_, LOW_, 8, 2>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, BQ_, s)           \
                   : launch_gemm_qs16_cfg<M_, LOW_, 8, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, BQ_, s))          \
          : (small ? launch_gemm_qs16_cfg<M_, LOW_, 6, 2>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, BQ_, s)           \
                   : launch_gemm_qs16_cfg<M_, LOW_, 6, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, BQ_, s)))
    if (b->frag16 && MODE == 0) return QAMD_QS16(0, false, nullptr);
    if (b->frag16 && MODE == 3) return QAMD_QS16(3, false, nullptr);
    if (MODE == 0)
        return wide ? launch_gemm_qs_cfg<0, false, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, nullptr, s)
                    : launch_gemm_qs_cfg<0, false, 3>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, nullptr, s);
    if (MODE == 3)
        return wide ? launch_gemm_qs_cfg<3, false, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, nullptr, s)
                    : launch_gemm_qs_cfg<3, false, 3>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, nullptr, s);
    constexpr int M = (MODE == 1 || MODE == 2) ? MODE : 1;
    const bool low = (h->meta.multiplier < 0.0f) != (MODE == 2);
    // per-query integer bounds of the pre-filter, behind the pivots in stream order
    int *bq = filt.query_bounds;
    if (low)
        hipLaunchKernelGGL(qs_bounds_kernel<true>, dim3((unsigned)(b->q_pad / 256)), dim3(256), 0, s, filt.pivot_scores,
                           b->offsets.as<float>(), h->meta.multiplier, filt.largest, (uint32_t)b->q_pad, bq);
    else
        hipLaunchKernelGGL(qs_bounds_kernel<false>, dim3((unsigned)(b->q_pad / 256)), dim3(256), 0, s, filt.pivot_scores,
                           b->offsets.as<float>(), h->meta.multiplier, filt.largest, (uint32_t)b->q_pad, bq);
    QAMD_HIP(hipGetLastError());
    if (b->frag16) return low ? QAMD_QS16(M, true, bq) : QAMD_QS16(M, false, bq);
#undef QAMD_QS16
    if (wide)
        return low ? launch_gemm_qs_cfg<M, true, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, bq, s)
                   : launch_gemm_qs_cfg<M, false, 4>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, bq, s);
    return low ? launch_gemm_qs_cfg<M, true, 3>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, bq, s)
               : launch_gemm_qs_cfg<M, false, 3>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, bq, s);
}

// The query-streaming kernel: rows short enough for a 128-row block in LDS, a fragment-order copy in
// the batch, enough queries to keep the 8 waves of a workgroup busy (64 queries per wave and turn).
bool qs_selected(const qamd_u8 *h, const qamd_u8_query_batch *b, bool filter_mode) {
    static const char *cfg = dev_env("QAMD_GEMM_CFG");
    if (qr_selected(h, b, filter_mode)) return true;  // a form of it (launch_gemm_qs dispatches)
    if (filter_mode && rq_selected(h, b)) return true;  // (the sample pass of such a batch takes the query-streaming forms)
    if (cfg && cfg[0] != 'q') return false;
    const float m = h->meta.multiplier;
    if (filter_mode && !(std::isfinite(m) && m != 0.0f)) return false;
    if (!b->frag.ptr || b->frag_nkb == 0 || b->frag_nkb > 12) return false;
    if (cfg) return true;
    // a store of fewer than ~4 row blocks per workgroup (a small Qdrant segment) leaves the persistent workgroups a
    // ragged tail; the row-streaming tiles split such a store evenly (100k x 768, 1024 queries: 0.24 against 0.28 ms)
    if (h->count < 131072 && h->meta.actual_dim <= 1152) return false;
    return b->n_queries >= qs_min_queries(b->frag_nkb);
}

// The row-streaming kernel: where the ping-pong kernel could run (same pre-filter conditions), the
// query tile fits in LDS, and the batch is small enough to be HBM-bound (QAMD_GEMM_CFG=r / p force).
bool rs_selected(const qamd_u8 *h, const qamd_u8_query_batch *b, bool filter_mode) {
    static const char *cfg = dev_env("QAMD_GEMM_CFG");
    if (cfg && cfg[0] != 'r') return false;
    const float m = h->meta.multiplier;
    if (filter_mode && !(std::isfinite(m) && m != 0.0f)) return false;
    const int mi = rs_frags(b->n_queries, h->meta.actual_dim);
    if (mi == 0 || h->meta.actual_dim > 32768) return false;
    if (cfg) return true;
    // One query tile: every row byte leaves HBM once, at the plain scan's rate.  Several tiles re-read the
    // rows (at HBM pace a line lives ~5 us in the XCD's L2, too short for the tiles' workgroups to share
    // it), which still beats the ping-pong kernel for 128-query tiles up to 768 queries (measured at
    // 10M x 768: 160 q 2.22 vs 2.41 ms, 384 q 3.39 vs 4.51, 512 q 4.29 vs 4.66, 1024 q 8.64 vs 8.70)
    // and loses with the 64-query tiles of longer rows (12.5M x 1536: 96 q 3.88 vs 3.74, 256 q 6.83 vs 5.58).
    const uint64_t tiles = (b->n_queries + 32 * mi - 1) / (32 * mi);
    return tiles == 1 || (mi == 4 && !qs_selected(h, b, filter_mode));
}

template <int MODE>
qamd_status launch_gemm(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint8_t *codes,
                        const float *v_offsets, uint64_t n_rows, float *out, uint64_t out_pitch,
                        const BatchFilter &filt, hipStream_t s) {
    if (n_rows == 0 || b->n_queries == 0) return QAMD_OK;
    // q_pad is a multiple of 256 and the row padding of every store covers a 256-row tile.
    if (qs_selected(h, b, MODE != 0)) return launch_gemm_qs<MODE>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
    if (rs_selected(h, b, MODE != 0)) return launch_gemm_rs<MODE>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
    if (pp_selected(h, b, MODE != 0)) return launch_gemm_pp<MODE>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
    if (b->n_queries > 128) {
        static const char *cfg = dev_env("QAMD_GEMM_CFG");
        if (cfg && cfg[0] == '3')  // two 4-wave workgroups per CU (61 KiB LDS each), 128 q x 256 rows
            return launch_gemm_cfg<MODE, 128, 256, 2, 2, 64>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
        if (cfg && cfg[0] == '4')  // same, 256 q x 128 rows
            return launch_gemm_cfg<MODE, 256, 128, 2, 2, 64>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
        if (cfg && cfg[0] == '5')  // one wave per SIMD, 128 x 128 outputs per wave (256 accumulator registers)
            return launch_gemm_cfg<MODE, 256, 256, 2, 2, 128>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
        return launch_gemm_cfg<MODE, 256, 256, 2, 4, 128>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
    }
    return launch_gemm_cfg<MODE, 128, 128, 2, 2, 128>(h, b, codes, v_offsets, n_rows, out, out_pitch, filt, s);
}

// The pivot sample of a store, gathered once per handle: sample row j is store row hash(j) whatever
// the sample size, so one sample of the largest size any call can ask for (8 n / 512 rows, at most
// 524288) serves every call as a prefix.  Rows past the prefix inside the last row tile are real
// sample rows instead of zeros: the kernels compute them and store nothing (row >= n_rows).
qamd_status sample_store(const qamd_u8 *h, uint32_t rows_all, hipStream_t s, const uint8_t **codes, const float **offs) {
    std::lock_guard<std::mutex> lk(h->sample_mu);
    if (h->sample_rows < rows_all) {
        const uint64_t ad = h->meta.actual_dim;
        qamd::DevBuf c, o;
        QAMD_TRY(c.alloc((uint64_t)(rows_all + 512) * ad));
        QAMD_TRY(o.alloc((uint64_t)(rows_all + 512) * 4));
        hipLaunchKernelGGL(gather_rows_kernel, dim3(device_info().cu_count * 8), dim3(256), 0, s, h->codes.as<uint4>(),
                           h->offsets.as<float>(), h->row_chunks, h->count, rows_all, 512u, c.as<uint4>(), o.as<float>());
        QAMD_HIP(hipGetLastError());
        QAMD_HIP(hipStreamSynchronize(s));  // once per handle: every later call, on any stream, just reads it
        h->sample_codes = std::move(c);
        h->sample_offsets = std::move(o);
        h->sample_rows = rows_all;
    }
    *codes = h->sample_codes.as<uint8_t>();
    *offs = h->sample_offsets.as<float>();
    return QAMD_OK;
}

}  // namespace

extern "C" {

#ifdef QAMD_GEMM_ABLATION
QAMD_API qamd_status qamd_dev_gemm_debug(unsigned int flags) {
    QAMD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg), &flags, sizeof(flags)));
    return QAMD_OK;
}
#endif

#ifdef QAMD_DEV
// Developer hook: device buffer of kStampBlocks * 8 * 16 u64 (or null to switch the timeline off).
QAMD_API qamd_status qamd_dev_gemm_stamps(void *dev_buffer) {
    QAMD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &dev_buffer, sizeof(void *)));
    return QAMD_OK;
}
#endif

qamd_status qamd_u8_encode_query_batch(const qamd_u8 *h, const float *queries, uint64_t n_queries, uint64_t qdim,
                                       qamd_mem queries_mem, void *stream, qamd_u8_query_batch **batch_io) {
    if (!h || !batch_io || (!queries && n_queries && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint64_t ad = qdim + (16 - qdim % 16) % 16;
    qamd_u8_query_batch *b = *batch_io;
    std::unique_ptr<qamd_u8_query_batch> fresh;
    if (!b) {
        fresh.reset(new qamd_u8_query_batch);
        b = fresh.get();
        b->device = h->device;
    }
    const uint64_t q_pad = round_up(std::max<uint64_t>(n_queries, 1), TQ);
    const uint64_t pitch = round_up(std::max<uint64_t>(ad, 16), 64);
    if (b->actual_dim != ad || b->q_pad != q_pad || !b->codes.ptr) {
        b->pitch = pitch;
        QAMD_TRY(b->codes.alloc(q_pad * pitch, true));
        QAMD_TRY(b->offsets.alloc(q_pad * sizeof(float), true));
        b->actual_dim = ad;
        b->q_pad = q_pad;
    } else {
        QAMD_HIP(hipMemsetAsync(b->codes.ptr, 0, b->codes.bytes, s));
        QAMD_HIP(hipMemsetAsync(b->offsets.ptr, 0, b->offsets.bytes, s));
    }
    b->n_queries = n_queries;
    if (n_queries) {
        DevBuf tmp;
        const float *qd = queries;
        if (queries_mem == QAMD_MEM_HOST) {
            QAMD_TRY(tmp.alloc(n_queries * std::max<uint64_t>(qdim, 1) * 4));
            QAMD_TRY(copy_in(tmp.ptr, queries, QAMD_MEM_HOST, n_queries * qdim * 4, s));
            qd = tmp.as<float>();
        }
        QAMD_TRY(u8_encode_queries_device(h, qd, n_queries, qdim, b->codes.as<uint8_t>(), b->pitch, b->offsets.as<float>(), s));
        const uint32_t nkb = (uint32_t)((ad + 127) / 128);
        if (nkb <= 12) {  // rows the query-streaming kernel can hold: the codes again, in MFMA fragment order
            const size_t frag_bytes = (size_t)(q_pad / 32) * nkb * 4096;
            if (b->frag.bytes < frag_bytes) QAMD_TRY(b->frag.alloc(frag_bytes));
            b->frag_nkb = nkb;
            b->frag16 = qs16_wanted(nkb);
            if (b->frag16)
                hipLaunchKernelGGL(swizzle_queries16_kernel, dim3((unsigned)std::min<uint64_t>(2048, (frag_bytes / 16 + 255) / 256)),
                                   dim3(256), 0, s, b->codes.as<uint8_t>(), (uint32_t)b->pitch, (uint32_t)q_pad, nkb, b->frag.as<uint4>());
            else
                hipLaunchKernelGGL(swizzle_queries_kernel, dim3((unsigned)std::min<uint64_t>(2048, (frag_bytes / 16 + 255) / 256)),
                                   dim3(256), 0, s, b->codes.as<uint8_t>(), (uint32_t)b->pitch, (uint32_t)q_pad, nkb, b->frag.as<uint4>());
            QAMD_HIP(hipGetLastError());
        } else {
            b->frag_nkb = 0;
        }
        if (queries_mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));
    }
    if (fresh) *batch_io = fresh.release();
    return QAMD_OK;
}

void qamd_u8_query_batch_free(qamd_u8_query_batch *b) { delete b; }

// Many (query, id list) pairs in one launch (lists.hpp): out[p] = score_point(query l, ids[p]) for p in
// [list_offsets[l], list_offsets[l + 1]) -- one HNSW hop of every in-flight search.
qamd_status qamd_u8_score_ids_batch(const qamd_u8 *h, const qamd_u8_query_batch *b, const uint32_t *list_offsets,
                                    uint32_t n_lists, const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem, float *out,
                                    qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_batch(h, b));
    if (n_lists > b->n_queries)
        return fail(QAMD_ERR_ARGUMENTS, "%u lists, but the batch holds %llu queries", n_lists, (unsigned long long)b->n_queries);
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    return run_lists(list_offsets, n_lists, ids, n_ids, nullptr, lists_mem, out, out_mem, h->count, s, [&](const ListArgs &a) {
        return u8_score_lists(h, b->codes.as<uint8_t>(), b->pitch, b->offsets.as<float>(), a, s);
    });
}

qamd_status qamd_u8_score_batch(const qamd_u8 *h, const qamd_u8_query_batch *b, float *out, qamd_mem out_mem,
                                void *stream) {
    QAMD_TRY(check_batch(h, b));
    if (h->count == 0 || b->n_queries == 0) return QAMD_OK;
    if (!out) return fail(QAMD_ERR_ARGUMENTS, "out is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint64_t total = b->n_queries * h->count;
    DevBuf tmp;
    float *out_dev = out;
    if (out_mem == QAMD_MEM_HOST) {
        QAMD_TRY(tmp.alloc(total * 4));
        out_dev = tmp.as<float>();
    }
    const uint32_t width = u8_multi_width(h);
    // (from three queries on the row-streaming MFMA kernel streams the rows as fast and does not slow down per query)
    if (h->meta.vector_parameters.distance_type == QAMD_L1 ||
        (width && b->n_queries >= 2 && b->n_queries <= width && (b->n_queries == 2 || !rs_selected(h, b, false)))) {
        // sum |q - v| is not a contraction (no MFMA form), and for a handful of queries the vector-ALU
        // multi-query scan streams the rows faster than the matrix-core kernel's LDS-DMA path
        QAMD_TRY(u8_score_batch_scans(h, b->codes.as<uint8_t>(), b->pitch, b->offsets.as<float>(), (uint32_t)b->n_queries,
                                      out_dev, s));
    } else {
        QAMD_TRY(launch_gemm<0>(h, b, h->codes.as<uint8_t>(), h->offsets.as<float>(), h->count, out_dev, h->count,
                                BatchFilter{}, s));
    }
    if (out_mem == QAMD_MEM_HOST) QAMD_TRY(copy_out(out, QAMD_MEM_HOST, out_dev, total * 4, s));
    return QAMD_OK;
}

qamd_status qamd_u8_topk_batch(const qamd_u8 *h, const qamd_u8_query_batch *b, uint32_t k, int largest,
                               uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_batch(h, b));
    if (k == 0 || b->n_queries == 0) return QAMD_OK;
    if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
    if (!out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null output");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint64_t Q = b->n_queries, n = h->count;
    const bool l1 = h->meta.vector_parameters.distance_type == QAMD_L1;
    if (n > (2u << 20) && (l1 || (Q <= u8_multi_width(h) && Q >= 2 && (Q == 2 || !rs_selected(h, b, true)))))
        // L1 has no matrix form; and two queries (one pass of the vector-ALU multi-query scan) stream the
        // rows at the single-query scan's rate with less overhead around it than the matrix-core pass (from
        // three queries on the row-streaming kernel wins: 1.26 against 1.31-1.39 ms per 10M x 768).  Per-query sample + pivot, ONE filtering pass per group of queries, one
        // status read-back per 32 queries.
        return u8_topk_batch_scans(h, b->codes.as<uint8_t>(), b->pitch, b->offsets.as<float>(), (uint32_t)Q, k, largest,
                                   out_ids, out_scores, out_mem, s);
    // Pivot rank r of S sampled rows: the number of rows at least as good as the pivot is about
    // n*Beta(r, S-r+1): mean n*r/S, relative spread 1/sqrt(r).  With many queries per call both tails
    // matter (a list that overflows kBatchCap or holds fewer than k rows sends its query to the exact
    // single-query path, milliseconds each), and so does the mean: every passing (query, row) pair
    // costs an exact epilogue + a list append inside the GEMM (2048 expected per query instead of 512
    // measured +7..10 % on the whole call).  So: about max(512, 3k) rows expected to pass at pivot
    // rank >= 8 (P(fewer than k) < 1e-7, P(more than 8192) ~ 0); S = 8 n / that, capped at 524288
    // sampled rows (beyond 33M rows the expected count grows instead of r shrinking).
    // Few queries: a smaller sample (cheaper pivot pass) and more candidates per query instead (measured at
    // 10M x 768, 5-128 queries: 1024 expected beats 512 by 0.01-0.02 ms and 2048 by 0.03-0.08 ms - the
    // per-query scatter and sort of the candidates grow faster than the sample pass shrinks).
    const double want = std::max<double>(Q <= 128 ? 1024.0 : 512.0, 3.0 * k);
    // small stores (Qdrant segments: 1e5..1e6 rows) take the same route with a smaller sample: the
    // per-query fallback costs a launch chain per query, the matrix-core pass one for the whole batch
    const double s_min = n < (1u << 20) ? 2048.0 : (double)kTopkSample;
    const double s_mem_cap = std::max<double>(s_min, std::floor((double)(6ull << 30) / (4.0 * (double)Q) / 256.0) * 256.0);  // <= 6 GB of sample scores
    const uint32_t S = (uint32_t)std::min<double>(std::min<double>(524288.0, s_mem_cap),
                                                  std::max<double>(s_min, round_up((uint64_t)(8.0 * (double)n / want), 256)));
    const double target = std::max<double>(want, 8.0 * (double)n / (double)S);
    const uint32_t r = n ? (uint32_t)std::ceil((double)S * target / (double)n) : 0;
    // (the sample scores are Q x S f32: 0.6 GB at 1024 queries and 10M rows, 5 GB at 8192 queries)
    const bool fused = n >= 32768 && r <= 64 && h->meta.vector_parameters.distance_type != QAMD_L1;

    StreamBuf ids_tmp, sc_tmp;
    uint32_t *ids_dev = out_ids;
    float *sc_dev = out_scores;
    if (out_mem == QAMD_MEM_HOST) {
        QAMD_TRY(ids_tmp.alloc(Q * k * 4, s));
        QAMD_TRY(sc_tmp.alloc(Q * k * 4, s));
        ids_dev = ids_tmp.as<uint32_t>();
        sc_dev = sc_tmp.as<float>();
    }
    std::vector<uint32_t> status(Q, 1);
    if (fused) {
        const uint64_t ad = h->meta.actual_dim;
        // ONE stream-ordered allocation for all scratch of the call, carved up below: hipFreeAsync
        // costs ~65 us per buffer on this runtime, and ten buffers were a third of a small batch's time.
        const bool qs = qs_selected(h, b, true);
        const bool rs = !qs && rs_selected(h, b, true);
        const bool pp = qs || rs || pp_selected(h, b, true);  // all three append to wave-private lists
        uint32_t n_lists = 0, wave_cap = 0;
        if (pp) {
            // wave-private lists: 4x the expected appends per wave (about `target`..2*target per query)
            n_lists = gemm_launches(h, b, rs, qs) * pp_waves_per_launch();
            const double per_wave = 2.0 * target * (double)Q / (double)n_lists;
            wave_cap = (uint32_t)std::min<double>(1u << 20, std::max<double>(1024.0, 4.0 * per_wave));
        }
        size_t arena_bytes = 0;
        auto reserve = [&](size_t bytes) {
            const size_t off = arena_bytes;
            arena_bytes += round_up(std::max<size_t>(bytes, 16), 256);
            return off;
        };
        // zero-initialised part first (one memset): pivots, counters, wave counts, overflow flag
        const size_t o_pivots = reserve(b->q_pad * 4);
        const size_t o_counters = reserve(b->q_pad * kCounterStride * 4);
        const size_t o_wcounts = reserve((uint64_t)n_lists * 4);
        const size_t zero_bytes = arena_bytes;
        // the sample sub-store: the handle's cached one (a prefix of it), else gathered into the arena
        const uint32_t rows_all = (uint32_t)std::min<double>(524288.0, std::max<double>(s_min, (double)round_up((uint64_t)(8.0 * (double)n / 512.0), 256)));
        const uint8_t *s_codes = nullptr;
        const float *s_offs = nullptr;
        const bool cached = S <= rows_all && sample_store(h, rows_all, s, &s_codes, &s_offs) == QAMD_OK;
        const size_t o_codes = reserve(cached ? 16 : (uint64_t)(S + 512) * ad);  // + one (largest) tile of zero rows
        const size_t o_offs = reserve(cached ? 16 : (uint64_t)(S + 512) * 4);
        const size_t o_scores = reserve(qs ? Q * ((uint64_t)S / 64 + 2) * 4 : Q * (uint64_t)S * 4);  // sample scores, or block bests
        const size_t o_cand = reserve(Q * (uint64_t)kBatchCap * 8);
        const size_t o_status = reserve((Q + 1) * 4);  // per-query status, then the wave-list overflow flag
        const size_t o_bounds = reserve(b->q_pad * 4);
        const size_t o_wcand = reserve((uint64_t)n_lists * wave_cap * sizeof(uint4));
        StreamBuf arena;
        QAMD_TRY(arena.alloc(arena_bytes, s));
        char *base = arena.as<char>();
        QAMD_HIP(hipMemsetAsync(base, 0, zero_bytes, s));
        float *pivots = reinterpret_cast<float *>(base + o_pivots);
        uint32_t *counters = reinterpret_cast<uint32_t *>(base + o_counters);
        uint32_t *wave_counts = reinterpret_cast<uint32_t *>(base + o_wcounts);
        float *s_scores = reinterpret_cast<float *>(base + o_scores);
        unsigned long long *cand = reinterpret_cast<unsigned long long *>(base + o_cand);
        // The per-query status words and the overflow flag come back through the calling thread's mapped
        // host scratch when they fit (the kernels write host memory directly: one stream sync, no copy
        // calls), else through one copy from the arena.
        const HostScratch hs = Q + 1 <= 2048 ? host_scratch() : HostScratch{};
        uint32_t *status_dev = hs.dev ? hs.dev : reinterpret_cast<uint32_t *>(base + o_status);
        uint32_t *overflow_dev = status_dev + Q;
        if (hs.host) hs.host[Q] = 0;
        else QAMD_HIP(hipMemsetAsync(overflow_dev, 0, 4, s));
        uint4 *wave_cand = reinterpret_cast<uint4 *>(base + o_wcand);
        if (!cached) {
            uint8_t *g_codes = reinterpret_cast<uint8_t *>(base + o_codes);
            float *g_offs = reinterpret_cast<float *>(base + o_offs);
            hipLaunchKernelGGL(gather_rows_kernel, dim3(device_info().cu_count * 8), dim3(256), 0, s, h->codes.as<uint4>(),
                               h->offsets.as<float>(), h->row_chunks, n, S, 512u, reinterpret_cast<uint4 *>(g_codes), g_offs);
            s_codes = g_codes;
            s_offs = g_offs;
        }
        if (qs) {
            // the query-streaming kernel hands back the best sample score of every (query, 128- or 96-row
            // block) instead of the Q x S score matrix (0.64 GB written and read back at 1024 queries and
            // 10M rows); the pivot is the r-th best of those: the r best sample rows of a query share a
            // block with probability ~ r^2 / (2 blocks), and a pivot that is a little off only moves the
            // candidate count (the filter pass, not the pivot, decides what is in the result)
            const uint32_t rows_per_block = qr_selected(h, b, true) ? 64 : (b->frag16 ? b->frag_nkb <= 8 : b->frag_nkb <= 9) ? 128 : 96, s_blocks = (S + rows_per_block - 1) / rows_per_block;
            BatchFilter fs{};
            fs.largest = largest;
            QAMD_TRY(launch_gemm_qs<3>(h, b, s_codes, s_offs, S, s_scores, s_blocks, fs, s));
            hipLaunchKernelGGL(batch_pivot_kernel, dim3((unsigned)b->q_pad), dim3(1024), 0, s, s_scores, s_blocks,
                               (uint64_t)s_blocks, r, largest, (uint32_t)Q, pivots, counters);
        } else {
            QAMD_TRY(launch_gemm<0>(h, b, s_codes, s_offs, S, s_scores, S, BatchFilter{}, s));
            hipLaunchKernelGGL(batch_pivot_kernel, dim3((unsigned)b->q_pad), dim3(1024), 0, s, s_scores, S, (uint64_t)S, r,
                               largest, (uint32_t)Q, pivots, counters);
        }
        BatchFilter f{};
        f.pivot_scores = pivots;
        f.counters = counters;
        f.candidates = cand;
        f.largest = largest;
        f.query_bounds = reinterpret_cast<int *>(base + o_bounds);
        if (pp) {
            f.wave_cap = wave_cap;
            f.wave_cand = wave_cand;
            f.wave_counts = wave_counts;
        }
        if (largest)
            QAMD_TRY(launch_gemm<1>(h, b, h->codes.as<uint8_t>(), h->offsets.as<float>(), n, nullptr, 0, f, s));
        else
            QAMD_TRY(launch_gemm<2>(h, b, h->codes.as<uint8_t>(), h->offsets.as<float>(), n, nullptr, 0, f, s));
        if (pp && Q <= 4096)
            hipLaunchKernelGGL(wave_scatter_grouped_kernel, dim3(std::min<uint32_t>(n_lists, 128)), dim3(1024), (size_t)Q * 8, s,
                               wave_cand, wave_counts, f.wave_cap, n_lists, (uint32_t)Q, counters, cand, overflow_dev);
        else if (pp)
            hipLaunchKernelGGL(wave_scatter_kernel, dim3(n_lists), dim3(256), 0, s, wave_cand, wave_counts, f.wave_cap,
                               counters, cand, overflow_dev);
        if (k <= kSmallTopkMaxK)
            hipLaunchKernelGGL(batch_emit_wave_kernel, dim3((unsigned)Q), dim3(1024), 0, s, cand, counters, n, k, largest,
                               ids_dev, sc_dev, status_dev);
        else
            hipLaunchKernelGGL(batch_emit_kernel, dim3((unsigned)Q), dim3(1024), 0, s, cand, counters, n, k, largest, ids_dev,
                               sc_dev, status_dev);
        QAMD_HIP(hipGetLastError());
        uint32_t overflow = 0;
        if (hs.host) {
            QAMD_HIP(hipStreamSynchronize(s));
            std::copy(hs.host, hs.host + Q, status.begin());
            overflow = hs.host[Q];
        } else {
            std::vector<uint32_t> back(Q + 1);
            QAMD_TRY(copy_out(back.data(), QAMD_MEM_HOST, status_dev, (Q + 1) * 4, s));  // synchronises
            std::copy(back.begin(), back.begin() + Q, status.begin());
            overflow = back[Q];
        }
        if (pp && overflow) std::fill(status.begin(), status.end(), 1u);  // a wave list overflowed: redo all exactly
        static const bool debug_topk = dev_env("QAMD_DEBUG_TOPK") != nullptr;
        if (debug_topk) {
            std::vector<uint32_t> cnt(b->q_pad * kCounterStride);
            (void)hipMemcpy(cnt.data(), counters, cnt.size() * 4, hipMemcpyDeviceToHost);
            uint32_t mx = 0, mn = ~0u, redo = 0;
            uint64_t sum = 0;
            for (uint64_t q = 0; q < Q; q++) {
                const uint32_t c = cnt[q * kCounterStride];
                mx = std::max(mx, c);
                mn = std::min(mn, c);
                sum += c;
                redo += status[q];
            }
            fprintf(stderr, "[qamd topk_batch] Q=%llu r=%u candidates min/mean/max = %u/%llu/%u, %u queries redone\n",
                    (unsigned long long)Q, r, mn, (unsigned long long)(sum / Q), mx, redo);
        }
    }
    // Queries not served by the fused pass (small stores, overflowed lists): exact single-query path.
    for (uint64_t q = 0; q < Q; q++) {
        if (!status[q]) continue;
        QAMD_TRY(u8_topk_single(h, b->codes.as<uint8_t>() + q * b->pitch, b->offsets.as<float>() + q, k, largest,
                                ids_dev + q * k, sc_dev + q * k, QAMD_MEM_DEVICE, s));
    }
    if (out_mem == QAMD_MEM_HOST) {
        QAMD_TRY(copy_out(out_ids, QAMD_MEM_HOST, ids_dev, Q * k * 4, s));
        QAMD_TRY(copy_out(out_scores, QAMD_MEM_HOST, sc_dev, Q * k * 4, s));
    } else {
        QAMD_HIP(hipStreamSynchronize(s));  // scratch buffers are released on return
    }
    return QAMD_OK;
}

}  // extern "C"
