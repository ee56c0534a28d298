// Product quantizer on MI355X (gfx950): nearest-centroid encode, query LUT, LUT-gather scan,
// and k-means centroid training.
//
// Host side mirrors EncodedVectorsPQ (quantization/src/encoded_vectors_pq.rs).
//
// HBM layout: rows are the reference's m code bytes (m = number of chunks, :109-114) at a
// device stride of round_up(m, 4) so that a row is a whole number of dwords; centroids stay
// centroid-major with full-dim rows, 256 x dim f32 (Metadata.centroids, :39-44).
//
// Scan (score_point_sse, :405-440, is the order reproduced bit for bit): the query's
// chunk-major LUT (m x 256 f32; 96 KiB at m = 96) is staged ONCE per workgroup in LDS — one
// 1024-thread workgroup per CU, the whole CU's 160 KiB — and four adjacent lanes own one row:
// lane k plays SSE lane k, walking chunks k, k+4, k+8, ... in order with one ds_read_b32
// gather and one f32 add each, so its running sum is exactly the reference's lane sum; the
// combine is (l0+l2)+(l1+l3) by two cross-lane adds and lane 0 appends the m%4 tail.  Bound:
// 96 B/row of HBM reads nominally, but the 96 bank-conflicting LDS gathers per row are the
// expected limiter.
#include <algorithm>
#include <cmath>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "lists.hpp"
#include "topk.hpp"
#include "topk_device.hpp"

#pragma clang fp contract(off)

using namespace qamd;

namespace {

constexpr int kBlock = 256;
constexpr int kScanBlock = 1024;
constexpr uint64_t kRowPad = 1024;
constexpr int kCentroids = QAMD_PQ_CENTROIDS;
constexpr uint64_t kKmeansSample = 10000;  // KMEANS_SAMPLE_SIZE (:22)
constexpr int kKmeansMaxIter = 100;        // KMEANS_MAX_ITERATIONS (:23)
constexpr float kKmeansAccuracy = 1e-5f;   // KMEANS_ACCURACY (:24)
constexpr size_t kLdsBudget = 160 * 1024 - 1024;
// The whole-store scan keeps a LUT slice of at most 9 row pieces (144 chunks, 144 KiB) in LDS per launch.  A row that needs
// several slices is laid out on a 128-byte pitch and scanned in slices of 8 pieces = one 128-byte line each: a slice
// launch then reads whole lines that no other slice touches (m = 192 on the natural 192-byte pitch cut every row's
// lines across both slices: 6 line fetches per 2 rows instead of 3).
constexpr uint32_t kMaxSlicePieces = 9, kSlicePiecesAligned = 8;
inline uint64_t valid_pieces(uint64_t m) { return (m + 15) / 16; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
    u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

// ------------------------------------------------------------------------------ scan
// LUT_IN_LDS: lut staged in dynamic LDS (m*1024 bytes); else gathered through L1/L2.
// ids == nullptr scans rows [0, n).
template <bool LUT_IN_LDS, bool VEC16>
__global__ __launch_bounds__(kScanBlock) void pq_scan_kernel(const uint32_t *__restrict__ rows32,
                                                            const float *__restrict__ lut_g,
                                                            const uint32_t *__restrict__ ids, uint64_t n,
                                                            uint32_t n_rows, uint32_t m, uint32_t row_words,
                                                            float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lut_s[];
    const float *lut = lut_g;
    if (LUT_IN_LDS) {
        const uint32_t total4 = m * (kCentroids / 4);
        const float4 *src = reinterpret_cast<const float4 *>(lut_g);
        float4 *dst = reinterpret_cast<float4 *>(lut_s);
        for (uint32_t i = threadIdx.x; i < total4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        lut = lut_s;
    }
    const int lane = threadIdx.x & 63;
    const int k = lane & 3;       // SSE lane
    const int rslot = lane >> 2;  // row within the wave step (16 rows)
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t wave = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_block;
    const uint32_t groups = m / 4;
    const float *lut_k = lut + k * kCentroids;
    const uint32_t shift = 8 * k;
    for (uint64_t base = wave * 16; base < n; base += n_waves * 16) {
        const uint64_t idx = base + rslot;
        const uint32_t row = idx < n ? (ids ? ids[idx] : (uint32_t)idx) : 0xFFFFFFFFu;
        const bool ok = row < n_rows;
        const uint32_t *p = rows32 + (uint64_t)(ok ? row : 0) * row_words;
        float acc = 0.0f;
        uint32_t t = 0;
        if (VEC16) {  // row_words % 4 == 0: 16-byte loads, four chunk groups per load
            const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
            for (; t + 4 <= groups; t += 4) {
                const uint4 w = ld_nt(p4 + (t >> 2));
                const float *l = lut_k + (size_t)t * 4 * kCentroids;
                acc += l[(w.x >> shift) & 255u];
                acc += l[4 * kCentroids + ((w.y >> shift) & 255u)];
                acc += l[8 * kCentroids + ((w.z >> shift) & 255u)];
                acc += l[12 * kCentroids + ((w.w >> shift) & 255u)];
            }
        }
        for (; t < groups; t++) {
            const uint32_t w = p[t];
            acc += lut_k[(size_t)t * 4 * kCentroids + ((w >> shift) & 255u)];
        }
        // (l0 + l2) + (l1 + l3)  (:430-432)
        float a = acc + __shfl_xor(acc, 2, 64);
        float s = a + __shfl_xor(a, 1, 64);
        if (k == 0 && idx < n) {
            if (ok) {
                for (uint32_t c = groups * 4; c < m; c++) {  // tail (:434-438)
                    const uint32_t code = (p[c >> 2] >> (8 * (c & 3))) & 255u;
                    s += lut[(size_t)c * kCentroids + code];
                }
                out[idx] = s;
            } else {
                out[idx] = __builtin_nanf("");
            }
        }
    }
}

// Single-launch top-k for small stores (topk.hpp small_topk; structure: u8_topk_small_kernel).  The
// LUT is staged in dynamic LDS once per workgroup, four adjacent lanes own a row and lane k is the
// reference's SSE lane k exactly as in pq_scan_kernel (same summation order => same score bits);
// lane 0 of a row turns the score into a key in the wave's staging row (16 rows per step).
template <bool VEC16>
__global__ __launch_bounds__(1024) void pq_topk_small_kernel(const uint32_t *__restrict__ rows32,
                                                             const float *__restrict__ lut_g, uint32_t n_rows, uint32_t m,
                                                             uint32_t row_words, uint32_t rows_per_wg, SmallTopk p) {
    extern __shared__ __attribute__((aligned(16))) float lut_s[];  // m * 256 f32
    __shared__ unsigned long long lds[2 * kSmallTopkWaves][64];
    unsigned long long(*lists)[64] = lds;
    {
        const uint32_t total4 = m * (kCentroids / 4);
        const float4 *src = reinterpret_cast<const float4 *>(lut_g);
        float4 *dst = reinterpret_cast<float4 *>(lut_s);
        for (uint32_t i = threadIdx.x; i < total4; i += 1024) dst[i] = src[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane & 3, rslot = lane >> 2;
    unsigned long long *stage = lds[kSmallTopkWaves + wave];
    const uint32_t groups = m / 4;
    const float *lut_k = lut_s + k * kCentroids;
    const uint32_t shift = 8 * k;
    const uint64_t wg_base = (uint64_t)blockIdx.x * rows_per_wg;
    SmallTopkWave acc_list;
    for (uint32_t step = wave; step * 16 < rows_per_wg; step += kSmallTopkWaves) {
        const uint32_t local = step * 16 + rslot;
        const uint64_t row = wg_base + local;
        const bool ok = local < rows_per_wg && row < n_rows;
        const uint32_t *pr = rows32 + (ok ? row : 0) * row_words;
        float acc = 0.0f;
        uint32_t t = 0;
        if (VEC16) {
            const uint4 *p4 = reinterpret_cast<const uint4 *>(pr);
            for (; t + 4 <= groups; t += 4) {
                const uint4 w = ld_nt(p4 + (t >> 2));
                const float *l = lut_k + (size_t)t * 4 * kCentroids;
                acc += l[(w.x >> shift) & 255u];
                acc += l[4 * kCentroids + ((w.y >> shift) & 255u)];
                acc += l[8 * kCentroids + ((w.z >> shift) & 255u)];
                acc += l[12 * kCentroids + ((w.w >> shift) & 255u)];
            }
        }
        for (; t < groups; t++) {
            const uint32_t w = pr[t];
            acc += lut_k[(size_t)t * 4 * kCentroids + ((w >> shift) & 255u)];
        }
        const float a = acc + __shfl_xor(acc, 2, 64);  // (l0 + l2) + (l1 + l3)  (:430-432)
        float sc = a + __shfl_xor(a, 1, 64);
        if (k == 0) {
            unsigned long long key = ~0ull;
            if (ok) {
                for (uint32_t c = groups * 4; c < m; c++) {  // tail (:434-438)
                    const uint32_t code = (pr[c >> 2] >> (8 * (c & 3))) & 255u;
                    sc += lut_s[(size_t)c * kCentroids + code];
                }
                key = ((unsigned long long)topk_ordered_bits(sc, p.largest != 0) << 32) | (uint32_t)row;
            }
            stage[acc_list.fill + rslot] = key;
        }
        acc_list.fill += 16;
        if (acc_list.fill == 64) small_topk_flush(acc_list, stage, lane);
    }
    if (acc_list.fill) small_topk_flush(acc_list, stage, lane);
    small_topk_finish(acc_list.best, lists, p);
}

// Fast path for the whole-store scan: m % 16 == 0 (rows are NV = m/16 aligned 16-byte pieces,
// NV <= 8) and the LUT fits in LDS.  Differences to pq_scan_kernel, all measured to matter
// (the first version was latency-bound at one 16-byte load in flight per lane):
//  * the four lanes of a row load DIFFERENT pieces (lane k: pieces k, k+4) and hand the
//    dwords round with DPP quad broadcasts, so a wave-load fetches 1 KiB of distinct bytes
//    instead of 256 B read four times;
//  * UNROLL row tiles are loaded before the first gather (UNROLL*ceil(NV/4) loads in flight
//    per lane) and their gather/add chains interleave (UNROLL independent f32 chains).
// Summation order per row is unchanged: lane k adds chunks k, k+4, ... in order.
template <int J> __device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J | (J << 2) | (J << 4) | (J << 6), 0xF, 0xF, true);
}
template <int J> __device__ __forceinline__ uint4 quad_bcast4(const uint4 &v) {
    return make_uint4(quad_bcast<J>(v.x), quad_bcast<J>(v.y), quad_bcast<J>(v.z), quad_bcast<J>(v.w));
}

// One launch handles a SLICE of NVS 16-byte pieces of every row ([piece0, piece0 + NVS), i.e.
// chunks [16*piece0, ...)) with that slice's LUT in LDS.  A LUT larger than the CU's LDS
// (m > 144; the reference bench uses m = 512) is scanned slice after slice: lane k's running
// sum — the reference's SSE lane sum, order unchanged — is parked per row in `partial`
// between launches (first slice: starts at 0; last slice: combine + tail + output).
// groups_total = m / 4: chunk groups past it (row padding) are skipped, so any m works.
// SIMPLE: the whole row is one slice of full pieces (m % 16 == 0, m <= 144): first == last == 1
// and no padding checks, all resolved at compile time (the m = 96 / m = 16 cases).
template <int NVS, int UNROLL, bool FILTER, bool SIMPLE>
__global__ __launch_bounds__(kScanBlock) void pq_scan_fast_kernel(const uint4 *__restrict__ rows4,
                                                                 uint32_t row_pieces_, uint32_t piece0_,
                                                                 const float *__restrict__ lut_g,
                                                                 uint32_t groups_total_, uint32_t m_, int first_,
                                                                 int last_, float *__restrict__ partial,
                                                                 uint32_t n_rows, float *__restrict__ out,
                                                                 TopkFilter filt) {
    extern __shared__ __attribute__((aligned(16))) float lut_s[];
    const uint32_t row_pieces = SIMPLE ? (uint32_t)NVS : row_pieces_;
    const uint32_t piece0 = SIMPLE ? 0u : piece0_;
    const uint32_t m = SIMPLE ? (uint32_t)NVS * 16u : m_;
    const uint32_t groups_total = SIMPLE ? (uint32_t)NVS * 4u : groups_total_;
    const bool first = SIMPLE ? true : first_ != 0;
    const bool last = SIMPLE ? true : last_ != 0;
    const uint32_t chunk0 = piece0 * 16;
    {
        const uint32_t chunks_here = min(m - chunk0, (uint32_t)NVS * 16u);
        const uint32_t total4 = chunks_here * (kCentroids / 4);
        const float4 *src = reinterpret_cast<const float4 *>(lut_g + (size_t)chunk0 * kCentroids);
        float4 *dst = reinterpret_cast<float4 *>(lut_s);
        for (uint32_t i = threadIdx.x; i < total4; i += kScanBlock) dst[i] = src[i];
        __syncthreads();
    }
    constexpr int JN = (NVS + 3) / 4;
    constexpr int TILE = 16 * UNROLL;
    const int lane = threadIdx.x & 63;
    const int k = lane & 3, rslot = lane >> 2;
    const uint64_t wave = (uint64_t)blockIdx.x * (kScanBlock / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (kScanBlock / 64);
    const float *lut_k = lut_s + k * kCentroids;
    const uint32_t shift = 8 * k;
    const uint32_t group0 = piece0 * 4;
    uint32_t pivot = 0;
    if (FILTER) pivot = *filt.pivot_key;
    for (uint64_t base = wave * TILE; base < n_rows; base += n_waves * TILE) {
        uint4 mine[UNROLL][JN];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint4 *p = rows4 + (base + u * 16 + rslot) * row_pieces + piece0;  // rows are zero-padded
#pragma unroll
            for (int j = 0; j < JN; j++) {
                const int piece = k + 4 * j;
                mine[u][j] = ld_nt(p + (piece < NVS ? piece : NVS - 1));
            }
        }
        float acc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc[u] = first ? 0.0f : partial[(base + u * 16 + rslot) * 4 + k];
#pragma unroll
        for (int pc = 0; pc < NVS; pc++) {
            const float *l = lut_k + pc * 16 * kCentroids;  // piece pc = chunk groups 4pc .. 4pc+3
            const uint32_t g = group0 + 4 * pc;
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const uint4 &held = mine[u][pc / 4];
                uint4 w;
                switch (pc & 3) {
                    case 0: w = quad_bcast4<0>(held); break;
                    case 1: w = quad_bcast4<1>(held); break;
                    case 2: w = quad_bcast4<2>(held); break;
                    default: w = quad_bcast4<3>(held); break;
                }
                if (SIMPLE || g + 3 < groups_total) {  // wave-uniform: a whole piece of real chunks (the common case)
                    acc[u] += l[(w.x >> shift) & 255u];
                    acc[u] += l[4 * kCentroids + ((w.y >> shift) & 255u)];
                    acc[u] += l[8 * kCentroids + ((w.z >> shift) & 255u)];
                    acc[u] += l[12 * kCentroids + ((w.w >> shift) & 255u)];
                } else {  // the row's last, partly padded piece
                    if (g + 0 < groups_total) acc[u] += l[(w.x >> shift) & 255u];
                    if (g + 1 < groups_total) acc[u] += l[4 * kCentroids + ((w.y >> shift) & 255u)];
                    if (g + 2 < groups_total) acc[u] += l[8 * kCentroids + ((w.z >> shift) & 255u)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint64_t row = base + u * 16 + rslot;
            if (!last) {
                partial[row * 4 + k] = acc[u];  // padded like rows
                continue;
            }
            // (l0 + l2) + (l1 + l3)  (:430-432): quad_perm DPP reads, no LDS round trip
            const float a = acc[u] + __int_as_float(__builtin_amdgcn_update_dpp(
                                         0, __float_as_int(acc[u]), 0x4E, 0xF, 0xF, false));  // lane ^ 2
            float sc = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF,
                                                                       false));  // lane ^ 1
            if (k == 0 && row < n_rows) {
                if (groups_total * 4 < m) {  // tail chunks (:434-438); their LUT rows are in this slice
                    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(rows4 + row * row_pieces);
                    for (uint32_t c = groups_total * 4; c < m; c++) {
                        const uint32_t code = (p32[c >> 2] >> (8 * (c & 3))) & 255u;
                        sc += lut_s[(size_t)(c - chunk0) * kCentroids + code];
                    }
                }
                if (FILTER) topk_offer(filt, pivot, sc, (uint32_t)row);
                else out[row] = sc;
            }
        }
    }
}

// The same scan with the LDS gathers free of bank conflicts: rows of 32, 64, 96 or 128 chunks, and - per slice - rows of
// several LUT slices with m % 32 == 0 (SLICED below).  Measurements, ablations and what is still open: profiles/r03_pq_skew.txt.
//
// pq_scan_fast_kernel's 32 lanes of a `ds_read_b32` group read 8 rows' independent codes out of only four chunk tables:
// 3.5 distinct addresses per bank on average, 6.9 LDS cycles per gather against 2 (profiles/r01_pmc_bin_pq_scans.txt).
// Here the LUT sits in LDS TRANSPOSED, [code][chunk]: entry (chunk c, code) is at float code * m + c, so with m % 32 == 0
// its bank is c % 32 whatever the code.  The eight quads of a lane group then only have to be at eight different chunk
// groups at the same moment, which a skew in TIME gives: quad q runs r = 8 - (q & 7) chunk groups behind, lane (q, k)
// reads chunk 4 (u - r) + k at step u, and the 32 lanes of a group hit 32 different banks (SQ_LDS_BANK_CONFLICT = 0).
// A lane's sum still walks chunks k, k + 4, ... of its row in order (the reference's SSE lane sum, :405-440) - the skew
// only moves WHEN it does so.  For that a quad needs its rows as one continuous byte stream: a wave copies each block of
// 16 consecutive rows (coalesced 16-byte loads, registers, `ds_write_b128`) into a two-slot ring of its own in LDS, quad q
// owns row q of every block, and lane (q, k) fetches its code bytes from the ring with `ds_read_u8` at 4 (u - r) + k - no
// cross-lane moves and no bit-field extraction, two vector-ALU operations per gather (address, add).  During the first
// r steps of a row time a quad is still finishing its row of the previous block (the ring's other slot); that slot is
// refilled right after those steps have been issued.  The ring reads are conflict-free too: rows are packed back to back (m / 4 dwords), so the eight
// quads of a group start (m / 4 + 1) q dwords apart, an odd stride.  Result bits are those of pq_scan_fast_kernel.
__global__ __launch_bounds__(kBlock) void pq_lut_transpose_kernel(const float *__restrict__ lut, uint32_t m,
                                                                 float *__restrict__ lut_t) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // = code * m + c
    if (i < m * kCentroids) lut_t[i] = lut[(size_t)(i % m) * kCentroids + i / m];
}

__device__ __forceinline__ uint32_t code_address(uint32_t code, uint32_t pitch, uint32_t base) { return __umul24(code, pitch) + base; }

// LDS: the waves' rings first (so that a lane's LUT base offset 4 k - 16 r, relative to the LUT, is never a negative
// address), then the LUT.  m = 128 fills the CU's 160 KiB exactly with 8 waves (32 KiB of rings + 128 KiB of LUT).
constexpr int skew_waves(uint32_t m) { return m > 96 ? 8 : 16; }
constexpr size_t skew_lds_bytes(uint32_t m) { return (size_t)skew_waves(m) * 32u * m + (size_t)m * kCentroids * 4; }

// SLICED: this launch handles chunks [chunk0, chunk0 + 16 NV) of rows of m_total chunks (a LUT larger than the LDS).  The
// store keeps such rows a second time as a PLANAR scan image (qamd_pq::planar): slice after slice, each a contiguous
// [rows][16 NV] byte array, so a slice launch reads exactly its own bytes, whole 128-byte lines, at the pace of a store
// whose rows ARE that slice (round 3 read them out of row-major rows on a 256-byte pitch: 1.33 x the bytes at m = 192).
// `rows4` is the slice's array.  A lane's sum for a row starts from partial[row][k] instead of 0 (unless first) and goes
// back there (unless last): the reference's SSE lane sum (:405-440), its order unchanged, parked between two launches.
// Several queries of a batch side by side (the filter pass of topk_batch): workgroup b sits on XCD b % 8, its slot b / 8 there
// gives query slot % n and row stream slot / n; the n workgroups of a stream - one per query, each with its own table in
// LDS - walk the same rows in the same order, so the codes leave HBM once per n queries and the others find them in
// that XCD's L2.  n <= 1: one query, every workgroup its own rows (the single-query scan).
struct SkewBatch {
    uint32_t n;              // queries in this launch (2, 4 or 8; 0 / 1: not a batch)
    uint32_t lut_stride;     // floats between two queries' [code][chunk] tables
    uint64_t partial_stride; // floats between two queries' lane-sum buffers (rows of several slices)
    TopkFilterSlices fsl;    // the queries' filter states
};
struct SkewSlice {
    uint32_t chunk0, m_total;
    int first, last;
    float *partial;
    uint32_t run_shift;  // a wave's blocks come in runs of 1 << run_shift consecutive ones (0 or 2; COAL: 2)
};

// R > 1: a ring row is R consecutive store rows (M = R m bytes, m % 4 == 0, m / 4 >= 8) and the LUT's columns repeat R
// times in LDS.  That is the conflict-free scan for rows whose length is not a multiple of 32 chunks: the eight quads of a
// lane group sit at eight consecutive chunk groups of their byte stream, and the banks those hit are distinct only if the
// stream's period in chunk groups is a multiple of 8 - R m / 4 is, m / 4 need not be (m = 48: R = 2, the m = 96 shape).  A
// quad then finishes a store row every m / 4 steps instead of every M / 4.  Per row nothing changes: lane k adds chunks
// k, k + 4, ... of ITS row in order, (l0 + l2) + (l1 + l3).
//
// Which blocks a wave takes, and how the scores leave (round 4, tools/experiments/pq_skew_probe.hip, write_mix_probe.hip):
// with its arithmetic taken out the kernel is a stream - 16 M bytes in, 64 bytes of scores out per block - and that
// stream, not the gathers, sets its time (m = 96, 10M rows, warm clocks: the stream alone 0.172 ms, the whole kernel 0.187,
// its LDS loop alone 0.132).  40 MB of scores cost as much as ~180 MB of reads however they are written, but least as whole
// lines: a wave takes RUNS of four consecutive blocks (run_shift = 2; 6 KiB of codes in a row), lane k of a quad keeps the
// score of the run's block k, and the run's 64 scores leave as one 256-byte nt store (COAL; 0.187 -> 0.176 ms) instead of
// four 64-byte pieces that each only fill half a line.  Small stores keep single blocks (run_shift = 0) so that every wave
// has work.
// PAD > 0 (rows of M - PAD bytes: m = 80, 112, and every m % 4 == 0 whose 16-byte-padded row is 48, 80 or 112 bytes): the ring rows are M bytes all the same, the row's PAD missing chunks get table
// columns of +0.0 - whatever byte lies in the ring's padding is a valid code, and a lane sum that starts at +0.0 never is -0.0,
// so the padded steps leave every sum bit for bit what it was.  The padded steps are executed (96 for 80, 128 for 112), but
// the kernel's time is its stream's (above), and that carries the row's own m bytes only.
template <int NV, int R, bool FILTER, bool SLICED, bool COAL, int PAD = 0>
__global__ __launch_bounds__(64 * skew_waves(16 * NV)) void pq_scan_skew_kernel(const uint4 *__restrict__ rows4,
                                                                 const float *__restrict__ lut_t_g, uint32_t n_rows,
                                                                 float *__restrict__ out, TopkFilter filt, SkewSlice sl, SkewBatch bt) {
    constexpr int M = 16 * NV, S = 4 * NV;
    uint32_t wg = blockIdx.x, n_wg = gridDim.x;  // this workgroup's place among those that share the rows
    if (bt.n > 1) {
        const uint32_t slot = blockIdx.x / 8u, streams = (gridDim.x / 8u) / bt.n, qi = slot % bt.n, stream_local = slot / bt.n;
        if (stream_local >= streams) return;
        wg = (blockIdx.x % 8u) * streams + stream_local;
        n_wg = 8u * streams;
        lut_t_g += (size_t)qi * bt.lut_stride;
        if (FILTER) filt = topk_filter_of(bt.fsl, qi);
        if (SLICED) sl.partial += (size_t)qi * bt.partial_stride;
    }
    constexpr int MR = M / R, SR = S / R;  // chunks / chunk groups of one store row
    constexpr int D = 4;  // blocks of codes in flight per wave (registers)
    static_assert(M % 32 == 0 && S >= 8, "shape");
    static_assert(M % R == 0 && MR % 4 == 0 && (SR >= 8 || (SR == 4 && R == 2)) && (R == 1 || !SLICED), "rows per ring row");
    static_assert(!COAL || (R == 1 && !FILTER && D == 4), "the coalesced score store: one store row per ring row, runs of D = 4 blocks");
    static_assert(PAD == 0 || (PAD == 16 && R == 1 && !SLICED && M >= 64), "padded rows: one 16-chunk piece short of the ring row");
    constexpr int MROW = M - PAD;  // bytes of a store row
    // SR = 4 (m = 16, two rows per 32-chunk ring row): the eight lags span TWO store rows - quads 1..4 finish a store row at
    // steps 0..3 (mod 4), quads 5..8 the store row before it at the same steps
    constexpr bool kTwoGen = SR < 8;
    constexpr int kWaves = skew_waves(M), kThreads = 64 * kWaves;
    constexpr uint32_t kSlot = 16u * M, kStage0 = 0, kLut0 = kWaves * 2u * kSlot;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    {
        float4 *dst = reinterpret_cast<float4 *>(lds_raw + kLut0);
        if (SLICED) {  // the slice's columns of the [code][m_total] table
            for (uint32_t i = threadIdx.x; i < (uint32_t)M * (kCentroids / 4); i += kThreads)
                dst[i] = *reinterpret_cast<const float4 *>(lut_t_g + (size_t)(i / (M / 4)) * sl.m_total + sl.chunk0 + 4u * (i % (M / 4)));
        } else if (R == 1 && (PAD || (sl.m_total && sl.m_total != (uint32_t)M))) {
            // [code][m] in memory, m = sl.m_total chunks (a multiple of 4, <= MROW): the ring row's other columns - the PAD ones and
            // those of the row's own padding to whole 16-byte pieces - are +0.0 for every code
            const uint32_t m_real = sl.m_total ? sl.m_total : (uint32_t)MROW;
            for (uint32_t i = threadIdx.x; i < (uint32_t)M * (kCentroids / 4); i += kThreads) {
                const uint32_t code = i / (M / 4), c4 = i % (M / 4);
                dst[i] = 4u * c4 < m_real ? *reinterpret_cast<const float4 *>(lut_t_g + (size_t)code * m_real + 4u * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (R > 1) {  // [code][MR] in memory, every row of it R times in LDS
            for (uint32_t i = threadIdx.x; i < (uint32_t)M * (kCentroids / 4); i += kThreads)
                dst[i] = *reinterpret_cast<const float4 *>(lut_t_g + (size_t)(i / (M / 4)) * MR + 4u * ((i % (M / 4)) % (MR / 4)));
        } else {
            const float4 *src = reinterpret_cast<const float4 *>(lut_t_g);
            for (uint32_t i = threadIdx.x; i < (uint32_t)M * (kCentroids / 4); i += kThreads) dst[i] = src[i];
        }
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = lane & 3, q = lane >> 2, r = 8u - (q & 7u);
    const uint32_t gw = wg * kWaves + (threadIdx.x >> 6);
    const uint32_t n_waves = n_wg * kWaves;
    const uint32_t n_blocks = ((n_rows + R - 1) / R + 15) / 16;  // blocks of 16 ring rows
    // this wave: the runs gw, gw + n_waves, ... of 1 << rs consecutive blocks; its j-th block is block_of(j)
    const uint32_t rs = COAL ? 2u : sl.run_shift, n_runs = (n_blocks + (1u << rs) - 1u) >> rs;
    if (gw >= n_runs) return;
    const uint32_t J = ((n_runs - gw + n_waves - 1) / n_waves) << rs;  // (blocks past the last one: read as the last, never stored)
    auto block_of = [&](uint32_t j) { return (gw << rs) + (j & ((1u << rs) - 1u)) + (((j >> rs) * n_waves) << rs); };
    const uint32_t stage = kStage0 + (threadIdx.x >> 6) * 2u * kSlot;
    // LUT byte offset of chunk 4 (u - r) + k without the code (step u adds 16 u as an immediate); steps u < r belong to
    // the previous row (chunk group u - r + S): one table row (4 M bytes) further
    const uint32_t off_cur = kLut0 + 4u * k - 16u * r, off_new = off_cur + 4u * M;
    // ring byte address of step u's code: even row times have the current block in slot 0, odd ones in slot 1.  Steps
    // u >= 8 (every quad in its current row): rd8 plus the immediate 4 (u - 8); steps u < 8: a register per step (a
    // lagging quad reads the other slot).  No register holds a negative address (wave 0's ring starts at LDS byte 0).
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
    uint32_t pivot = 0;
    if (FILTER) pivot = *filt.pivot_key;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, FILTER ? 0 : n_rows * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t partial_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(SLICED ? sl.partial : nullptr, 0, SLICED ? n_rows * 16u : 0, 0x00020000);

    // A block is 16 M contiguous bytes: 1024 of them go as one 16-byte piece per lane, the other 512 (M = 32: the only
    // 512, M = 96: the second round) as 8 bytes per lane - every lane loads and writes in every round, so the refill has no
    // branch (a branch would cut the loop body into basic blocks, and a code byte that crosses one is masked to 8 bits again)
    // (PAD: a block is MROW 16-byte pieces - 80 or 112: lane l takes pieces l and 64 + l, the latter clamped to the last one, and
    // writes piece p to ring row p / (MROW / 16), column p % (MROW / 16))
    constexpr int kWide = PAD ? 2 : 16 * M / 1024;             // rounds of 16 bytes per lane: 0 (M = 32), 1, 1, 2 (M = 128)
    constexpr bool kHalf = !PAD && (16 * M) % 1024 != 0;       // one more round of 8 bytes per lane (M = 32, 96)
    const uint32_t pad_p0 = min(lane, (uint32_t)MROW - 1u), pad_p1 = min(64u + lane, (uint32_t)MROW - 1u);
    const uint32_t pad_a0 = (pad_p0 / (MROW / 16)) * M + (pad_p0 % (MROW / 16)) * 16u, pad_a1 = (pad_p1 / (MROW / 16)) * M + (pad_p1 % (MROW / 16)) * 16u;
    struct Held {
        uint4 wide[kWide > 0 ? kWide : 1];
        uint2 half;
        float init;  // SLICED: the lane's sum so far for its row of the block
    };
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(gw), n_waves_u = __builtin_amdgcn_readfirstlane(n_waves);
    const uint32_t J_u = __builtin_amdgcn_readfirstlane(J), rs_u = __builtin_amdgcn_readfirstlane(rs);
    const uint32_t n_blocks_u = __builtin_amdgcn_readfirstlane(n_blocks);
    const uint8_t *rows_b = reinterpret_cast<const uint8_t *>(rows4);
    auto request = [&](Held &h, uint32_t j) {  // block j of this wave (past the end: its last block again, unused)
        const uint32_t jc = j < J_u ? j : J_u - 1;  // wave-uniform, like everything up to `p`
        const uint32_t b_run = (wave_u << rs_u) + (jc & ((1u << rs_u) - 1u)) + (((jc >> rs_u) * n_waves_u) << rs_u);
        const uint32_t blk = b_run < n_blocks_u ? b_run : n_blocks_u - 1u;
        const uint8_t *p = rows_b + (size_t)blk * 16u * MROW;  // (SLICED: `rows4` is the slice's own [rows][M] array)
        if (PAD) {
            h.wide[0] = ld_nt(reinterpret_cast<const uint4 *>(p) + pad_p0);
            h.wide[kWide > 1 ? 1 : 0] = ld_nt(reinterpret_cast<const uint4 *>(p) + pad_p1);
        } else {
#pragma unroll
            for (int i = 0; i < kWide; i++) h.wide[i] = ld_nt(reinterpret_cast<const uint4 *>(p + 1024 * i) + lane);
        }
        if (kHalf) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p + 1024u * kWide + 8u * lane));
            h.half = make_uint2(t.x, t.y);
        }
        h.init = 0.0f;
        if (SLICED && !sl.first) h.init = sl.partial[((size_t)blk * 16u + q) * 4u + k];
    };
    auto refill = [&](const Held &h, uint32_t slot) {
        uint8_t *d = lds_raw + stage + slot * kSlot;
        if (PAD) {
            *reinterpret_cast<uint4 *>(d + pad_a0) = h.wide[0];
            *reinterpret_cast<uint4 *>(d + pad_a1) = h.wide[kWide > 1 ? 1 : 0];
            return;
        }
#pragma unroll
        for (int i = 0; i < kWide; i++) *reinterpret_cast<uint4 *>(d + 1024u * i + 16u * lane) = h.wide[i];
        if (kHalf) *reinterpret_cast<uint2 *>(d + 1024u * kWide + 8u * lane) = h.half;
    };
    Held buf[D];  // buf[j % D] = block j
    float init_of[2];  // the initial sums of the blocks in the ring's slots
#pragma unroll
    for (int j = 0; j < D; j++) request(buf[j], j);
    refill(buf[0], 0);
    init_of[0] = buf[0].init;
    init_of[1] = 0.0f;
    request(buf[0], D);
    // A step is two dependent LDS round trips (code byte, table entry) and an add.  Steps go in groups of eight through a
    // three-stage pipeline kept in this order by scheduling barriers - A(G + 1): ask for the next group's code bytes,
    // C(G - 1): add the previous group's table entries, B(G): addresses and gathers of this group - so a wave always has
    // reads in flight while it computes (left to itself the compiler emits read, wait, compute, read, wait, ...).
    constexpr int GR = S / 8, NG = D * GR;  // groups per row time; groups per trip of the loop (D row times)
    auto ring_addr = [&](int jj, int u) {  // byte address of step u's code in row time jj
        return u < 8 ? ((jj & 1) ? rd_odd[u] : rd_even[u]) : rd8 + (uint32_t)(4 * (u - 8)) + ((jj & 1) ? kSlot : 0u);
    };
    uint32_t codes[2][8];
    float vals[2][8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        codes[0][e] = lds_raw[ring_addr(0, e)];  // A(0)
        vals[1][e] = 0.0f;
    }
    float acc = 0.0f, done = 0.0f, keep = 0.0f;
    // All of the prologue's loads have landed before the loop starts (once per wave).  The compiler places the loop's
    // `s_waitcnt vmcnt(n)` for the state of BOTH ways into the loop header, and it orders the prologue's loads its own way
    // (all 8-byte rounds first): coming from there a buffer had 3 younger loads, so the loop waited with vmcnt(3) - for
    // all but the last block asked for, one block of prefetch instead of four (measured: no overlap of loads and compute).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    for (uint32_t j0 = 0; j0 < J + 2; j0 += D) {
#pragma unroll
        for (int G = 0; G < NG; G++) {
            const int jj = G / GR, g = G % GR;
            float init_new = 0.0f;
            if (g == 0) {  // the slot of block j - 1 is free (its last reads went out with A of this group): block j + 1
                           // goes there, and its registers take block j + 1 + D
                refill(buf[(jj + 1) % D], (jj + 1) & 1);
                init_new = buf[(jj + 1) % D].init;
                request(buf[(jj + 1) % D], j0 + jj + 1 + D);
            }
            {  // A(G + 1)
                const int Gn = (G + 1) % NG, jjn = Gn / GR, gn = Gn % GR;
#pragma unroll
                for (int e = 0; e < 8; e++) codes[(G + 1) & 1][e] = lds_raw[ring_addr(jjn, 8 * gn + e)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; e++) {  // B(G)
                const int u = 8 * g + e;
                const uint32_t base = u < 8 ? lut_base[u] : off_cur;
                                vals[G & 1][e] = *reinterpret_cast<const float *>(lds_raw + code_address(codes[G & 1][e], 4u * M, base) + 16 * u);
            }
            __builtin_amdgcn_sched_barrier(0);
            {  // C(G - 1)
                const int Gp = (G + NG - 1) % NG, jjp = Gp / GR, gp = Gp % GR;
                const uint32_t jp = G == 0 ? j0 - 1u : j0 + (uint32_t)jjp;  // its row time
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    acc += vals[(G + 1) & 1][e];
                    const int u = 8 * gp + e, w = u % SR, sub = u / SR;  // step w of store row `sub` for a quad without lag
                    if (w < 8) {  // quads r = w + 1 (mod SR): that was the last chunk group of their previous store row
                        const bool fin = kTwoGen ? (int)((r - 1u) & (uint32_t)(SR - 1)) == w : (int)r == w + 1;
                        done = fin ? acc : done;
                        acc = fin ? (SLICED ? init_of[jjp & 1] : 0.0f) : acc;  // their next store row starts
                    }
                    if (w == 7 % SR) {  // every quad's previous store row is complete:  (l0 + l2) + (l1 + l3)  (:430-432)
                        const float a = done + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(done), 0x4E, 0xF, 0xF, false));
                        const float sc = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF, false));
                        // sub == 0: the last store row of the ring row of block jp - 1; else store row sub - 1 of block jp's
                        uint32_t jb = sub == 0 ? jp - 1u : jp, srow = (uint32_t)(sub == 0 ? R - 1 : sub - 1);
                        bool started = true;
                        if (kTwoGen) {  // store row F = R jp + sub - 1 of the quad's stream, one earlier for the quads 5..8
                            const int F = (int)R * (int)jp + sub - 1 - (r > (uint32_t)SR ? 1 : 0);
                            started = F >= 0;
                            jb = (uint32_t)(F >> 1), srow = (uint32_t)(F & 1);  // (R == 2)
                        }
                        const uint32_t row = (block_of(jb) * 16u + q) * (uint32_t)R + srow;
                        const bool live = started && jb < J && row < n_rows;  // (jb = -1, -2 as unsigned: the pipeline's first trips)
                        if (SLICED && !sl.last) {  // the lane sums go back to `partial` for the next slice
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(done), partial_rsrc, live ? (row * 4u + k) * 4u : 0xFFFFFFFFu, 0, 0);
                        } else if (FILTER) {
                            if (k == 0 && live) topk_offer(filt, pivot, sc, row);
                        } else if (COAL) {
                            // jb & 3 is a constant of the unrolled loop (D = 4, j0 % 4 == 0): lane k keeps block k's score, and
                            // after the run's last block the quads' 64 scores leave as one 256-byte store
                            const int c = G == 0 ? 2 : ((jjp + 3) & 3);
                            keep = (int)k == c ? sc : keep;
                            if (c == 3) {
                                const uint32_t jbk = jb - 3u + k;
                                const uint32_t row_k = block_of(jbk) * 16u + q;
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(keep), out_rsrc,
                                                                      jbk < J && row_k < n_rows ? row_k * 4u : 0xFFFFFFFFu, 0, 2 /* nt */);
                            }
                        } else {
                            // all four lanes of the quad hold the same bits (f32 addition commutes): they store the same word;
                            // a buffer store drops the lanes whose offset is out of range, so there is no branch here either
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sc), out_rsrc, live ? row * 4u : 0xFFFFFFFFu, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0) init_of[(jj + 1) & 1] = init_new;  // (after C: with one group per row time C still used the slot's old value)
        }
    }
}

// ------------------------------------------------------------------------------ encode
// encode_vector (:237-265): a thread owns one (row, chunk) and walks the 256 centroids in
// index order with the reference's strict '<', so ties and NaNs resolve identically.  The
// chunk's 256 sub-centroids sit in LDS and are read as wave-wide broadcasts.
// Workgroup = 256 rows x all chunks of one chunk-slice (blockIdx.y).
__global__ __launch_bounds__(kBlock) void pq_encode_kernel(const float *__restrict__ data, uint64_t n_rows,
                                                          uint32_t dim, uint32_t chunk_size, uint32_t m,
                                                          const float *__restrict__ centroids /*[256][dim]*/,
                                                          uint8_t *__restrict__ rows, uint32_t row_stride,
                                                          uint64_t row0, uint32_t chunks_per_slice) {
    extern __shared__ __attribute__((aligned(16))) float cen_s[];  // [256][chunk_size]
    const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = r < n_rows;
    const float *src = data + (active ? r : 0) * dim;
    const uint32_t c_begin = blockIdx.y * chunks_per_slice;
    const uint32_t c_end = min(m, c_begin + chunks_per_slice);
    for (uint32_t c = c_begin; c < c_end; c++) {
        const uint32_t lo = c * chunk_size;
        const uint32_t len = min(chunk_size, dim - lo);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < kCentroids * len; i += kBlock) {
            const uint32_t kc = i / len, j = i - kc * len;
            cen_s[kc * chunk_size + j] = centroids[(size_t)kc * dim + lo + j];
        }
        __syncthreads();
        if (!active) continue;
        float min_d = 3.40282347e+38f;
        uint32_t min_i = 0;
        for (uint32_t kc = 0; kc < (uint32_t)kCentroids; kc++) {
            const float *cen = cen_s + kc * chunk_size;
            float d = 0.0f;
            for (uint32_t j = 0; j < len; j++) {
                const float t = src[lo + j] - cen[j];
                d += t * t;  // (a-b).powi(2), sequential f32 sum
            }
            if (d < min_d) {
                min_d = d;
                min_i = kc;
            }
        }
        rows[(row0 + r) * row_stride + c] = (uint8_t)min_i;
    }
}

// Compile-time chunk size (dim % CS == 0).  One thread per row; its sub-vector lives in
// registers.  Centroid values are the same for every lane, so they come through the SCALAR data
// path: the table is re-laid once per encode as [chunk][pair][j] = (c_2p[j], c_2p+1[j]) and the
// pair loop reads it with wave-uniform addresses (s_load_dwordx16 into SGPRs, no LDS, no
// barrier).  Two centroids are evaluated per step as the two halves of packed-f32 instructions
// (v_pk_add_f32 / v_pk_mul_f32, the SGPR pair as one operand).  Each half is the same IEEE
// sub, mul, add chain as pq_encode_kernel (no contraction), compared in index order with the
// same strict '<'.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(kBlock) void pq_pair_table_kernel(const float *__restrict__ centroids, uint32_t dim,
                                                              uint32_t cs, uint32_t m, float *__restrict__ table) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // over m * 256 * cs
    if (i >= m * kCentroids * cs) return;
    const uint32_t half = i & 1, j = (i >> 1) % cs, pr = (i / (2 * cs)) % (kCentroids / 2), c = i / (kCentroids * cs);
    table[i] = centroids[(size_t)(2 * pr + half) * dim + c * cs + j];
}
template <int CS>
__global__ __launch_bounds__(kBlock) void pq_encode_cs_kernel(const float *__restrict__ data, uint64_t n_rows,
                                                             uint32_t dim, uint32_t m,
                                                             const f32x2 *__restrict__ pair_table,
                                                             uint8_t *__restrict__ rows, uint32_t row_stride,
                                                             uint64_t row0, uint32_t chunks_per_slice) {
    const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = r < n_rows;
    const float *src = data + (active ? r : 0) * dim;
    const uint32_t c_begin = blockIdx.y * chunks_per_slice;
    const uint32_t c_end = min(m, c_begin + chunks_per_slice);
    for (uint32_t c = c_begin; c < c_end; c++) {
        const uint32_t lo = c * CS;
        float a[CS];
#pragma unroll
        for (int j = 0; j < CS; j++) a[j] = src[lo + j];
        // U pairs (32 floats; 64 for CS = 32) per step, software-pipelined one step ahead: SMEM
        // returns out of order, so every wait is lgkmcnt(0) -- issuing the next step's s_load
        // before this step's arithmetic is what hides its latency.
        constexpr int U = CS >= 16 ? 1 : 16 / CS;
        constexpr int STEPS = kCentroids / 2 / U;
        struct Step { f32x2 v[U * CS]; };
        const Step *steps = reinterpret_cast<const Step *>(pair_table + (size_t)c * (kCentroids / 2) * CS);
        float min_d = 3.40282347e+38f;
        uint32_t min_i = 0;
        Step cur = steps[0];
        for (uint32_t st = 0; st < (uint32_t)STEPS; st++) {
            const Step nxt = steps[st + 1 < (uint32_t)STEPS ? st + 1 : st];
#pragma unroll
            for (int u = 0; u < U; u++) {
                f32x2 d = {0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < CS; j++) {
                    const f32x2 aa = {a[j], a[j]};
                    const f32x2 t = aa - cur.v[u * CS + j];
                    d += t * t;
                }
                const uint32_t pr = st * U + u;
                if (d.x < min_d) {
                    min_d = d.x;
                    min_i = 2 * pr;
                }
                if (d.y < min_d) {
                    min_d = d.y;
                    min_i = 2 * pr + 1;
                }
            }
            cur = nxt;
        }
        if (active) rows[(row0 + r) * row_stride + c] = (uint8_t)min_i;
    }
}

// encode_query (:525-547): one thread per LUT entry, sequential f32 over the chunk
// (DistanceType::distance, encoded_vectors.rs:37-45).
__global__ __launch_bounds__(kBlock) void pq_lut_kernel(const float *__restrict__ query, uint32_t dim,
                                                       uint32_t chunk_size, uint32_t m,
                                                       const float *__restrict__ centroids, int distance,
                                                       int invert, float *__restrict__ lut, float *__restrict__ lut_t) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m * kCentroids) return;
    const uint32_t c = i / kCentroids, kc = i % kCentroids;
    const uint32_t lo = c * chunk_size, len = min(chunk_size, dim - lo);
    const float *a = query + lo, *b = centroids + (size_t)kc * dim + lo;
    float s = 0.0f;
    if (distance == QAMD_DOT)
        for (uint32_t j = 0; j < len; j++) s += a[j] * b[j];
    else if (distance == QAMD_L1)
        for (uint32_t j = 0; j < len; j++) s += fabsf(a[j] - b[j]);
    else
        for (uint32_t j = 0; j < len; j++) s += (a[j] - b[j]) * (a[j] - b[j]);
    lut[i] = invert ? -s : s;
    if (lut_t) lut_t[kc * m + c] = invert ? -s : s;  // [code][chunk] for pq_scan_skew_kernel
}

// encode_query for a block of queries: blockIdx.y = query, same arithmetic as pq_lut_kernel.
__global__ __launch_bounds__(kBlock) void pq_lut_batch_kernel(const float *__restrict__ queries, uint32_t dim,
                                                             uint32_t chunk_size, uint32_t m,
                                                             const float *__restrict__ centroids, int distance,
                                                             int invert, float *__restrict__ luts, float *__restrict__ luts_t) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m * kCentroids) return;
    const uint32_t c = i / kCentroids, kc = i % kCentroids;
    const uint32_t lo = c * chunk_size, len = min(chunk_size, dim - lo);
    const float *a = queries + (size_t)blockIdx.y * dim + lo, *b = centroids + (size_t)kc * dim + lo;
    float s = 0.0f;
    if (distance == QAMD_DOT)
        for (uint32_t j = 0; j < len; j++) s += a[j] * b[j];
    else if (distance == QAMD_L1)
        for (uint32_t j = 0; j < len; j++) s += fabsf(a[j] - b[j]);
    else
        for (uint32_t j = 0; j < len; j++) s += (a[j] - b[j]) * (a[j] - b[j]);
    luts[(size_t)blockIdx.y * m * kCentroids + i] = invert ? -s : s;
    if (luts_t) luts_t[(size_t)blockIdx.y * m * kCentroids + kc * m + c] = invert ? -s : s;
}

// score_internal (:566-593): decode both rows to centroid sub-vectors, sequential f32.
__global__ void pq_internal_kernel(const uint8_t *__restrict__ rows, uint32_t row_stride, uint32_t dim,
                                   uint32_t chunk_size, uint32_t m, const float *__restrict__ centroids,
                                   int distance, int invert, uint32_t i, uint32_t j, float *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint8_t *ci = rows + (size_t)i * row_stride, *cj = rows + (size_t)j * row_stride;
    float total = 0.0f;
    for (uint32_t c = 0; c < m; c++) {
        const uint32_t lo = c * chunk_size, len = min(chunk_size, dim - lo);
        const float *a = centroids + (size_t)ci[c] * dim + lo, *b = centroids + (size_t)cj[c] * dim + lo;
        float s = 0.0f;
        if (distance == QAMD_DOT)
            for (uint32_t t = 0; t < len; t++) s += a[t] * b[t];
        else if (distance == QAMD_L1)
            for (uint32_t t = 0; t < len; t++) s += fabsf(a[t] - b[t]);
        else
            for (uint32_t t = 0; t < len; t++) s += (a[t] - b[t]) * (a[t] - b[t]);
        total += s;
    }
    *out = invert ? -total : total;
}

// score_point for bursts of (query, id) pairs (lists.hpp): the pairs of list l are scored with LUT l of
// a query batch, gathered through L1/L2 (a burst touches each LUT a few dozen times: staging 96 KiB
// per workgroup would cost more than the gathers).  Four lanes per pair, lane k = the reference's SSE
// lane k, exactly as pq_scan_kernel => the same score bits.
template <bool VEC16>
__global__ __launch_bounds__(kBlock) void pq_lists_kernel(const uint32_t *__restrict__ rows32,
                                                         const float *__restrict__ luts, uint64_t lut_stride,
                                                         const uint32_t *__restrict__ lists, uint32_t n_lists,
                                                         const uint32_t *__restrict__ ids, uint64_t n, uint32_t n_rows,
                                                         uint32_t m, uint32_t row_words, uint32_t pairs_per_block,
                                                         float *__restrict__ out) {
    constexpr int GROUPS = kBlock / 4;
    __shared__ uint32_t first_list;
    const int lane = threadIdx.x & 63;
    const int k = lane & 3, group = threadIdx.x / 4;
    const uint64_t p0 = (uint64_t)blockIdx.x * pairs_per_block;
    const uint64_t p1 = p0 + pairs_per_block < n ? p0 + pairs_per_block : n;
    uint32_t l = first_list_of_block(lists, n_lists, p0, &first_list);  // (lists.hpp)
    const uint32_t groups = m / 4, shift = 8 * k;
    // all four lanes of a pair stay in the loop together (the quad exchanges below): the bound is the pair's
    for (uint64_t idx = p0 + group; idx < p1; idx += GROUPS) {
        const uint32_t row = ids[idx];
        const bool ok = row < n_rows;
        l = advance_list(lists, n_lists, l, idx);
        const float *lut = luts + (size_t)l * lut_stride;
        const float *lut_k = lut + k * kCentroids;
        const uint32_t *p = rows32 + (uint64_t)(ok ? row : 0) * row_words;
        float acc = 0.0f;
        uint32_t t = 0;
        if (VEC16) {
            const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
            for (; t + 4 <= groups; t += 4) {
                const uint4 w = p4[t >> 2];
                const float *q = lut_k + (size_t)t * 4 * kCentroids;
                acc += q[(w.x >> shift) & 255u];
                acc += q[4 * kCentroids + ((w.y >> shift) & 255u)];
                acc += q[8 * kCentroids + ((w.z >> shift) & 255u)];
                acc += q[12 * kCentroids + ((w.w >> shift) & 255u)];
            }
        }
        for (; t < groups; t++) {
            const uint32_t w = p[t];
            acc += lut_k[(size_t)t * 4 * kCentroids + ((w >> shift) & 255u)];
        }
        float a = acc + __shfl_xor(acc, 2, 64);  // (l0 + l2) + (l1 + l3)  (:430-432)
        float sc = a + __shfl_xor(a, 1, 64);
        if (k == 0) {
            if (ok) {
                for (uint32_t c = groups * 4; c < m; c++) {  // tail (:434-438)
                    const uint32_t code = (p[c >> 2] >> (8 * (c & 3))) & 255u;
                    sc += lut[(size_t)c * kCentroids + code];
                }
                out[idx] = sc;
            } else {
                out[idx] = __builtin_nanf("");
            }
        }
    }
}

// The same for bursts whose lists are LONG (hundreds of pairs per query; the reference's PQ bench scores every query against
// random rows, demos/benches/pq.rs:12-46): gathering LUT l through L1/L2 then costs more than the rows themselves - 1M pairs
// in lists of 1024 dragged 96 KiB of LUT per list through the vector caches, 0.25 ms.  Here a 1024-thread workgroup takes
// 1024 consecutive pairs, walks the list segments inside that range, and for every segment of at least kListStageMin pairs
// stages the list's LUT in LDS ONCE (coalesced 16-byte loads) and gathers from there; shorter segments gather through L1/L2
// as above.  A lane group has its up to four pairs' ids and row pieces in flight BEFORE the staging barrier, so the random
// row fetches, the LUT fetch and nothing else stand in a segment's way; the four lanes of a pair load different pieces and
// hand the dwords round the quad (pq_scan_fast_kernel's scheme).  The workgroup's first list is found by all threads
// counting offsets at once (ten dependent loads of a binary search were 5 us in front of every workgroup).  Per pair
// nothing changes: four lanes, lane k = the reference's SSE lane k, (l0 + l2) + (l1 + l3), the tail after it.
constexpr uint32_t kListStageMin = 128, kListStagePairs = kScanBlock;
template <int NP>  // 16-byte pieces per device row (ds / 16)
__global__ __launch_bounds__(kScanBlock) void pq_lists_staged_kernel(const uint4 *__restrict__ rows4,
                                                                    const float *__restrict__ luts, uint64_t lut_stride,
                                                                    const uint32_t *__restrict__ lists, uint32_t n_lists,
                                                                    const uint32_t *__restrict__ ids, uint64_t n, uint32_t n_rows,
                                                                    uint32_t m, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lut_s[];
    __shared__ uint32_t first_list;
    constexpr int JN = (NP + 3) / 4, U = kListStagePairs / (kScanBlock / 4);
    const int lane = threadIdx.x & 63;
    const int k = lane & 3, group = threadIdx.x / 4;
    const uint64_t p0 = (uint64_t)blockIdx.x * kListStagePairs;
    const uint64_t p1 = p0 + kListStagePairs < n ? p0 + kListStagePairs : n;
    // the list of pair p0: the last l with lists[l] <= p0 (empty lists in front of it have the same offset and lose)
    if (threadIdx.x == 0) first_list = 0;
    __syncthreads();
    {
        uint32_t cnt = 0;
        for (uint32_t i = 1 + threadIdx.x; i < n_lists; i += kScanBlock) cnt += lists[i] <= (uint32_t)p0;
        for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (lane == 0 && cnt) atomicAdd(&first_list, cnt);
    }
    __syncthreads();
    uint32_t l = first_list;
    const uint32_t groups_total = m / 4, shift = 8 * k;
    for (uint64_t cur = p0; cur < p1;) {  // one trip per list segment inside [p0, p1): the control flow is workgroup-uniform
        l = advance_list(lists, n_lists, l, cur);
        const uint64_t seg_end = l + 1 < n_lists ? (p1 < lists[l + 1] ? p1 : (uint64_t)lists[l + 1]) : p1;
        const float *lut_g = luts + (size_t)l * lut_stride;
        // this group's pairs of the segment (at most U: a workgroup's range is U passes of its 256 groups)
        uint32_t row[U];
        uint4 mine[U][JN];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t idx = cur + group + (uint64_t)u * (kScanBlock / 4);
            row[u] = idx < seg_end ? ids[idx] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint4 *p = rows4 + (uint64_t)(row[u] < n_rows ? row[u] : 0u) * NP;
#pragma unroll
            for (int j = 0; j < JN; j++) mine[u][j] = p[k + 4 * j < NP ? k + 4 * j : NP - 1];
        }
        const bool staged = seg_end - cur >= kListStageMin;
        if (staged) {
            __syncthreads();  // the previous segment's gathers are done with the LDS image
            const float4 *src = reinterpret_cast<const float4 *>(lut_g);
            float4 *dst = reinterpret_cast<float4 *>(lut_s);
            float4 t[NP];  // m / 16 <= NP rounds, every load in flight before the first LDS store (a load -> store loop is one
                           // memory round trip per round)
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const uint32_t at = threadIdx.x + (uint32_t)i * kScanBlock;
                t[i] = src[at < m * (kCentroids / 4) ? at : 0u];
            }
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const uint32_t at = threadIdx.x + (uint32_t)i * kScanBlock;
                if (at < m * (kCentroids / 4)) dst[at] = t[i];
            }
            __syncthreads();
        }
        auto score = [&](const float *lut) {  // called once with the LDS image and once with the global LUT: two address spaces
            const float *lut_k = lut + k * kCentroids;
            float acc[U];
#pragma unroll
            for (int u = 0; u < U; u++) acc[u] = 0.0f;
#pragma unroll
            for (int pc = 0; pc < NP; pc++) {
                const float *t = lut_k + pc * 16 * kCentroids;  // piece pc = chunk groups 4 pc .. 4 pc + 3
                const uint32_t g = 4 * pc;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint4 &held = mine[u][pc / 4];
                    uint4 w;
                    switch (pc & 3) {
                        case 0: w = quad_bcast4<0>(held); break;
                        case 1: w = quad_bcast4<1>(held); break;
                        case 2: w = quad_bcast4<2>(held); break;
                        default: w = quad_bcast4<3>(held); break;
                    }
                    if (g + 0 < groups_total) acc[u] += t[(w.x >> shift) & 255u];
                    if (g + 1 < groups_total) acc[u] += t[4 * kCentroids + ((w.y >> shift) & 255u)];
                    if (g + 2 < groups_total) acc[u] += t[8 * kCentroids + ((w.z >> shift) & 255u)];
                    if (g + 3 < groups_total) acc[u] += t[12 * kCentroids + ((w.w >> shift) & 255u)];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t idx = cur + group + (uint64_t)u * (kScanBlock / 4);
                const float a = acc[u] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc[u]), 0x4E, 0xF, 0xF, false));
                float sc = a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xF, 0xF, false));
                if (k == 0 && idx < seg_end) {
                    if (row[u] < n_rows) {
                        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(rows4 + (uint64_t)row[u] * NP);
                        for (uint32_t c = groups_total * 4; c < m; c++) {  // tail (:434-438)
                            const uint32_t code = (p32[c >> 2] >> (8 * (c & 3))) & 255u;
                            sc += lut[(size_t)c * kCentroids + code];
                        }
                        out[idx] = sc;
                    } else {
                        out[idx] = __builtin_nanf("");
                    }
                }
            }
        };
        if (staged) score(lut_s);
        else score(lut_g);
        cur = seg_end;
    }
}

// score_internal (:566-593) for bursts of pairs: 16 lanes per pair.  Lane `sub` decodes chunks
// sub, sub + 16, ... of both rows to their centroid sub-vectors and forms the chunk's metric with the
// reference's sequential f32 loop; the chunk results are then added IN CHUNK ORDER (the reference's
// `.sum()` over the chunk iterator), 16 at a time through the group's lanes.  Which row is the "query":
// lists == nullptr -> single_row for every pair; else stored row list_rows[l] for the pairs of list l.
__global__ __launch_bounds__(kBlock) void pq_internal_pairs_kernel(const uint8_t *__restrict__ rows, uint32_t row_stride,
                                                                  uint32_t dim, uint32_t chunk_size, uint32_t m,
                                                                  const float *__restrict__ centroids, int distance,
                                                                  int invert, uint32_t single_row,
                                                                  const uint32_t *__restrict__ lists, uint32_t n_lists,
                                                                  const uint32_t *__restrict__ list_rows,
                                                                  const uint32_t *__restrict__ ids, uint64_t n,
                                                                  uint32_t n_rows, uint32_t pairs_per_block,
                                                                  float *__restrict__ out) {
    constexpr int G = 16, GROUPS = kBlock / G;
    __shared__ uint32_t first_list;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, group = threadIdx.x / G;
    const uint64_t p0 = (uint64_t)blockIdx.x * pairs_per_block;
    const uint64_t p1 = p0 + pairs_per_block < n ? p0 + pairs_per_block : n;
    uint32_t l = lists ? first_list_of_block(lists, n_lists, p0, &first_list) : 0u;  // (lists.hpp)
    for (uint64_t k = p0 + group; k < p1; k += GROUPS) {
        const uint32_t rj = ids[k];
        uint32_t ri = single_row;
        if (lists) {
            l = advance_list(lists, n_lists, l, k);
            ri = list_rows[l];
        }
        const bool ok = rj < n_rows && ri < n_rows;
        const uint8_t *ci = rows + (size_t)(ok ? ri : 0u) * row_stride, *cj = rows + (size_t)(ok ? rj : 0u) * row_stride;
        float total = 0.0f;
        for (uint32_t c0 = 0; c0 < m; c0 += G) {  // group-uniform trip count: the shuffles below involve all 16 lanes
            const uint32_t c = c0 + sub;
            float sc = 0.0f;
            if (c < m) {
                const uint32_t lo = c * chunk_size, len = min(chunk_size, dim - lo);
                const float *a = centroids + (size_t)ci[c] * dim + lo, *b = centroids + (size_t)cj[c] * dim + lo;
                if (distance == QAMD_DOT)
                    for (uint32_t t = 0; t < len; t++) sc += a[t] * b[t];
                else if (distance == QAMD_L1)
                    for (uint32_t t = 0; t < len; t++) sc += fabsf(a[t] - b[t]);
                else
                    for (uint32_t t = 0; t < len; t++) sc += (a[t] - b[t]) * (a[t] - b[t]);
            }
#pragma unroll
            for (int q = 0; q < G; q++) {
                const float v = __shfl(sc, q, G);
                if (c0 + q < m) total += v;
            }
        }
        if (sub == 0) out[k] = ok ? (invert ? -total : total) : __builtin_nanf("");
    }
}

// Reference rows (stride m) <-> device rows (stride round_up(m,4)).
__global__ __launch_bounds__(kBlock) void pq_restride_kernel(const uint8_t *__restrict__ src, uint32_t src_stride,
                                                            uint8_t *__restrict__ dst, uint32_t dst_stride,
                                                            uint64_t n_rows, uint32_t m) {
    const uint64_t total = n_rows * m;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        const uint64_t r = t / m;
        const uint32_t c = (uint32_t)(t - r * m);
        dst[r * dst_stride + c] = src[r * src_stride + c];
    }
}

// ------------------------------------------------------------------------------ k-means (kmeans.rs)
// All chunks iterate in lockstep; a converged chunk (done[c] != 0) is frozen.
// sample: [S][dim]; cen: [256][dim] (centroid-major, like Metadata.centroids).
//
// Deterministic and in the reference's summation order: update_centroids (kmeans.rs:49-137)
// gives each of its `max_threads` workers a contiguous row range (chunk = S / T rows, the last
// worker takes the remainder, :77-82), every worker adds its rows into f64 accumulators in row
// order (:84-93), the partials are merged in worker order (:97-108), and the centroid shift is a
// sequential f32 sum over [centroid][j] (:125-135).  The kernels below do exactly that: rows are
// grouped per (chunk, centroid) in ascending row order, one thread per centroid value walks its
// group with the same partial boundaries, one thread per chunk adds up the shifts.  Given the
// same sample rows and no empty cluster the centroids equal the reference's bit for bit.

// Step 1: per chunk, the rows of every centroid in ascending row order (a stable counting sort).
//   order[c][.]: row ids grouped by centroid; start[c][kc] .. start[c][kc+1]: centroid kc's group.
// The sample's rows are cut into P segments; workgroup (chunk c, segment p) stages its piece of the
// chunk's assignment column in LDS and thread kc scans it -- sixteen rows per 16-byte LDS read, all
// four bytes of a dword compared at once (x ^ kc*0x01010101 has a zero byte exactly where the row is
// assigned to kc).  km_count_kernel counts, km_fill_kernel turns the counts into positions (segment p
// of centroid kc starts after every earlier centroid and after the earlier segments of kc: ascending
// row order is kept) and writes the row ids.  Round 2 had one workgroup per chunk and a byte-wise
// compare: 621 us per iteration at 10 000 x 96 chunks; this form: see DESIGN 3.4.
__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {  // 0x80 in every byte of v that is zero (exact)
    return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
}
constexpr uint32_t kKmMaxSegments = 16;

// LDS image of segment [lo, hi) of column c, as dwords; bytes past `hi` are filled with `fill`.
__device__ __forceinline__ void km_stage_column(const uint8_t *__restrict__ assign, uint32_t m, uint32_t c, uint32_t lo,
                                                uint32_t hi, uint32_t words, uint8_t *col) {
    for (uint32_t i = threadIdx.x; i < words * 4; i += kCentroids) col[i] = lo + i < hi ? assign[(size_t)(lo + i) * m + c] : 0;
    __syncthreads();
}

__global__ __launch_bounds__(kCentroids) void km_count_kernel(const uint8_t *__restrict__ assign /*[S][m]*/, uint32_t S,
                                                             uint32_t m, uint32_t seg_rows, const int *__restrict__ done,
                                                             uint32_t *__restrict__ cnt /*[m][P][256]*/) {
    extern __shared__ __attribute__((aligned(16))) uint8_t col[];
    const uint32_t c = blockIdx.x, p = blockIdx.y, P = gridDim.y, kc = threadIdx.x;
    if (done[c]) return;  // uniform per workgroup
    const uint32_t lo = min(p * seg_rows, S), hi = min(lo + seg_rows, S), n_rows = hi - lo;
    const uint32_t words = (n_rows + 15) / 16 * 4;  // whole 16-byte pieces
    km_stage_column(assign, m, c, lo, hi, words, col);
    const uint32_t kc4 = kc * 0x01010101u;
    const uint4 *col4 = reinterpret_cast<const uint4 *>(col);
    uint32_t n = 0;
    for (uint32_t w = 0; w < words / 4; w++) {
        const uint4 x = col4[w];  // every thread reads the same address: an LDS broadcast
        n += __builtin_popcount(zero_bytes(x.x ^ kc4)) + __builtin_popcount(zero_bytes(x.y ^ kc4)) +
             __builtin_popcount(zero_bytes(x.z ^ kc4)) + __builtin_popcount(zero_bytes(x.w ^ kc4));
    }
    if (kc == 0) n -= words * 4 - n_rows;  // the zero padding past the segment looks like centroid 0
    cnt[((size_t)c * P + p) * kCentroids + kc] = n;
}

__global__ __launch_bounds__(kCentroids) void km_fill_kernel(const uint8_t *__restrict__ assign /*[S][m]*/, uint32_t S,
                                                            uint32_t m, uint32_t seg_rows, const int *__restrict__ done,
                                                            const uint32_t *__restrict__ cnt /*[m][P][256]*/,
                                                            uint32_t *__restrict__ order /*[m][S]*/,
                                                            uint32_t *__restrict__ start /*[m][257]*/) {
    extern __shared__ __attribute__((aligned(16))) uint8_t col[];
    __shared__ uint32_t cum[kCentroids];
    const uint32_t c = blockIdx.x, p = blockIdx.y, P = gridDim.y, kc = threadIdx.x;
    if (done[c]) return;
    uint32_t total = 0, before = 0;  // rows of centroid kc in all segments / in the segments before this one
    for (uint32_t q = 0; q < P; q++) {
        const uint32_t v = cnt[((size_t)c * P + q) * kCentroids + kc];
        total += v;
        if (q < p) before += v;
    }
    cum[kc] = total;
    __syncthreads();
    for (int off = 1; off < kCentroids; off <<= 1) {  // inclusive scan over the centroids
        const uint32_t v = (int)kc >= off ? cum[kc - off] : 0;
        __syncthreads();
        cum[kc] += v;
        __syncthreads();
    }
    const uint32_t first = cum[kc] - total;
    if (p == 0) {
        start[(size_t)c * (kCentroids + 1) + kc] = first;
        if (kc == kCentroids - 1) start[(size_t)c * (kCentroids + 1) + kCentroids] = cum[kc];
    }
    const uint32_t lo = min(p * seg_rows, S), hi = min(lo + seg_rows, S), n_rows = hi - lo;
    const uint32_t words = (n_rows + 15) / 16 * 4;
    km_stage_column(assign, m, c, lo, hi, words, col);
    const uint32_t kc4 = kc * 0x01010101u;
    const uint32_t *col32 = reinterpret_cast<const uint32_t *>(col);
    uint32_t *dst = order + (size_t)c * S;
    uint32_t pos = first + before;
    for (uint32_t w = 0; w < words; w++) {
        uint32_t z = zero_bytes(col32[w] ^ kc4);
        while (z) {  // rare: one row in 256 belongs to kc
            const uint32_t b = (uint32_t)__builtin_ctz(z) >> 3;
            z &= z - 1;
            const uint32_t row = lo + 4 * w + b;
            if (row < hi) dst[pos++] = row;
        }
    }
}

// Step 2: one thread per centroid value: f64 sums with the reference's partial boundaries, mean,
// cast to f32 (:110-124), the shift |old - new| kept per value for step 3.
__global__ __launch_bounds__(kBlock) void km_update_kernel(const float *__restrict__ sample, uint32_t S, uint32_t dim,
                                                          uint32_t chunk_size, uint32_t m, uint32_t workers,
                                                          float *__restrict__ cen, const int *__restrict__ done,
                                                          const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ start,
                                                          float *__restrict__ shift /*[256][dim]*/,
                                                          uint32_t *__restrict__ empties, uint32_t iter) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (uint32_t)kCentroids * dim) return;
    const uint32_t kc = i / dim, j = i - kc * dim;
    const uint32_t c = j / chunk_size;
    if (done[c]) return;
    const uint32_t *st = start + (size_t)c * (kCentroids + 1);
    const uint32_t lo = st[kc], hi = st[kc + 1];
    const uint32_t *rows = order + (size_t)c * S;
    float nv;
    if (hi == lo) {
        // kmeans.rs:111-118 takes a thread_rng row; here a fixed hash of (chunk, centroid, iter) --
        // the one step of the reference that cannot be reproduced (oracle: same rule, stated there)
        uint32_t hsh = (c * 2654435761u) ^ (kc * 40503u) ^ (iter * 2246822519u);
        hsh ^= hsh >> 15;
        hsh *= 2246822519u;
        hsh ^= hsh >> 13;
        nv = sample[(size_t)(hsh % S) * dim + j];
        if (j == c * chunk_size) atomicAdd(empties, 1u);
    } else {
        const uint32_t per = S / workers;  // :77
        double acc = 0.0;
        uint32_t p = lo;
        for (uint32_t w = 0; w < workers; w++) {
            const uint32_t row_end = w + 1 == workers ? S : per * (w + 1);
            double part = 0.0;
            while (p < hi && rows[p] < row_end) {
                part += (double)sample[(size_t)rows[p] * dim + j];
                p++;
            }
            acc += part;  // :101-107
        }
        nv = (float)(acc / (double)(hi - lo));
    }
    shift[i] = fabsf(cen[i] - nv);  // :127-133
    cen[i] = nv;
}

// Step 3: the chunk's total shift, a sequential f32 sum over [centroid][j] (:125-135), and the stopping rule
// (:136: done once the shift is below KMEANS_ACCURACY).  One 256-thread workgroup per chunk gathers the
// 256 x len values into LDS in summation order, 4096 at a time; thread 0 adds them up one after the other
// (the order is the reference's; only the loads are parallel).  Round 2: one thread per chunk doing the 2048
// dependent global loads itself, 201 us per iteration.  The kernel also keeps the device-side state of the
// iteration: done[c], and in `progress` the number of chunks still running after this iteration.
constexpr uint32_t kKmShiftTile = 4096;
__global__ __launch_bounds__(kCentroids) void km_shift_sum_kernel(const float *__restrict__ shift, uint32_t dim,
                                                                 uint32_t chunk_size, uint32_t m, int *__restrict__ done,
                                                                 float *__restrict__ diff /*[m]*/, float accuracy,
                                                                 uint32_t *__restrict__ running /* chunks not done, device */) {
    __shared__ __attribute__((aligned(16))) float vals[kKmShiftTile];
    const uint32_t c = blockIdx.x, t = threadIdx.x;
    if (done[c]) return;  // uniform per workgroup
    const uint32_t lo = c * chunk_size, len = min(chunk_size, dim - lo), total = (uint32_t)kCentroids * len;
    float sum = 0.0f;
    for (uint32_t base = 0; base < total; base += kKmShiftTile) {
        const uint32_t n = min(kKmShiftTile, total - base);
        for (uint32_t i = t; i < n; i += kCentroids) {
            const uint32_t v = base + i, kc = v / len, j = v - kc * len;
            vals[i] = shift[(size_t)kc * dim + lo + j];
        }
        __syncthreads();
        if (t == 0) {
            uint32_t i = 0;
            for (; i + 4 <= n; i += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(vals + i);
                sum += x.x;
                sum += x.y;
                sum += x.z;
                sum += x.w;
            }
            for (; i < n; i++) sum += vals[i];
        }
        __syncthreads();
    }
    if (t == 0) {
        diff[c] = sum;
        if (sum < accuracy) {
            done[c] = 1;
            atomicSub(running, 1u);
        }
    }
}

// The k-means sample: n_out evenly strided rows of a device-resident [count][dim] array.
__global__ __launch_bounds__(kBlock) void km_gather_rows_kernel(const float *__restrict__ data, uint64_t count,
                                                               uint32_t dim, uint64_t n_out, float *__restrict__ out) {
    const uint64_t total = n_out * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t k = i / dim, j = i - k * dim;
        const uint64_t r = (uint64_t)((unsigned __int128)k * count / n_out);
        out[i] = data[r * dim + j];
    }
}

// Streaming form: the sample rows that fall into one batch (rows r_base .. of the whole data).
__global__ __launch_bounds__(kBlock) void km_gather_batch_kernel(const float *__restrict__ batch, uint64_t r_base,
                                                                uint32_t dim, uint64_t count, uint64_t n_out,
                                                                uint64_t k0, uint64_t k1, float *__restrict__ out) {
    const uint64_t total = (k1 - k0) * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t k = k0 + i / dim, j = i % dim;
        const uint64_t r = (uint64_t)((unsigned __int128)k * count / n_out);
        out[k * dim + j] = batch[(r - r_base) * dim + j];
    }
}

int grid_for(uint64_t work_items, uint64_t per_block, int blocks_per_cu) {
    uint64_t want = (work_items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)device_info().cu_count * blocks_per_cu;
    if (want < 1) want = 1;
    return (int)(want > cap ? cap : want);
}

uint64_t chunks_of(uint64_t dim, uint64_t chunk_size) { return (dim + chunk_size - 1) / chunk_size; }

}  // namespace

struct qamd_pq {
    int device = 0;
    qamd_vector_parameters vp{};
    uint64_t chunk_size = 0;
    uint64_t m = 0;           // chunks = reference row bytes
    uint64_t ds = 0;          // device row stride
    uint64_t count = 0;
    std::vector<float> centroids_host;  // [256][dim]
    DevBuf centroids;                   // same on device
    uint32_t kmeans_workers = 1;        // max_kmeans_threads: fixes the f64 summation order (kmeans.rs:77-107)
    uint32_t kmeans_iterations = 0;     // iterations the slowest chunk took (0: centroids were given)
    uint32_t kmeans_empty_clusters = 0; // clusters re-seeded over all iterations (kmeans.rs:111-118)
    DevBuf rows;                        // [padded][ds]
    // Rows of several LUT slices (m > 144) with m % 32 == 0: the scan image - slice after slice, each a contiguous
    // [padded][chunks of the slice] array (slice i starts at byte slice_chunk0[i] * padded) - that pq_scan_skew_kernel<SLICED>
    // reads; `rows` stays the random-access image (score_ids, bursts, score_internal, export).
    DevBuf planar;
    std::vector<uint32_t> slice_chunk0, slice_chunks;
    uint64_t padded = 0;
};

struct qamd_pq_query {
    int device = 0;
    uint64_t m = 0;
    DevBuf lut;  // m*256 f32, chunk-major; when the store takes pq_scan_skew_kernel the [code][chunk] copy follows it
    bool transposed = false;
    const float *lut_t() const { return transposed ? lut.as<float>() + m * kCentroids : nullptr; }
    ReadyEvent ready;  // the last encode_query
};

namespace {

// LUT slices of the planar scan image: 96 chunks where that tiles the row in at most three (m = 192, 288: every launch with
// 16 waves per CU), else 128-chunk slices and a last one of 32 / 64 / 96 / 128 (the reference bench's m = 512: four of 128).
void plan_slices(uint64_t m, std::vector<uint32_t> &chunk0, std::vector<uint32_t> &chunks) {
    chunk0.clear();
    chunks.clear();
    if (valid_pieces(m) <= kMaxSlicePieces || m % 32 != 0) return;
    const uint32_t per = (m % 96 == 0 && m <= 288) ? 96u : 128u;
    for (uint32_t c = 0; c < m; c += per) {
        chunk0.push_back(c);
        chunks.push_back(std::min<uint32_t>(per, (uint32_t)m - c));
    }
}

qamd_status alloc_store(qamd_pq *h) {
    h->m = chunks_of(h->vp.dim, h->chunk_size);
    plan_slices(h->m, h->slice_chunk0, h->slice_chunks);
    if (h->count >= (1ull << 28)) h->slice_chunks.clear();  // (the partial sums leave through 32-bit buffer offsets)
    const bool planar = !h->slice_chunks.empty();
    // rows of 16-byte pieces (the scan's load unit); tiny rows keep whole dwords; rows that pq_scan_fast_kernel scans in
    // slices (m > 144, not a multiple of 32) sit on a 128-byte pitch so that a slice launch reads whole lines
    h->ds = h->m < 16 ? round_up(std::max<uint64_t>(h->m, 1), 4)
          : (valid_pieces(h->m) > kMaxSlicePieces && !planar) ? round_up(h->m, 16 * kSlicePiecesAligned) : round_up(h->m, 16);
    h->padded = round_up(h->count, kRowPad) + kRowPad;
    if (planar) QAMD_TRY(h->planar.alloc(h->padded * h->m, true));
    return h->rows.alloc(h->padded * h->ds, true);
}

// rows [r0, r0 + nr) of the row-major image -> the planar scan image (16-byte pieces; m % 32 == 0 there)
__global__ __launch_bounds__(kBlock) void pq_planar_kernel(const uint4 *__restrict__ rows4, uint32_t row_pieces, uint64_t r0,
                                                          uint64_t nr, uint4 *__restrict__ planar4, uint64_t padded,
                                                          uint32_t piece0, uint32_t slice_pieces) {
    const uint64_t total = nr * slice_pieces;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = r0 + i / slice_pieces;
        const uint32_t pc = (uint32_t)(i % slice_pieces);
        planar4[(uint64_t)piece0 * padded + r * slice_pieces + pc] = rows4[r * row_pieces + piece0 + pc];
    }
}

qamd_status build_planar(qamd_pq *h, uint64_t r0, uint64_t nr, hipStream_t s) {
    if (!h->planar.ptr || nr == 0) return QAMD_OK;
    for (size_t i = 0; i < h->slice_chunks.size(); i++) {
        const uint32_t pieces = h->slice_chunks[i] / 16;
        hipLaunchKernelGGL(pq_planar_kernel, dim3(grid_for(nr * pieces, kBlock * 4, 8)), dim3(kBlock), 0, s, h->rows.as<uint4>(),
                           (uint32_t)(h->ds / 16), r0, nr, h->planar.as<uint4>(), h->padded, h->slice_chunk0[i] / 16, pieces);
    }
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

qamd_status set_centroids(qamd_pq *h, const float *centroids_host, hipStream_t s) {
    const size_t n = (size_t)kCentroids * h->vp.dim;
    h->centroids_host.assign(centroids_host, centroids_host + n);
    QAMD_TRY(h->centroids.alloc(std::max<size_t>(n, 4) * sizeof(float)));
    return copy_in(h->centroids.ptr, h->centroids_host.data(), QAMD_MEM_HOST, n * sizeof(float), s);
}

bool fast_capable(const qamd_pq *h, uint64_t n) {
    return n >= 4096 && h->m >= 16 && n == h->count;
}

// One slice (at most 9 pieces: 144 chunks, 144 KiB of LUT) when the row fits, else line-sized slices (kMaxSlicePieces above).

// pq_scan_skew_kernel: whole rows of 32, 64, 96 or 128 chunks, and of 48 (two rows per ring row), on their natural pitch.
std::atomic<bool> g_skew_unusable{false};  // a device refused the kernel's LDS size: every store goes back to pq_scan_fast_kernel
bool skew_enabled() {
    static const bool on = [] { const char *e = dev_env("QAMD_PQ_SKEW"); return !(e && e[0] == '0'); }();  // developer A/B (tools/lib build): 0 = the older kernel
    if (!on) return false;
    return !g_skew_unusable.load(std::memory_order_relaxed);
}
// Opt in to the instance's dynamic LDS (up to the CU's whole 160 KiB) once per device; false: not available here.
template <int NV, int R, bool FILTER, bool SLICED, int PAD = 0> bool skew_ready() {
    static DeviceOnce once;
    const qamd_status st = once.run([] {
        bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&pq_scan_skew_kernel<NV, R, FILTER, SLICED, false, PAD>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)skew_lds_bytes(16 * NV)) == hipSuccess;
        if constexpr (R == 1 && !FILTER)  // the instance with the coalesced score store
            ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(&pq_scan_skew_kernel<NV, R, FILTER, SLICED, true, PAD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)skew_lds_bytes(16 * NV)) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            g_skew_unusable.store(true, std::memory_order_relaxed);  // (a refusal is final: the set-up counts as done)
        }
        return QAMD_OK;
    });
    return st == QAMD_OK && !g_skew_unusable.load(std::memory_order_relaxed);
}
// Runs of four blocks per wave once every wave has at least four runs of them; below that single blocks keep all waves busy.
inline uint32_t skew_run_shift(uint64_t n_rows, uint32_t rows_per_ring_row, int grid, uint32_t m_ring) {
    static const bool off = [] { const char *e = dev_env("QAMD_PQ_RUNS"); return e && e[0] == '0'; }();  // developer A/B (tools/lib build)
    if (off) return 0u;
    const uint64_t n_blocks = ((n_rows + rows_per_ring_row - 1) / rows_per_ring_row + 15) / 16;
    return n_blocks >= 16ull * (uint64_t)grid * (uint64_t)skew_waves(m_ring) ? 2u : 0u;
}
template <bool FILTER, bool SLICED> bool skew_ready_for(uint32_t nv) {
    switch (nv) {
        case 2: return skew_ready<2, 1, FILTER, SLICED>();
        case 4: return skew_ready<4, 1, FILTER, SLICED>();
        case 6: return skew_ready<6, 1, FILTER, SLICED>();
        case 8: return skew_ready<8, 1, FILTER, SLICED>();
    }
    return false;
}
// rows of several LUT slices with m % 32 == 0: the store holds a planar scan image (alloc_store)
bool skew_sliced_capable(const qamd_pq *h) { return skew_enabled() && h->planar.ptr != nullptr; }
// store rows per ring row: 1 for m = 32 / 64 / 96 / 128, 2 for m = 48 and m = 16; 0: not a shape of the kernel
uint32_t skew_rows_per_ring_row(const qamd_pq *h) {
    // (scores leave through one buffer resource: 32-bit byte offsets)
    if (!skew_enabled() || h->ds != h->m || h->count >= (1ull << 30)) return 0;
    if (h->m % 32 == 0 && h->m <= 128) return 1;
    return (h->m == 48 || h->m == 16) ? 2 : 0;
}
// Rows whose chunks do not fill a ring row of 32 / 64 / 96 / 128: m % 4 == 0 (no tail chunks: the lane sums are the reference's),
// stored on their pitch of whole 16-byte pieces (ds = 32 .. 128); the ring row is the next multiple of 32 bytes (PAD = 0 or 16 more
// than ds) and every table column past m is +0.0 - m = 80 -> 96, 112 -> 128, 120 (dim 960 at chunk 8) -> 128, 100 -> 112 -> 128 ...
// 0: not such a shape (or one the kernel takes as it is); else the ring row's chunks, and *pad its bytes past ds.
uint32_t skew_padded_ring(const qamd_pq *h, uint32_t *pad) {
    *pad = 0;
    if (!skew_enabled() || h->count >= (1ull << 30) || h->m % 4 != 0 || h->ds % 16 != 0 || h->ds < 32 || h->ds > 128) return 0;
    if (h->ds != round_up(h->m, 16)) return 0;
    const uint32_t ring = (uint32_t)round_up(h->ds, 32);
    if (ring == h->m || h->m == 48 || h->m == 16) return 0;  // whole ring rows, or two store rows per ring row
    *pad = ring - (uint32_t)h->ds;
    return ring;
}
uint32_t skew_pad(const qamd_pq *h) {  // (the ring row's chunks; 0: not a padded shape)
    uint32_t pad;
    return skew_padded_ring(h, &pad);
}
bool skew_capable(const qamd_pq *h) { return skew_rows_per_ring_row(h) != 0 || skew_pad(h) != 0; }

// `many` (FILTER only): the launch serves many->n queries side by side (SkewBatch) - only the conflict-free kernel does that;
// *many_done says whether it was launched that way (false: nothing was launched, the caller goes query by query).
template <bool FILTER>
qamd_status launch_fast(const qamd_pq *h, const float *lut_dev, float *out_dev, const TopkFilter *filt,
                        hipStream_t s, const float *lut_t_dev, const SkewBatch *many = nullptr, bool *many_done = nullptr) {
    const uint32_t m = (uint32_t)h->m, pitch_pieces = (uint32_t)(h->ds / 16), pieces = (uint32_t)valid_pieces(m);
    const uint32_t per = pieces <= kMaxSlicePieces ? pieces : kSlicePiecesAligned;
    const uint32_t n_slices = (pieces + per - 1) / per;
    const uint64_t n = h->count;
    SkewBatch batch{};
    if (many) batch = *many;
    if (many_done) *many_done = false;
    // (side by side: all 256 CUs, eight XCDs of 32 - the slot / stream map of the kernel)
    const int grid = many ? 256 : (int)std::min<uint64_t>(device_info().cu_count, (n + 1023) / 1024);
    const int grid_rows = many ? 256 / (int)many->n : grid;  // workgroups that share the rows among themselves
    uint32_t pad_bytes = 0;
    const uint32_t ring_rows = skew_rows_per_ring_row(h), pad = skew_padded_ring(h, &pad_bytes);  // pad: the padded ring row's chunks
    auto padded_ready = [&]() {
        switch (pad / 16 * 100 + pad_bytes) {
            case 200: return skew_ready<2, 1, FILTER, false, 0>();
            case 400: return skew_ready<4, 1, FILTER, false, 0>();
            case 416: return skew_ready<4, 1, FILTER, false, 16>();
            case 600: return skew_ready<6, 1, FILTER, false, 0>();
            case 616: return skew_ready<6, 1, FILTER, false, 16>();
            case 800: return skew_ready<8, 1, FILTER, false, 0>();
            case 816: return skew_ready<8, 1, FILTER, false, 16>();
        }
        return false;
    };
    if (pad ? padded_ready()
            : ring_rows == 2 ? (m == 48 ? skew_ready<6, 2, FILTER, false>() : skew_ready<2, 2, FILTER, false>())
                             : ring_rows == 1 && skew_ready_for<FILTER, false>(m / 16)) {
        // the LUT as [code][chunk]: encode_query leaves that copy behind the chunk-major one; a caller without it pays a
        // transposing launch (96 KiB, L2-resident)
        const float *lut_t = lut_t_dev;
        float *ws = nullptr;
        if (!lut_t && many) return QAMD_OK;  // (a side-by-side launch needs the batch's [code][chunk] tables)
        if (!lut_t) {
            QAMD_TRY(thread_ws_acquire(WS_PARTIAL, (size_t)m * kCentroids * sizeof(float), s, reinterpret_cast<void **>(&ws)));
            hipLaunchKernelGGL(pq_lut_transpose_kernel, dim3((m * kCentroids + kBlock - 1) / kBlock), dim3(kBlock), 0, s, lut_dev, m, ws);
            lut_t = ws;
        }
        const uint32_t run_shift = skew_run_shift(n, pad ? 1 : ring_rows, grid_rows, pad ? pad : ring_rows * m);
        SkewSlice whole{};
        whole.run_shift = run_shift;
        if (pad) whole.m_total = m;  // (the table in memory is [code][m]; the kernel pads it with +0.0 columns)
#define QAMD_PQ_SKEW_AS(NVV, RR, CO, PD)                                                                      \
    hipLaunchKernelGGL((pq_scan_skew_kernel<NVV, RR, FILTER, false, CO, PD>), dim3(grid), dim3(64 * skew_waves(16 * NVV)), \
                       skew_lds_bytes(16 * NVV), s, h->rows.as<uint4>(), lut_t, (uint32_t)n, out_dev,         \
                       filt ? *filt : TopkFilter{}, whole, batch)
#define QAMD_PQ_SKEW_P(NVV, RR, PD)                                                \
    do {                                                                           \
        if constexpr (!FILTER && RR == 1) {                                        \
            if (run_shift == 2) QAMD_PQ_SKEW_AS(NVV, RR, true, PD);                \
            else QAMD_PQ_SKEW_AS(NVV, RR, false, PD);                              \
        } else {                                                                   \
            QAMD_PQ_SKEW_AS(NVV, RR, false, PD);                                   \
        }                                                                          \
    } while (0)
#define QAMD_PQ_SKEW(NVV, RR) QAMD_PQ_SKEW_P(NVV, RR, 0)
        if (pad) switch (pad / 16 * 100 + pad_bytes) {
            case 200: QAMD_PQ_SKEW_P(2, 1, 0); break;
            case 400: QAMD_PQ_SKEW_P(4, 1, 0); break;
            case 416: QAMD_PQ_SKEW_P(4, 1, 16); break;
            case 600: QAMD_PQ_SKEW_P(6, 1, 0); break;
            case 616: QAMD_PQ_SKEW_P(6, 1, 16); break;
            case 800: QAMD_PQ_SKEW_P(8, 1, 0); break;
            case 816: QAMD_PQ_SKEW_P(8, 1, 16); break;
        }
        else if (ring_rows == 2 && m == 48) QAMD_PQ_SKEW(6, 2);
        else if (ring_rows == 2) QAMD_PQ_SKEW(2, 2);
        else switch (m / 16) {
            case 2: QAMD_PQ_SKEW(2, 1); break;
            case 4: QAMD_PQ_SKEW(4, 1); break;
            case 6: QAMD_PQ_SKEW(6, 1); break;
            case 8: QAMD_PQ_SKEW(8, 1); break;
        }
#undef QAMD_PQ_SKEW_AS
#undef QAMD_PQ_SKEW_P
#undef QAMD_PQ_SKEW
        if (ws) thread_ws_release(WS_PARTIAL, s);
        QAMD_HIP(hipGetLastError());
        if (many_done) *many_done = true;
        return QAMD_OK;
    }
    if (lut_t_dev && skew_sliced_capable(h)) {  // rows of several LUT slices: one launch per slice of the planar scan image
        bool ready = true;
        for (uint32_t c : h->slice_chunks) ready = ready && skew_ready_for<FILTER, true>(c / 16);
        if (ready) {
            float *partial = nullptr;
            QAMD_TRY(thread_ws_acquire(WS_PARTIAL, h->padded * 16 * (many ? many->n : 1), s, reinterpret_cast<void **>(&partial)));
            batch.partial_stride = h->padded * 4;  // (floats: every query of a side-by-side launch has its own lane sums)
            const size_t ns = h->slice_chunks.size();
            for (size_t sl = 0; sl < ns; sl++) {
                const uint32_t run_shift = skew_run_shift(n, 1, grid_rows, h->slice_chunks[sl]);
                const SkewSlice ss{h->slice_chunk0[sl], m, sl == 0, sl + 1 == ns, partial, run_shift};
                const uint4 *slice = reinterpret_cast<const uint4 *>(h->planar.as<uint8_t>() + (uint64_t)h->slice_chunk0[sl] * h->padded);
                const bool coal = !FILTER && run_shift == 2 && sl + 1 == ns;  // the last slice writes the scores
#define QAMD_PQ_SKEW_SLICE_AS(NVV, CO)                                                                       \
        hipLaunchKernelGGL((pq_scan_skew_kernel<NVV, 1, FILTER, true, CO>), dim3(grid), dim3(64 * skew_waves(16 * NVV)), \
                           skew_lds_bytes(16 * NVV), s, slice, lut_t_dev, (uint32_t)n, out_dev,              \
                           filt ? *filt : TopkFilter{}, ss, batch)
#define QAMD_PQ_SKEW_SLICE(NVV)                                                                              \
    case NVV:                                                                                               \
        if constexpr (!FILTER) {                                                                            \
            if (coal) QAMD_PQ_SKEW_SLICE_AS(NVV, true);                                                     \
            else QAMD_PQ_SKEW_SLICE_AS(NVV, false);                                                         \
        } else {                                                                                            \
            QAMD_PQ_SKEW_SLICE_AS(NVV, false);                                                              \
        }                                                                                                   \
        break;
                switch (h->slice_chunks[sl] / 16) { QAMD_PQ_SKEW_SLICE(2) QAMD_PQ_SKEW_SLICE(4) QAMD_PQ_SKEW_SLICE(6) QAMD_PQ_SKEW_SLICE(8) }
#undef QAMD_PQ_SKEW_SLICE_AS
#undef QAMD_PQ_SKEW_SLICE
            }
            thread_ws_release(WS_PARTIAL, s);
            QAMD_HIP(hipGetLastError());
            if (many_done) *many_done = true;
            return QAMD_OK;
        }
    }
    if (many) return QAMD_OK;  // (the older kernels take one query at a time: nothing launched)
    float *partial = nullptr;
    if (n_slices > 1) {
        const uint64_t padded = round_up(n, kRowPad) + kRowPad;
        QAMD_TRY(thread_ws_acquire(WS_PARTIAL, padded * 16, s, reinterpret_cast<void **>(&partial)));
    }
    for (uint32_t sl = 0; sl < n_slices; sl++) {
        const uint32_t piece0 = sl * per, nvs = std::min(per, pieces - piece0);
        const int first = sl == 0, last = sl + 1 == n_slices;
        const bool simple = n_slices == 1 && m % 16 == 0;
        const size_t lds = (size_t)std::min<uint32_t>(nvs * 16, m - piece0 * 16) * kCentroids * sizeof(float);
#define QAMD_PQ_FAST(NVV)                                                                                   \
    case NVV: {                                                                                             \
        QAMD_LDS_OPT_IN((&pq_scan_fast_kernel<NVV, 4, FILTER, true>), kLdsBudget);                           \
        QAMD_LDS_OPT_IN((&pq_scan_fast_kernel<NVV, 4, FILTER, false>), kLdsBudget);                          \
        if (simple)                                                                                         \
            hipLaunchKernelGGL((pq_scan_fast_kernel<NVV, 4, FILTER, true>), dim3(grid), dim3(kScanBlock), lds, s, \
                               h->rows.as<uint4>(), pieces, piece0, lut_dev, m / 4, m, first, last, partial, \
                               (uint32_t)n, out_dev, filt ? *filt : TopkFilter{});                          \
        else                                                                                                \
            hipLaunchKernelGGL((pq_scan_fast_kernel<NVV, 4, FILTER, false>), dim3(grid), dim3(kScanBlock), lds, s, \
                               h->rows.as<uint4>(), pitch_pieces, piece0, lut_dev, m / 4, m, first, last, partial, \
                               (uint32_t)n, out_dev, filt ? *filt : TopkFilter{});                          \
        break;                                                                                              \
    }
        switch (nvs) {
            QAMD_PQ_FAST(1) QAMD_PQ_FAST(2) QAMD_PQ_FAST(3) QAMD_PQ_FAST(4) QAMD_PQ_FAST(5)
            QAMD_PQ_FAST(6) QAMD_PQ_FAST(7) QAMD_PQ_FAST(8) QAMD_PQ_FAST(9)
        }
#undef QAMD_PQ_FAST
    }
    if (partial) thread_ws_release(WS_PARTIAL, s);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

qamd_status scan_launch(const qamd_pq *h, const float *lut_dev, const uint32_t *ids_dev, uint64_t n,
                        float *out_dev, hipStream_t s, const TopkFilter *filt = nullptr, const float *lut_t_dev = nullptr) {
    if (n == 0) return QAMD_OK;
    const uint32_t m = (uint32_t)h->m, row_words = (uint32_t)(h->ds / 4);
    const size_t lds = (size_t)m * kCentroids * sizeof(float);
    const bool in_lds = lds <= kLdsBudget && n >= 4096;  // small batches: not worth staging 96 KiB per CU
    const bool vec16 = (row_words % 4) == 0;
    const int cu = device_info().cu_count;
#define QAMD_PQ_LAUNCH(LDSF, V16, GRID, SH)                                                                 \
    hipLaunchKernelGGL((pq_scan_kernel<LDSF, V16>), dim3(GRID), dim3(kScanBlock), SH, s, h->rows.as<uint32_t>(), \
                       lut_dev, ids_dev, n, (uint32_t)h->count, m, row_words, out_dev)
    if (!ids_dev && fast_capable(h, n)) {
        return filt ? launch_fast<true>(h, lut_dev, out_dev, filt, s, lut_t_dev) : launch_fast<false>(h, lut_dev, out_dev, filt, s, lut_t_dev);
    } else if (in_lds) {
        QAMD_LDS_OPT_IN((&pq_scan_kernel<true, true>), kLdsBudget);  // > 64 KiB of dynamic LDS: once per kernel and device
        QAMD_LDS_OPT_IN((&pq_scan_kernel<true, false>), kLdsBudget);
        const int grid = (int)std::min<uint64_t>(cu, (n + 255) / 256);
        if (vec16) QAMD_PQ_LAUNCH(true, true, grid, lds);
        else QAMD_PQ_LAUNCH(true, false, grid, lds);
    } else {
        const int grid = (int)std::min<uint64_t>((uint64_t)cu * 2, (n + 255) / 256);
        if (vec16) QAMD_PQ_LAUNCH(false, true, grid, 0);
        else QAMD_PQ_LAUNCH(false, false, grid, 0);
    }
#undef QAMD_PQ_LAUNCH
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// The single-launch top-k when the store qualifies (<= 2M rows, k <= 64, LUT <= 128 KiB): false = not applicable.
bool pq_topk_small(const qamd_pq *h, const float *lut, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                   qamd_mem out_mem, hipStream_t s, qamd_status &status) {
    SmallTopkPlan plan;
    const size_t lut_bytes = (size_t)h->m * kCentroids * sizeof(float);
    // every workgroup stages the whole LUT (m KiB) before its first row: worth it from ~512 rows on
    if (h->m < 1 || lut_bytes > 128 * 1024 || !small_topk_plan(h->count, k, 16, 512, plan)) return false;
    status = [] {
        QAMD_LDS_OPT_IN((&pq_topk_small_kernel<true>), 128 * 1024);
        QAMD_LDS_OPT_IN((&pq_topk_small_kernel<false>), 128 * 1024);
        return QAMD_OK;
    }();
    if (status != QAMD_OK) return true;
    const uint32_t row_words = (uint32_t)(h->ds / 4);
    status = small_topk(plan, k, largest, out_ids, out_scores, out_mem, s, [&](const SmallTopk &p, hipStream_t st) {
        if (row_words % 4 == 0)
            hipLaunchKernelGGL(pq_topk_small_kernel<true>, dim3(plan.workgroups), dim3(1024), lut_bytes, st,
                               h->rows.as<uint32_t>(), lut, (uint32_t)h->count, (uint32_t)h->m, row_words, plan.rows_per_wg, p);
        else
            hipLaunchKernelGGL(pq_topk_small_kernel<false>, dim3(plan.workgroups), dim3(1024), lut_bytes, st,
                               h->rows.as<uint32_t>(), lut, (uint32_t)h->count, (uint32_t)h->m, row_words, plan.rows_per_wg, p);
        QAMD_HIP(hipGetLastError());
        return QAMD_OK;
    });
    return true;
}

qamd_status check_query(const qamd_pq *h, const qamd_pq_query *q) {
    if (!h || !q) return fail(QAMD_ERR_ARGUMENTS, "null handle or query");
    if (q->m != h->m) return fail(QAMD_ERR_ARGUMENTS, "query LUT has %llu chunks, store has %llu",
                                  (unsigned long long)q->m, (unsigned long long)h->m);
    return QAMD_OK;
}

// Nearest-centroid codes of `nr` device-resident rows: rows_out[(r0 + r) * row_stride + c].
// `pair_table` (built by build_pair_table from the same centroids) selects the scalar-path kernel.
bool cs_fast_shape(uint64_t dim, uint64_t chunk_size) {
    return dim % chunk_size == 0 && (chunk_size == 1 || chunk_size == 2 || chunk_size == 4 || chunk_size == 8 ||
                                     chunk_size == 16 || chunk_size == 32);
}

qamd_status build_pair_table(const float *centroids_dev, uint64_t dim, uint64_t chunk_size, uint64_t m, DevBuf &table,
                             hipStream_t s) {
    const uint32_t n = (uint32_t)(m * kCentroids * chunk_size);
    if (!table.ptr) QAMD_TRY(table.alloc((size_t)n * 4));
    hipLaunchKernelGGL(pq_pair_table_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, centroids_dev,
                       (uint32_t)dim, (uint32_t)chunk_size, (uint32_t)m, table.as<float>());
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

qamd_status launch_assign(const float *src, uint64_t nr, uint64_t dim, uint64_t chunk_size, uint64_t m,
                          const float *centroids_dev, const DevBuf *pair_table, uint8_t *rows_out, uint64_t row_stride,
                          uint64_t r0, hipStream_t s) {
    const size_t lds = (size_t)kCentroids * chunk_size * sizeof(float);
    if (lds > kLdsBudget) return fail(QAMD_ERR_ARGUMENTS, "chunk_size %llu too large", (unsigned long long)chunk_size);
    QAMD_LDS_OPT_IN((&pq_encode_kernel), (int)kLdsBudget);
    const uint32_t gx = (uint32_t)((nr + kBlock - 1) / kBlock);
    // few row blocks -> split the chunk loop over blockIdx.y to fill the chip
    uint32_t slices = 1;
    const uint32_t want = (uint32_t)device_info().cu_count * 4;
    if (gx < want) slices = std::min<uint32_t>((uint32_t)m, (want + gx - 1) / gx);
    const uint32_t per = (uint32_t)((m + slices - 1) / slices);
    slices = (uint32_t)((m + per - 1) / per);
#define QAMD_PQ_ENC(CSV)                                                                                     \
    case CSV:                                                                                               \
        hipLaunchKernelGGL((pq_encode_cs_kernel<CSV>), dim3(gx, slices), dim3(kBlock), 0, s, src, nr, (uint32_t)dim, \
                           (uint32_t)m, pair_table->as<f32x2>(), rows_out, (uint32_t)row_stride, r0, per);   \
        break;
    bool fast = pair_table && pair_table->ptr && cs_fast_shape(dim, chunk_size);
    if (fast) {
        switch (chunk_size) {
            QAMD_PQ_ENC(1) QAMD_PQ_ENC(2) QAMD_PQ_ENC(4) QAMD_PQ_ENC(8) QAMD_PQ_ENC(16) QAMD_PQ_ENC(32)
            default: fast = false;
        }
    }
#undef QAMD_PQ_ENC
    if (!fast)
        hipLaunchKernelGGL(pq_encode_kernel, dim3(gx, slices), dim3(kBlock), lds, s, src, nr, (uint32_t)dim,
                           (uint32_t)chunk_size, (uint32_t)m, centroids_dev, rows_out, (uint32_t)row_stride, r0, per);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

qamd_status encode_rows(qamd_pq *h, const float *data, qamd_mem data_mem, qamd_stop_fn stop, void *stop_user,
                        hipStream_t s) {
    const uint64_t dim = h->vp.dim, count = h->count;
    if (count == 0 || dim == 0) return QAMD_OK;
    // host rows are staged 256 MiB at a time; device rows are read in place, 8 GiB per launch
    const uint64_t batch_bytes = data_mem == QAMD_MEM_HOST ? (256ull << 20) : (8ull << 30);
    const uint64_t batch_rows = std::max<uint64_t>(1, std::min<uint64_t>(count, batch_bytes / (dim * 4)));
    DevBuf stage;
    if (data_mem == QAMD_MEM_HOST) QAMD_TRY(stage.alloc(batch_rows * dim * 4));
    DevBuf pair_table;  // [chunk][centroid pair][j][2], read through the scalar cache
    if (cs_fast_shape(dim, h->chunk_size))
        QAMD_TRY(build_pair_table(h->centroids.as<float>(), dim, h->chunk_size, h->m, pair_table, s));
    for (uint64_t r0 = 0; r0 < count; r0 += batch_rows) {
        if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :198-200
        const uint64_t nr = std::min(batch_rows, count - r0);
        const float *src = data + r0 * dim;
        if (data_mem == QAMD_MEM_HOST) {
            QAMD_TRY(copy_in(stage.ptr, src, QAMD_MEM_HOST, nr * dim * 4, s));
            src = stage.as<float>();
        }
        QAMD_TRY(launch_assign(src, nr, dim, h->chunk_size, h->m, h->centroids.as<float>(), &pair_table,
                               h->rows.as<uint8_t>(), h->ds, r0, s));
        QAMD_TRY(build_planar(h, r0, nr, s));
        if (data_mem == QAMD_MEM_HOST || stop) QAMD_HIP(hipStreamSynchronize(s));
    }
    QAMD_HIP(hipStreamSynchronize(s));
    return QAMD_OK;
}

// kmeans (kmeans.rs:7-47) for every chunk at once on a device-resident sample [S][dim].
// The iteration state (done[c], the number of running chunks) lives on the device; every iteration
// publishes the running count into one word of mapped host memory, and the host reads iteration i's
// word only after it has enqueued iteration i + 1: the GPU never waits for the host (round 2: two
// synchronisations per iteration).  An iteration enqueued after the last chunk has converged finds
// every done[c] set and does nothing, so the centroids are exactly those of the reference's loop.
__global__ void km_publish_kernel(const uint32_t *__restrict__ running, uint32_t *__restrict__ slot) {
    __hip_atomic_store(slot, *running, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

qamd_status train_from_sample(qamd_pq *h, const float *sample, uint32_t S, qamd_stop_fn stop, void *stop_user,
                              hipStream_t s) {
    const uint64_t dim = h->vp.dim;
    const uint32_t m = (uint32_t)h->m;
    const uint32_t workers = std::max<uint32_t>(1, h->kmeans_workers);
    const size_t ncen = (size_t)kCentroids * dim;
    // segments of the sample per chunk for the grouping step: enough workgroups for two per CU, segments of >= 1024 rows
    const uint32_t want_wgs = (uint32_t)device_info().cu_count * 2;
    uint32_t P = std::max<uint32_t>(1, std::min<uint32_t>(kKmMaxSegments, (want_wgs + m - 1) / std::max<uint32_t>(m, 1)));
    P = std::max<uint32_t>(1, std::min<uint32_t>(P, (S + 1023) / 1024));
    const uint32_t seg_rows = (uint32_t)round_up((S + P - 1) / P, 16);
    P = (S + seg_rows - 1) / seg_rows;
    DevBuf cen, shift, done, diff, assign, order, start, counters, cnt, pair_table;
    QAMD_TRY(cen.alloc(ncen * 4));
    QAMD_TRY(shift.alloc(ncen * 4));
    QAMD_TRY(done.alloc((size_t)m * 4, true));
    QAMD_TRY(diff.alloc((size_t)m * 4));
    QAMD_TRY(assign.alloc((size_t)S * m));
    QAMD_TRY(order.alloc((size_t)S * m * 4));
    QAMD_TRY(start.alloc((size_t)m * (kCentroids + 1) * 4));
    QAMD_TRY(cnt.alloc((size_t)m * P * kCentroids * 4));
    QAMD_TRY(counters.alloc(8, true));  // [0] empty-cluster re-seeds, [1] chunks still running
    uint32_t *empties = counters.as<uint32_t>(), *running = counters.as<uint32_t>() + 1;
    QAMD_TRY(copy_in(running, &m, QAMD_MEM_HOST, 4, s));
    // progress words in mapped host memory, one per iteration (0xFFFFFFFF: not published yet)
    struct Mapped {
        uint32_t *host = nullptr, *dev = nullptr;
        ~Mapped() {
            if (host) (void)hipHostFree(host);
        }
    } prog;
    QAMD_HIP(hipHostMalloc(reinterpret_cast<void **>(&prog.host), kKmeansMaxIter * 4, hipHostMallocMapped));
    QAMD_HIP(hipHostGetDevicePointer(reinterpret_cast<void **>(&prog.dev), prog.host, 0));
    memset(prog.host, 0xFF, kKmeansMaxIter * 4);
    struct Events {
        hipEvent_t ev[2] = {nullptr, nullptr};
        ~Events() {
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } evs;
    for (hipEvent_t &e : evs.ev) QAMD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // initial centroids = the first 256 sample rows (kmeans.rs:25)
    QAMD_HIP(hipMemcpyAsync(cen.ptr, sample, ncen * 4, hipMemcpyDeviceToDevice, s));
    const bool cs = cs_fast_shape(dim, h->chunk_size);
    const size_t col_lds = (size_t)round_up(seg_rows, 16);
    h->kmeans_iterations = 0;
    auto iteration_done = [&](int it, bool &all) -> qamd_status {  // waits for iteration `it`, reads its progress word
        QAMD_HIP(hipEventSynchronize(evs.ev[it & 1]));
        const uint32_t left = *const_cast<volatile uint32_t *>(prog.host + it);
        h->kmeans_iterations = (uint32_t)it + 1;
        all = left == 0;
        return QAMD_OK;
    };
    int enqueued = 0;
    bool all = false;
    for (int iter = 0; iter < kKmeansMaxIter && !all; iter++) {
        if (stop && stop(stop_user)) {
            (void)hipStreamSynchronize(s);  // the mapped progress words are freed on return
            return fail(QAMD_ERR_STOPPED, "Stopped");  // kmeans.rs:29-31
        }
        // update_indexes (:139-166) = the PQ encoder's own nearest-centroid kernel on the sample
        if (cs) QAMD_TRY(build_pair_table(cen.as<float>(), dim, h->chunk_size, m, pair_table, s));
        QAMD_TRY(launch_assign(sample, S, dim, h->chunk_size, m, cen.as<float>(), &pair_table, assign.as<uint8_t>(), m,
                               0, s));
        hipLaunchKernelGGL(km_count_kernel, dim3(m, P), dim3(kCentroids), col_lds, s, assign.as<uint8_t>(), S, m, seg_rows,
                           done.as<int>(), cnt.as<uint32_t>());
        hipLaunchKernelGGL(km_fill_kernel, dim3(m, P), dim3(kCentroids), col_lds, s, assign.as<uint8_t>(), S, m, seg_rows,
                           done.as<int>(), cnt.as<uint32_t>(), order.as<uint32_t>(), start.as<uint32_t>());
        hipLaunchKernelGGL(km_update_kernel, dim3((uint32_t)((ncen + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, sample, S,
                           (uint32_t)dim, (uint32_t)h->chunk_size, m, workers, cen.as<float>(), done.as<int>(),
                           order.as<uint32_t>(), start.as<uint32_t>(), shift.as<float>(), empties, (uint32_t)iter);
        hipLaunchKernelGGL(km_shift_sum_kernel, dim3(m), dim3(kCentroids), 0, s, shift.as<float>(), (uint32_t)dim,
                           (uint32_t)h->chunk_size, m, done.as<int>(), diff.as<float>(), kKmeansAccuracy, running);
        hipLaunchKernelGGL(km_publish_kernel, dim3(1), dim3(1), 0, s, running, prog.dev + iter);
        QAMD_HIP(hipGetLastError());
        QAMD_HIP(hipEventRecord(evs.ev[iter & 1], s));
        enqueued = iter + 1;
        if (iter >= 1) QAMD_TRY(iteration_done(iter - 1, all));  // one iteration behind: the GPU is never idle
    }
    if (!all && enqueued) QAMD_TRY(iteration_done(enqueued - 1, all));
    QAMD_HIP(hipStreamSynchronize(s));
    QAMD_TRY(copy_out(&h->kmeans_empty_clusters, QAMD_MEM_HOST, empties, 4, s));
    std::vector<float> cen_h(ncen);
    QAMD_TRY(copy_out(cen_h.data(), QAMD_MEM_HOST, cen.ptr, ncen * 4, s));
    return set_centroids(h, cen_h.data(), s);
}

// find_centroids (:278-342) for count > 256: k-means on a <= 10 000-row sample.
qamd_status train_centroids(qamd_pq *h, const float *data, qamd_mem data_mem, qamd_stop_fn stop, void *stop_user,
                            hipStream_t s) {
    const uint64_t dim = h->vp.dim, count = h->count;
    const uint32_t S = (uint32_t)std::min<uint64_t>(kKmeansSample, count);
    // The reference samples with a random Permutor and sorts the picks (:300-307); here an
    // evenly strided subset in index order (deterministic; given the same rows the centroids are
    // the reference's, see the kernels).
    DevBuf sample;
    QAMD_TRY(sample.alloc((size_t)S * dim * 4));
    if (data_mem == QAMD_MEM_HOST) {  // gather on the host, one upload (not S small copies)
        std::vector<float> picks((size_t)S * dim);
        for (uint32_t k = 0; k < S; k++) {
            const uint64_t r = (uint64_t)((unsigned __int128)k * count / S);
            memcpy(picks.data() + (size_t)k * dim, data + r * dim, dim * 4);
        }
        QAMD_TRY(copy_in(sample.ptr, picks.data(), QAMD_MEM_HOST, picks.size() * 4, s));
    } else {
        hipLaunchKernelGGL(km_gather_rows_kernel, dim3(grid_for((uint64_t)S * dim, kBlock * 4, 8)), dim3(kBlock), 0, s,
                           data, count, (uint32_t)dim, (uint64_t)S, sample.as<float>());
        QAMD_HIP(hipGetLastError());
    }
    if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :303-305
    return train_from_sample(h, sample.as<float>(), S, stop, stop_user, s);
}

std::string centroids_json(const qamd_pq *h) {
    std::string js = "[";
    const uint64_t dim = h->vp.dim;
    for (int k = 0; k < kCentroids; k++) {
        js += k ? ",[" : "[";
        for (uint64_t j = 0; j < dim; j++) {
            if (j) js += ",";
            js += json_f32(h->centroids_host[(size_t)k * dim + j]);
        }
        js += "]";
    }
    return js + "]";
}

}  // namespace

extern "C" {

uint64_t qamd_pq_quantized_vector_size(const qamd_vector_parameters *vp, uint64_t chunk_size) {
    return chunk_size ? chunks_of(vp->dim, chunk_size) : 0;
}

qamd_status qamd_pq_encode(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                           uint64_t chunk_size, const float *centroids, uint32_t max_kmeans_threads,
                           qamd_stop_fn stop, void *stop_user, void *stream, qamd_pq **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (chunk_size == 0) return fail(QAMD_ERR_ARGUMENTS, "chunk_size must be > 0");
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    if (vp->count > 0 && vp->dim > 0 && !data) return fail(QAMD_ERR_ARGUMENTS, "data is null");
    QAMD_ON_DEVICE(current_device());
    hipStream_t s = as_stream(stream);
    std::unique_ptr<qamd_pq> h(new qamd_pq);
    h->device = current_device();
    h->vp = *vp;
    h->chunk_size = chunk_size;
    h->count = vp->count;
    // the reference's worker count only matters here through the order of its f64 partial sums
    h->kmeans_workers = std::max<uint32_t>(1, max_kmeans_threads);
    QAMD_TRY(alloc_store(h.get()));
    const uint64_t dim = vp->dim, count = vp->count;
    if (centroids) {
        QAMD_TRY(set_centroids(h.get(), centroids, s));
    } else if (count <= (uint64_t)kCentroids) {
        // :290-297: the vectors themselves, zero-filled up to 256
        std::vector<float> cen((size_t)kCentroids * dim, 0.0f);
        if (count && dim) {
            if (data_mem == QAMD_MEM_HOST) memcpy(cen.data(), data, count * dim * 4);
            else QAMD_TRY(copy_out(cen.data(), QAMD_MEM_HOST, data, count * dim * 4, s));
        }
        QAMD_TRY(set_centroids(h.get(), cen.data(), s));
    } else {
        QAMD_TRY(train_centroids(h.get(), data, data_mem, stop, stop_user, s));
    }
    QAMD_TRY(encode_rows(h.get(), data, data_mem, stop, stop_user, s));
    if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :95-106
    *out = h.release();
    return QAMD_OK;
}

qamd_status qamd_pq_from_rows(const uint8_t *rows, qamd_mem rows_mem, const qamd_vector_parameters *vp,
                              uint64_t chunk_size, const float *centroids, void *stream, qamd_pq **out) {
    if (!vp || !out || !centroids) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (chunk_size == 0) return fail(QAMD_ERR_ARGUMENTS, "chunk_size must be > 0");
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    QAMD_ON_DEVICE(current_device());
    hipStream_t s = as_stream(stream);
    std::unique_ptr<qamd_pq> h(new qamd_pq);
    h->device = current_device();
    h->vp = *vp;
    h->chunk_size = chunk_size;
    h->count = vp->count;
    QAMD_TRY(alloc_store(h.get()));
    QAMD_TRY(set_centroids(h.get(), centroids, s));
    if (h->count && h->m) {
        if (!rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
        if (h->m == h->ds) {
            QAMD_TRY(copy_in(h->rows.ptr, rows, rows_mem, h->count * h->m, s));
        } else {
            DevBuf stage;
            const uint8_t *src = rows;
            if (rows_mem == QAMD_MEM_HOST) {
                QAMD_TRY(stage.alloc(h->count * h->m));
                QAMD_TRY(copy_in(stage.ptr, rows, QAMD_MEM_HOST, h->count * h->m, s));
                src = stage.as<uint8_t>();
            }
            hipLaunchKernelGGL(pq_restride_kernel, dim3(grid_for(h->count * h->m, kBlock * 4, 8)), dim3(kBlock), 0, s,
                               src, (uint32_t)h->m, h->rows.as<uint8_t>(), (uint32_t)h->ds, h->count, (uint32_t)h->m);
            QAMD_HIP(hipGetLastError());
            QAMD_HIP(hipStreamSynchronize(s));
        }
    }
    QAMD_TRY(build_planar(h.get(), 0, h->count, s));
    QAMD_HIP(hipStreamSynchronize(s));
    *out = h.release();
    return QAMD_OK;
}

// Rows [first_row, first_row + n_rows) as the reference's storage holds them (m code bytes per row;
// push_vector_data, encoded_storage.rs:17-25), in bounded pieces.
qamd_status qamd_pq_export_rows_range(const qamd_pq *h, uint64_t first_row, uint64_t n_rows, uint8_t *rows,
                                      qamd_mem rows_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (first_row > h->count || n_rows > h->count - first_row)
        return fail(QAMD_ERR_OUT_OF_RANGE, "rows [%llu, +%llu) out of range (count %llu)", (unsigned long long)first_row,
                    (unsigned long long)n_rows, (unsigned long long)h->count);
    if (n_rows == 0 || h->m == 0) return QAMD_OK;
    if (!rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint8_t *src = h->rows.as<uint8_t>() + first_row * h->ds;
    if (h->m == h->ds) return copy_out(rows, rows_mem, src, n_rows * h->m, s);
    DevBuf stage;
    uint8_t *dst = rows;
    if (rows_mem == QAMD_MEM_HOST) {
        QAMD_TRY(stage.alloc(n_rows * h->m));
        dst = stage.as<uint8_t>();
    }
    hipLaunchKernelGGL(pq_restride_kernel, dim3(grid_for(n_rows * h->m, kBlock * 4, 8)), dim3(kBlock), 0, s, src,
                       (uint32_t)h->ds, dst, (uint32_t)h->m, n_rows, (uint32_t)h->m);
    QAMD_HIP(hipGetLastError());
    if (rows_mem == QAMD_MEM_HOST) QAMD_TRY(copy_out(rows, QAMD_MEM_HOST, dst, n_rows * h->m, s));
    return QAMD_OK;
}

qamd_status qamd_pq_export_rows(const qamd_pq *h, uint8_t *rows, qamd_mem rows_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return qamd_pq_export_rows_range(h, 0, h->count, rows, rows_mem, stream);
}

qamd_status qamd_pq_get_centroids(const qamd_pq *h, float *centroids) {
    if (!h || !centroids) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    memcpy(centroids, h->centroids_host.data(), h->centroids_host.size() * sizeof(float));
    return QAMD_OK;
}

// save (:498-506): Metadata{centroids, vector_division:[{start,end}], vector_parameters}.
qamd_status qamd_pq_save(const qamd_pq *h, const char *data_path, const char *meta_path) {
    if (!h || !data_path || !meta_path) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::string js = "{\"centroids\":" + centroids_json(h) + ",\"vector_division\":[";
    for (uint64_t c = 0; c < h->m; c++) {
        const uint64_t lo = c * h->chunk_size, hi = std::min<uint64_t>(lo + h->chunk_size, h->vp.dim);
        if (c) js += ",";
        js += "{\"start\":" + std::to_string(lo) + ",\"end\":" + std::to_string(hi) + "}";
    }
    js += "],\"vector_parameters\":" + vector_parameters_json(h->vp) + "}";
    make_parent_dirs(meta_path);
    if (!write_file(meta_path, js.data(), js.size())) return fail(QAMD_ERR_IO, "cannot write %s", meta_path);
    std::vector<uint8_t> rows(h->count * h->m);
    QAMD_TRY(qamd_pq_export_rows(h, rows.data(), QAMD_MEM_HOST, nullptr));
    make_parent_dirs(data_path);
    if (!write_file(data_path, rows.data(), rows.size())) return fail(QAMD_ERR_IO, "cannot write %s", data_path);
    return QAMD_OK;
}

// load (:508-523): row size = metadata.vector_division.len(), count from the caller.
qamd_status qamd_pq_load(const char *data_path, const char *meta_path, const qamd_vector_parameters *vp,
                         qamd_pq **out) {
    if (!data_path || !meta_path || !vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    JsonValue root;
    QAMD_TRY(read_metadata(meta_path, root));
    qamd_vector_parameters file_vp{};
    std::vector<float> cen;
    uint64_t m = 0, chunk_size = 0;
    {   // Metadata{centroids: Vec<Vec<f32>>, vector_division: Vec<Range<usize>>, vector_parameters} (:39-44), any key order
        std::string err;
        const JsonValue *vpj = json_field(root, "vector_parameters", err);
        if (!vpj || !parse_vector_parameters(*vpj, file_vp, err)) return fail(QAMD_ERR_IO, "%s: %s", meta_path, err.c_str());
        const JsonValue *cj = json_field(root, "centroids", err);
        if (!cj) return fail(QAMD_ERR_IO, "%s: %s", meta_path, err.c_str());
        if (cj->kind != JsonValue::Array) return fail(QAMD_ERR_IO, "%s: centroids: expected a sequence", meta_path);
        if (cj->items.size() != (size_t)kCentroids)
            return fail(QAMD_ERR_IO, "%s: metadata holds %zu centroids, expected %d", meta_path, cj->items.size(), kCentroids);
        cen.reserve((size_t)kCentroids * file_vp.dim);
        for (const JsonValue &row : cj->items) {
            if (row.kind != JsonValue::Array || row.items.size() != file_vp.dim)
                return fail(QAMD_ERR_IO, "%s: every centroid must be a sequence of dim = %llu numbers", meta_path,
                            (unsigned long long)file_vp.dim);
            for (const JsonValue &x : row.items) {
                float f = 0.0f;
                if (!json_number_as_f32(x, f, err)) return fail(QAMD_ERR_IO, "%s: centroids: %s", meta_path, err.c_str());
                cen.push_back(f);
            }
        }
        // vector_division: the ranges get_vector_division (:116-121) produces - the first one's length is the chunk size,
        // their number the row size
        const JsonValue *dj = json_field(root, "vector_division", err);
        if (!dj) return fail(QAMD_ERR_IO, "%s: %s", meta_path, err.c_str());
        if (dj->kind != JsonValue::Array) return fail(QAMD_ERR_IO, "%s: vector_division: expected a sequence", meta_path);
        m = dj->items.size();
        for (uint64_t c = 0; c < m; c++) {
            uint64_t st = 0, en = 0;
            if (!json_usize(dj->items[c], "start", st, err) || !json_usize(dj->items[c], "end", en, err))
                return fail(QAMD_ERR_IO, "%s: vector_division[%llu]: %s", meta_path, (unsigned long long)c, err.c_str());
            if (c == 0) chunk_size = en > st ? en - st : 0;
            if (chunk_size == 0 || st != c * chunk_size || en != std::min<uint64_t>(st + chunk_size, file_vp.dim))
                return fail(QAMD_ERR_IO, "%s: vector_division does not tile dim %llu in chunks of %llu", meta_path,
                            (unsigned long long)file_vp.dim, (unsigned long long)chunk_size);
        }
        if (chunk_size == 0) chunk_size = 1;  // dim == 0: no ranges
    }
    std::string bytes;
    if (!read_file(data_path, bytes)) return fail(QAMD_ERR_IO, "cannot read %s", data_path);
    const uint64_t expected = m * vp->count;
    if (bytes.size() != expected)
        return fail(QAMD_ERR_IO, "Loaded storage size %zu is not equal to expected size %llu", bytes.size(),
                    (unsigned long long)expected);
    qamd_vector_parameters eff = file_vp;
    eff.count = vp->count;
    if (chunks_of(eff.dim, chunk_size) != m) return fail(QAMD_ERR_IO, "vector_division does not tile dim");
    return qamd_pq_from_rows(reinterpret_cast<const uint8_t *>(bytes.data()), QAMD_MEM_HOST, &eff, chunk_size,
                             cen.data(), nullptr, out);
}

qamd_status qamd_pq_encode_query(const qamd_pq *h, const float *query, uint64_t qdim, qamd_mem query_mem,
                                 void *stream, qamd_pq_query **query_io) {
    if (!h || !query_io || (!query && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (qdim != h->vp.dim)  // the reference slices query[range] and would panic (:529)
        return fail(QAMD_ERR_ARGUMENTS, "query has %llu dims, store has %llu", (unsigned long long)qdim,
                    (unsigned long long)h->vp.dim);
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    qamd_pq_query *q = *query_io;
    std::unique_ptr<qamd_pq_query> fresh;
    if (!q) {
        fresh.reset(new qamd_pq_query);
        q = fresh.get();
        q->device = h->device;
    }
    const size_t n = (size_t)h->m * kCentroids;
    const bool with_t = skew_capable(h) || skew_sliced_capable(h);
    if (q->m != h->m || !q->lut.ptr || q->transposed != with_t) {
        QAMD_TRY(q->lut.alloc(std::max<size_t>(n, 4) * sizeof(float) * (with_t ? 2 : 1)));
        q->m = h->m;
        q->transposed = with_t;
    }
    if (n) {
        DevBuf qtmp;
        const float *qd = query;
        if (query_mem == QAMD_MEM_HOST) {
            QAMD_TRY(qtmp.alloc(qdim * 4));
            QAMD_TRY(copy_in(qtmp.ptr, query, QAMD_MEM_HOST, qdim * 4, s));
            qd = qtmp.as<float>();
        }
        hipLaunchKernelGGL(pq_lut_kernel, dim3((uint32_t)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, qd,
                           (uint32_t)h->vp.dim, (uint32_t)h->chunk_size, (uint32_t)h->m, h->centroids.as<float>(),
                           h->vp.distance_type, h->vp.invert, q->lut.as<float>(), with_t ? q->lut.as<float>() + n : nullptr);
        QAMD_HIP(hipGetLastError());
        if (query_mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));  // qtmp is freed on return
    }
    QAMD_TRY(q->ready.record(s));
    if (fresh) *query_io = fresh.release();
    return QAMD_OK;
}

qamd_status qamd_pq_query_read(const qamd_pq_query *q, float *lut, uint64_t capacity, uint64_t *len) {
    if (!q) return fail(QAMD_ERR_ARGUMENTS, "null query");
    const uint64_t n = q->m * kCentroids;
    if (len) *len = n;
    if (lut) {
        if (capacity < n) return fail(QAMD_ERR_ARGUMENTS, "lut buffer too small");
        QAMD_ON_DEVICE(q->device);
        QAMD_TRY(q->ready.wait(nullptr));
        QAMD_TRY(copy_out(lut, QAMD_MEM_HOST, q->lut.ptr, n * 4, nullptr));
    }
    return QAMD_OK;
}

void qamd_pq_query_free(qamd_pq_query *q) { delete q; }

qamd_status qamd_pq_score_all(const qamd_pq *h, const qamd_pq_query *q, float *out, qamd_mem out_mem,
                              void *stream) {
    QAMD_TRY(check_query(h, q));
    if (h->count == 0) return QAMD_OK;
    if (!out) return fail(QAMD_ERR_ARGUMENTS, "out is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(q->ready.wait(s));
    if (out_mem == QAMD_MEM_DEVICE) return scan_launch(h, q->lut.as<float>(), nullptr, h->count, out, s, nullptr, q->lut_t());
    float *tmp = nullptr;  // per-thread workspace: no hipMalloc / hipFree per query
    QAMD_TRY(thread_ws_acquire(WS_SCORES, h->count * 4, s, reinterpret_cast<void **>(&tmp)));
    qamd_status st = scan_launch(h, q->lut.as<float>(), nullptr, h->count, tmp, s, nullptr, q->lut_t());
    if (st == QAMD_OK) st = copy_out(out, QAMD_MEM_HOST, tmp, h->count * 4, s);
    thread_ws_release(WS_SCORES, s, st == QAMD_OK);  // the download synchronised the stream
    return st;
}

qamd_status qamd_pq_score_ids(const qamd_pq *h, const qamd_pq_query *q, const uint32_t *ids, uint64_t n_ids,
                              qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_query(h, q));
    if (n_ids == 0) return QAMD_OK;
    if (!ids || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(q->ready.wait(s));
    DevBuf ids_tmp, out_tmp;
    const uint32_t *ids_dev = ids;
    // per-pair granularity (score_point and friends): ids and results through the calling
    // thread's mapped host scratch -- no allocation, no copy calls
    const HostScratch hs = (ids_mem == QAMD_MEM_HOST && out_mem == QAMD_MEM_HOST && n_ids <= 1024) ? host_scratch()
                                                                                                  : HostScratch{};
    if (hs.host) {
        for (uint64_t k = 0; k < n_ids; k++) {
            if (ids[k] >= h->count)
                return fail(QAMD_ERR_OUT_OF_RANGE, "row id %u out of range (count %llu)", ids[k],
                            (unsigned long long)h->count);
            hs.host[k] = ids[k];
        }
        QAMD_TRY(scan_launch(h, q->lut.as<float>(), hs.dev, n_ids, reinterpret_cast<float *>(hs.dev + 1024), s));
        QAMD_HIP(hipStreamSynchronize(s));
        memcpy(out, hs.host + 1024, n_ids * 4);
        return QAMD_OK;
    }
    if (ids_mem == QAMD_MEM_HOST) {
        for (uint64_t k = 0; k < n_ids; k++)
            if (ids[k] >= h->count)
                return fail(QAMD_ERR_OUT_OF_RANGE, "row id %u out of range (count %llu)", ids[k],
                            (unsigned long long)h->count);
        QAMD_TRY(ids_tmp.alloc(n_ids * 4));
        QAMD_TRY(copy_in(ids_tmp.ptr, ids, QAMD_MEM_HOST, n_ids * 4, s));
        ids_dev = ids_tmp.as<uint32_t>();
    }
    float *out_dev = out;
    if (out_mem == QAMD_MEM_HOST) {
        QAMD_TRY(out_tmp.alloc(n_ids * 4));
        out_dev = out_tmp.as<float>();
    }
    QAMD_TRY(scan_launch(h, q->lut.as<float>(), ids_dev, n_ids, out_dev, s));
    if (out_mem == QAMD_MEM_HOST) QAMD_TRY(copy_out(out, QAMD_MEM_HOST, out_dev, n_ids * 4, s));
    else if (ids_mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));
    return QAMD_OK;
}

qamd_status qamd_pq_score_point(const qamd_pq *h, const qamd_pq_query *q, uint32_t i, float *out) {
    return qamd_pq_score_ids(h, q, &i, 1, QAMD_MEM_HOST, out, QAMD_MEM_HOST, nullptr);
}

namespace {
qamd_status internal_pairs_launch(const qamd_pq *h, uint32_t single_row, const uint32_t *lists, uint32_t n_lists,
                                  const uint32_t *list_rows, const uint32_t *ids_dev, uint64_t n, float *out_dev,
                                  hipStream_t s) {
    if (n == 0) return QAMD_OK;
    const uint32_t ppb = pairs_per_block(n, 16);  // 16 lane groups per workgroup
    hipLaunchKernelGGL(pq_internal_pairs_kernel, dim3((unsigned)((n + ppb - 1) / ppb)), dim3(kBlock), 0, s,
                       h->rows.as<uint8_t>(), (uint32_t)h->ds, (uint32_t)h->vp.dim, (uint32_t)h->chunk_size, (uint32_t)h->m,
                       h->centroids.as<float>(), h->vp.distance_type, h->vp.invert, single_row, lists, n_lists, list_rows,
                       ids_dev, n, (uint32_t)h->count, ppb, out_dev);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}
}  // namespace

// score_internal (:566-593) for one stored row against many: out[k] = score_internal(i, ids[k]).
qamd_status qamd_pq_score_internal_ids(const qamd_pq *h, uint32_t i, const uint32_t *ids, uint64_t n_ids,
                                       qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (n_ids == 0) return QAMD_OK;
    if (!ids || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (i >= h->count) return fail(QAMD_ERR_OUT_OF_RANGE, "row id %u out of range (count %llu)", i, (unsigned long long)h->count);
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    // one list of n_ids ids: the list machinery with the row passed by value (no offsets needed)
    if (ids_mem == QAMD_MEM_DEVICE) {
        if (out_mem == QAMD_MEM_HOST) {
            StreamBuf res;
            QAMD_TRY(res.alloc(n_ids * 4, s));
            QAMD_TRY(internal_pairs_launch(h, i, nullptr, 0, nullptr, ids, n_ids, res.as<float>(), s));
            return copy_out(out, QAMD_MEM_HOST, res.ptr, n_ids * 4, s);
        }
        return internal_pairs_launch(h, i, nullptr, 0, nullptr, ids, n_ids, out, s);
    }
    const uint32_t offs[2] = {0u, (uint32_t)n_ids};
    if (n_ids > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "too many ids");
    return run_lists(offs, 1, ids, n_ids, nullptr, QAMD_MEM_HOST, out, out_mem, h->count, s, [&](const ListArgs &a) {
        return internal_pairs_launch(h, i, nullptr, 0, nullptr, a.ids, a.n_pairs, a.out, s);
    });
}

qamd_status qamd_pq_score_internal(const qamd_pq *h, uint32_t i, uint32_t j, float *out) {
    if (!h || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (j >= h->count) return fail(QAMD_ERR_OUT_OF_RANGE, "row id out of range (count %llu)", (unsigned long long)h->count);
    return qamd_pq_score_internal_ids(h, i, &j, 1, QAMD_MEM_HOST, out, QAMD_MEM_HOST, nullptr);
}

// Many stored rows, each against its own id list, in one launch (lists.hpp).
qamd_status qamd_pq_score_internal_ids_batch(const qamd_pq *h, const uint32_t *rows, const uint32_t *list_offsets,
                                             uint32_t n_lists, const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem,
                                             float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (n_lists && !rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    return run_lists(list_offsets, n_lists, ids, n_ids, rows, lists_mem, out, out_mem, h->count, s, [&](const ListArgs &a) {
        return internal_pairs_launch(h, 0, a.offsets, a.n_lists, a.rows, a.ids, a.n_pairs, a.out, s);
    });
}

qamd_status qamd_pq_topk(const qamd_pq *h, const qamd_pq_query *q, uint32_t k, int largest, uint32_t *out_ids,
                         float *out_scores, qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_query(h, q));
    if (k == 0) return QAMD_OK;
    if (!out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null output");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(q->ready.wait(s));
    const float *lut = q->lut.as<float>();
    {   // small stores: one launch (LUT + the merge lists fit the LDS up to m = 128)
        qamd_status st = QAMD_OK;
        if (pq_topk_small(h, lut, k, largest, out_ids, out_scores, out_mem, s, st)) return st;
    }
    if (!fast_capable(h, h->count)) {
        float *scores = nullptr;
        QAMD_TRY(thread_ws_acquire(WS_SCORES, std::max<uint64_t>(h->count, 1) * 4, s, reinterpret_cast<void **>(&scores)));
        qamd_status st = scan_launch(h, lut, nullptr, h->count, scores, s);
        if (st == QAMD_OK) st = topk_finish(scores, h->count, k, largest, out_ids, out_scores, out_mem, s);
        thread_ws_release(WS_SCORES, s);
        return st;
    }
    FusedScan scan;
    const float *lut_t = q->lut_t();
    scan.scan_scores = [&](float *scores, hipStream_t st) { return scan_launch(h, lut, nullptr, h->count, scores, st, nullptr, lut_t); };
    scan.scan_filter = [&](const TopkFilter &f, hipStream_t st) {
        return scan_launch(h, lut, nullptr, h->count, nullptr, st, &f, lut_t);
    };
    scan.score_ids = [&](const uint32_t *ids, uint64_t n_ids, float *out, hipStream_t st) {
        return scan_launch(h, lut, ids, n_ids, out, st);
    };
    return fused_topk(h->count, k, largest, out_ids, out_scores, out_mem, s, scan);
}

void qamd_pq_free(qamd_pq *h) { delete h; }

const char *qamd_pq_scan_kernel(const qamd_pq *h, uint32_t *n_launches) {
    uint32_t launches = 1;
    const char *name = "pq_scan_kernel";
    if (h && fast_capable(h, h->count)) {
        const uint32_t pieces = (uint32_t)valid_pieces(h->m);
        if (skew_capable(h)) {
            name = "pq_scan_skew_kernel";
        } else if (skew_sliced_capable(h)) {
            name = "pq_scan_skew_kernel<SLICED>";
            launches = (uint32_t)h->slice_chunks.size();
        } else {
            name = "pq_scan_fast_kernel";
            if (pieces > kMaxSlicePieces) launches = (pieces + kSlicePiecesAligned - 1) / kSlicePiecesAligned;
        }
    }
    if (n_launches) *n_launches = launches;
    return name;
}

qamd_status qamd_pq_kmeans_info(const qamd_pq *h, uint32_t *iterations, uint32_t *empty_clusters) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (iterations) *iterations = h->kmeans_iterations;
    if (empty_clusters) *empty_clusters = h->kmeans_empty_clusters;
    return QAMD_OK;
}

}  // extern "C"

// ============================================================================= streaming encode
// EncodedVectorsPQ::encode (:56-107) walks its clonable iterator twice: find_centroids (:278-342,
// the <= 10 000 sampled rows) and encode_storage (:136-226, one code row per vector, pushed in row
// order).  observe() is the first walk, push() the second, in bounded batches; same kernels and the
// same sample rows as qamd_pq_encode, so the result is byte-identical to the one-shot call.
struct qamd_pq_encoder {
    int device = 0;
    hipStream_t stream = nullptr;
    qamd_stop_fn stop = nullptr;
    void *stop_user = nullptr;
    std::unique_ptr<qamd_pq> h;
    bool have_centroids = false;
    uint64_t observed = 0, pushed = 0;
    uint32_t S = 0;  // sample rows kept by observe()
    DevBuf sample, stage, pair_table;
};

namespace {

qamd_status pq_encoder_close_pass1(qamd_pq_encoder *e) {
    if (e->have_centroids) return QAMD_OK;
    qamd_pq *h = e->h.get();
    const uint64_t dim = h->vp.dim, count = h->count;
    if (e->observed != count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu (observe pass ended early)",
                    (unsigned long long)e->observed, (unsigned long long)count);
    if (count <= (uint64_t)kCentroids) {  // :290-297: the vectors themselves, zero-filled up to 256
        std::vector<float> cen((size_t)kCentroids * dim, 0.0f);
        if (count && dim) QAMD_TRY(copy_out(cen.data(), QAMD_MEM_HOST, e->sample.ptr, count * dim * 4, e->stream));
        QAMD_TRY(set_centroids(h, cen.data(), e->stream));
    } else {
        if (e->stop && e->stop(e->stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :303-305
        QAMD_TRY(train_from_sample(h, e->sample.as<float>(), e->S, e->stop, e->stop_user, e->stream));
    }
    e->sample.release();
    e->have_centroids = true;
    return QAMD_OK;
}

}  // namespace

extern "C" {

qamd_status qamd_pq_encoder_begin(const qamd_vector_parameters *vp, uint64_t chunk_size, const float *centroids,
                                  uint32_t max_kmeans_threads, qamd_stop_fn stop, void *stop_user, void *stream,
                                  qamd_pq_encoder **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (chunk_size == 0) return fail(QAMD_ERR_ARGUMENTS, "chunk_size must be > 0");
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    QAMD_ON_DEVICE(current_device());
    std::unique_ptr<qamd_pq_encoder> e(new qamd_pq_encoder);
    e->device = current_device();
    e->stream = as_stream(stream);
    e->stop = stop;
    e->stop_user = stop_user;
    e->h.reset(new qamd_pq);
    qamd_pq *h = e->h.get();
    h->device = e->device;
    h->vp = *vp;
    h->chunk_size = chunk_size;
    h->count = vp->count;
    h->kmeans_workers = std::max<uint32_t>(1, max_kmeans_threads);
    QAMD_TRY(alloc_store(h));
    if (centroids) {
        QAMD_TRY(set_centroids(h, centroids, e->stream));
        e->have_centroids = true;
    } else {
        e->S = (uint32_t)std::min<uint64_t>(kKmeansSample, vp->count);
        QAMD_TRY(e->sample.alloc(std::max<uint64_t>((uint64_t)e->S * vp->dim, 4) * 4));
    }
    *out = e.release();
    return QAMD_OK;
}

qamd_status qamd_pq_encoder_observe(qamd_pq_encoder *e, const float *batch, uint64_t n_rows, qamd_mem batch_mem) {
    if (!e || (!batch && n_rows && e->h->vp.dim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (e->stop && e->stop(e->stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");
    if (e->have_centroids) {
        e->observed += n_rows;
        return QAMD_OK;
    }
    const uint64_t dim = e->h->vp.dim, count = e->h->count;
    if (e->pushed) return fail(QAMD_ERR_ARGUMENTS, "observe after push");
    if (e->observed + n_rows > count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)(e->observed + n_rows), (unsigned long long)count);
    QAMD_ON_DEVICE(e->device);
    const uint64_t piece_rows = std::max<uint64_t>(1, (256ull << 20) / std::max<uint64_t>(dim * 4, 1));
    for (uint64_t r = 0; r < n_rows && dim && e->S; r += piece_rows) {
        const uint64_t nr = std::min(piece_rows, n_rows - r), base = e->observed + r;
        // sample slots whose row floor(k * count / S) falls into [base, base + nr)
        const uint64_t k0 = (uint64_t)(((unsigned __int128)base * e->S + count - 1) / count);
        const uint64_t k1 = std::min<uint64_t>(e->S, (uint64_t)(((unsigned __int128)(base + nr) * e->S + count - 1) / count));
        if (k1 <= k0) continue;
        const void *src = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(batch + r * dim, batch_mem, nr * dim * 4, e->stage, e->stream, &src, &staged));
        hipLaunchKernelGGL(km_gather_batch_kernel, dim3(grid_for((k1 - k0) * dim, kBlock * 4, 8)), dim3(kBlock), 0,
                           e->stream, static_cast<const float *>(src), base, (uint32_t)dim, count, (uint64_t)e->S, k0, k1,
                           e->sample.as<float>());
        QAMD_HIP(hipGetLastError());
        if (staged) QAMD_HIP(hipStreamSynchronize(e->stream));  // the staging buffer is reused
    }
    e->observed += n_rows;
    return QAMD_OK;
}

qamd_status qamd_pq_encoder_push(qamd_pq_encoder *e, const float *batch, uint64_t n_rows, qamd_mem batch_mem) {
    if (!e || (!batch && n_rows && e->h->vp.dim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (e->stop && e->stop(e->stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :198-200
    qamd_pq *h = e->h.get();
    if (e->pushed + n_rows > h->count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)(e->pushed + n_rows), (unsigned long long)h->count);
    QAMD_ON_DEVICE(e->device);
    QAMD_TRY(pq_encoder_close_pass1(e));
    const uint64_t dim = h->vp.dim;
    if (dim && !e->pair_table.ptr && cs_fast_shape(dim, h->chunk_size))
        QAMD_TRY(build_pair_table(h->centroids.as<float>(), dim, h->chunk_size, h->m, e->pair_table, e->stream));
    const uint64_t piece_rows = std::max<uint64_t>(1, (256ull << 20) / std::max<uint64_t>(dim * 4, 1));
    for (uint64_t r = 0; r < n_rows && dim; r += piece_rows) {
        const uint64_t nr = std::min(piece_rows, n_rows - r);
        const void *src = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(batch + r * dim, batch_mem, nr * dim * 4, e->stage, e->stream, &src, &staged));
        QAMD_TRY(launch_assign(static_cast<const float *>(src), nr, dim, h->chunk_size, h->m, h->centroids.as<float>(),
                               &e->pair_table, h->rows.as<uint8_t>(), h->ds, e->pushed + r, e->stream));
        QAMD_TRY(build_planar(h, e->pushed + r, nr, e->stream));
        if (staged) QAMD_HIP(hipStreamSynchronize(e->stream));
    }
    e->pushed += n_rows;
    return QAMD_OK;
}

qamd_status qamd_pq_encoder_finish(qamd_pq_encoder *e, qamd_pq **out) {
    if (!e || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_pq_encoder> own(e);
    if (e->pushed != e->h->count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)e->pushed, (unsigned long long)e->h->count);
    QAMD_ON_DEVICE(e->device);
    QAMD_TRY(pq_encoder_close_pass1(e));  // count == 0, or a caller that never pushed an empty store
    QAMD_HIP(hipStreamSynchronize(e->stream));
    if (e->stop && e->stop(e->stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :95-106
    *out = e->h.release();
    return QAMD_OK;
}

void qamd_pq_encoder_abort(qamd_pq_encoder *e) {
    if (!e) return;
    DeviceGuard g(e->device);
    (void)hipStreamSynchronize(e->stream);
    delete e;
}

}  // extern "C"

// ============================================================================= many queries at once
// The caller's OUTER loop over queries (demos/src/ann_benchmark.rs:245-260), each with its 30-entry
// heap (ann_benchmark_data.rs:151-167).  A PQ scan serves one query at a time -- the LDS holds one
// chunk-major LUT (96 KiB at m = 96) and the gather rate of that LDS, not HBM, bounds it -- so the
// batch form builds all LUTs with one launch and then ENQUEUES the per-query scans back to back:
// no per-query allocation, no per-query synchronisation (fused_topk_batch reads the statuses back
// once per 32 queries).  Every score and every list is bit-identical to the single-query calls.
struct qamd_pq_query_batch {
    int device = 0;
    uint64_t m = 0, n_queries = 0;
    DevBuf luts;  // [n_queries][m * 256] f32, chunk-major; `transposed`: the [code][chunk] copies follow, same order
    bool transposed = false;
    const float *lut_t(uint64_t q) const { return transposed ? luts.as<float>() + (n_queries + q) * m * kCentroids : nullptr; }
};

extern "C" {

qamd_status qamd_pq_encode_query_batch(const qamd_pq *h, const float *queries, uint64_t n_queries, uint64_t qdim,
                                       qamd_mem queries_mem, void *stream, qamd_pq_query_batch **batch_io) {
    if (!h || !batch_io || (!queries && n_queries && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (n_queries && qdim != h->vp.dim)
        return fail(QAMD_ERR_ARGUMENTS, "queries have %llu dims, store has %llu", (unsigned long long)qdim,
                    (unsigned long long)h->vp.dim);
    if (n_queries > 65535) return fail(QAMD_ERR_ARGUMENTS, "at most 65535 queries per batch");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    qamd_pq_query_batch *b = *batch_io;
    std::unique_ptr<qamd_pq_query_batch> fresh;
    if (!b) {
        fresh.reset(new qamd_pq_query_batch);
        b = fresh.get();
        b->device = h->device;
    }
    const bool with_t = skew_capable(h) || skew_sliced_capable(h);
    const size_t per = (size_t)h->m * kCentroids, need = std::max<size_t>(per * n_queries, 4) * sizeof(float) * (with_t ? 2 : 1);
    if (b->luts.bytes < need) QAMD_TRY(b->luts.alloc(need));
    b->m = h->m;
    b->n_queries = n_queries;
    b->transposed = with_t;
    if (per && n_queries) {
        DevBuf qtmp;
        const void *qd = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(queries, queries_mem, n_queries * qdim * 4, qtmp, s, &qd, &staged));
        hipLaunchKernelGGL(pq_lut_batch_kernel, dim3((uint32_t)((per + kBlock - 1) / kBlock), (uint32_t)n_queries),
                           dim3(kBlock), 0, s, static_cast<const float *>(qd), (uint32_t)h->vp.dim, (uint32_t)h->chunk_size,
                           (uint32_t)h->m, h->centroids.as<float>(), h->vp.distance_type, h->vp.invert, b->luts.as<float>(),
                           with_t ? b->luts.as<float>() + per * n_queries : nullptr);
        QAMD_HIP(hipGetLastError());
        if (staged) QAMD_HIP(hipStreamSynchronize(s));  // qtmp is freed on return
    }
    if (fresh) *batch_io = fresh.release();
    return QAMD_OK;
}

void qamd_pq_query_batch_free(qamd_pq_query_batch *b) { delete b; }

static qamd_status pq_check_batch(const qamd_pq *h, const qamd_pq_query_batch *b) {
    if (!h || !b) return fail(QAMD_ERR_ARGUMENTS, "null handle or query batch");
    if (b->m != h->m) return fail(QAMD_ERR_ARGUMENTS, "query LUTs have %llu chunks, store has %llu",
                                  (unsigned long long)b->m, (unsigned long long)h->m);
    return QAMD_OK;
}

// Many (query, id list) pairs in one launch (lists.hpp): out[p] = score_point(query l, ids[p]).
qamd_status qamd_pq_score_ids_batch(const qamd_pq *h, const qamd_pq_query_batch *b, const uint32_t *list_offsets,
                                    uint32_t n_lists, const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem, float *out,
                                    qamd_mem out_mem, void *stream) {
    QAMD_TRY(pq_check_batch(h, b));
    if (n_lists > b->n_queries)
        return fail(QAMD_ERR_ARGUMENTS, "%u lists, but the batch holds %llu queries", n_lists, (unsigned long long)b->n_queries);
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const size_t per = (size_t)h->m * kCentroids;
    return run_lists(list_offsets, n_lists, ids, n_ids, nullptr, lists_mem, out, out_mem, h->count, s, [&](const ListArgs &a) {
        const uint32_t row_words = (uint32_t)(h->ds / 4);
        const size_t lut_bytes = per * sizeof(float);
        // long lists (what a caller knows without reading device lists: the mean length): one LUT staging per list segment
        if (a.n_pairs >= 4096 && a.n_pairs / a.n_lists >= 2 * kListStageMin && lut_bytes <= kLdsBudget && h->ds % 16 == 0 &&
            h->ds / 16 <= 10) {
            const unsigned grid = (unsigned)((a.n_pairs + kListStagePairs - 1) / kListStagePairs);
#define QAMD_PQ_STAGED(NPV)                                                                                  \
    case NPV:                                                                                               \
        QAMD_LDS_OPT_IN((&pq_lists_staged_kernel<NPV>), kLdsBudget);                                        \
        hipLaunchKernelGGL(pq_lists_staged_kernel<NPV>, dim3(grid), dim3(kScanBlock), lut_bytes, s, h->rows.as<uint4>(), \
                           b->luts.as<float>(), (uint64_t)per, a.offsets, a.n_lists, a.ids, a.n_pairs, (uint32_t)h->count, \
                           (uint32_t)h->m, a.out);                                                          \
        break;
            switch (h->ds / 16) {
                QAMD_PQ_STAGED(1) QAMD_PQ_STAGED(2) QAMD_PQ_STAGED(3) QAMD_PQ_STAGED(4) QAMD_PQ_STAGED(5)
                QAMD_PQ_STAGED(6) QAMD_PQ_STAGED(7) QAMD_PQ_STAGED(8) QAMD_PQ_STAGED(9) QAMD_PQ_STAGED(10)
            }
#undef QAMD_PQ_STAGED
            QAMD_HIP(hipGetLastError());
            return QAMD_OK;
        }
        const uint32_t ppb = pairs_per_block(a.n_pairs, 64, 1);  // 64 lane groups per workgroup, one pass: the window of LUTs in use stays L2-sized
        const unsigned grid = (unsigned)((a.n_pairs + ppb - 1) / ppb);
        if (row_words % 4 == 0)
            hipLaunchKernelGGL(pq_lists_kernel<true>, dim3(grid), dim3(kBlock), 0, s, h->rows.as<uint32_t>(), b->luts.as<float>(),
                               (uint64_t)per, a.offsets, a.n_lists, a.ids, a.n_pairs, (uint32_t)h->count, (uint32_t)h->m,
                               row_words, ppb, a.out);
        else
            hipLaunchKernelGGL(pq_lists_kernel<false>, dim3(grid), dim3(kBlock), 0, s, h->rows.as<uint32_t>(), b->luts.as<float>(),
                               (uint64_t)per, a.offsets, a.n_lists, a.ids, a.n_pairs, (uint32_t)h->count, (uint32_t)h->m,
                               row_words, ppb, a.out);
        QAMD_HIP(hipGetLastError());
        return QAMD_OK;
    });
}

qamd_status qamd_pq_score_batch(const qamd_pq *h, const qamd_pq_query_batch *b, float *out, qamd_mem out_mem,
                                void *stream) {
    QAMD_TRY(pq_check_batch(h, b));
    if (h->count == 0 || b->n_queries == 0) return QAMD_OK;
    if (!out) return fail(QAMD_ERR_ARGUMENTS, "out is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const size_t per = (size_t)h->m * kCentroids;
    StreamBuf tmp;
    float *out_dev = out;
    if (out_mem == QAMD_MEM_HOST) {
        QAMD_TRY(tmp.alloc(b->n_queries * h->count * 4, s));
        out_dev = tmp.as<float>();
    }
    for (uint64_t q = 0; q < b->n_queries; q++)
        QAMD_TRY(scan_launch(h, b->luts.as<float>() + q * per, nullptr, h->count, out_dev + q * h->count, s, nullptr, b->lut_t(q)));
    if (out_mem == QAMD_MEM_HOST) return copy_out(out, QAMD_MEM_HOST, out_dev, b->n_queries * h->count * 4, s);
    return QAMD_OK;
}

qamd_status qamd_pq_topk_batch(const qamd_pq *h, const qamd_pq_query_batch *b, uint32_t k, int largest,
                               uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    QAMD_TRY(pq_check_batch(h, b));
    if (k == 0 || b->n_queries == 0) return QAMD_OK;
    if (!out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null output");
    QAMD_ON_DEVICE(h->device);
    const size_t per = (size_t)h->m * kCentroids;
    const float *luts = b->luts.as<float>();
    BatchScan scan;
    scan.filter_capable = fast_capable(h, h->count);
    scan.scan_scores = [&](uint32_t q, float *scores, hipStream_t st) {
        return scan_launch(h, luts + q * per, nullptr, h->count, scores, st, nullptr, b->lut_t(q));
    };
    scan.scan_filter = [&](uint32_t q, const TopkFilter &f, hipStream_t st) {
        return scan_launch(h, luts + q * per, nullptr, h->count, nullptr, st, &f, b->lut_t(q));
    };
    scan.score_ids = [&](uint32_t q, const uint32_t *ids, uint64_t n_ids, float *out, hipStream_t st) {
        return scan_launch(h, luts + q * per, ids, n_ids, out, st);
    };
    scan.topk_small = [&](uint32_t q, uint32_t *ids, float *sc, hipStream_t st, qamd_status &status) {
        return pq_topk_small(h, luts + q * per, k, largest, ids, sc, QAMD_MEM_DEVICE, st, status);
    };
    // Four (two) queries' filter passes side by side: every query's table in the LDS of its own CUs, the rows shared through L2
    // (SkewBatch).  Measured, ms per query of the whole batch step (bench.py --quantizer pq --batch-queries, profiles/r04_pq.txt):
    //   10M x 768 (m = 96): query by query 0.223-0.229, two 0.2075, four 0.201-0.205, EIGHT 0.45 (all eight workgroups of a stream
    //   ask the same L2 lines at the same time); 12.5M x 1536 (m = 192): 0.585, two 0.563, four 0.545.  Plain instead of nt loads: worse
    //   everywhere (0.298 query by query).  Stores of a million rows and more (a quarter of the machine per query keeps its waves fed).
    static const char *eside = dev_env("QAMD_PQ_SIDE");  // developer A/B: 0 = query by query; 2 / 4 / 8: at most so many side by side
    if (b->transposed && fast_capable(h, h->count) && device_info().cu_count == 256 && h->count >= (1u << 20) && !(eside && eside[0] == '0'))
        scan.scan_filter_multi = [&](uint32_t q, uint32_t avail, const TopkFilterSlices &fsl, hipStream_t st, qamd_status &status) -> uint32_t {
            const uint32_t most = eside ? (uint32_t)atoi(eside) : 4u;
            const uint32_t g = avail >= 8 && most >= 8 ? 8u : avail >= 4 && most >= 4 ? 4u : avail >= 2 && most >= 2 ? 2u : 0u;
            if (g == 0) return 0;
            SkewBatch many{};
            many.n = g;
            many.lut_stride = (uint32_t)per;
            many.fsl = fsl;
            bool done = false;
            status = launch_fast<true>(h, nullptr, nullptr, nullptr, st, b->lut_t(q), &many, &done);
            return status == QAMD_OK && done ? g : 0u;
        };
    return fused_topk_batch(h->count, (uint32_t)b->n_queries, k, largest, out_ids, out_scores, out_mem, as_stream(stream),
                            scan);
}

}  // extern "C"

namespace qamd {

// find_centroids (:278-342) on its own: what the sharded encoder (sharded.hip) runs once before
// every shard encodes with the result.  Runs on the current device.
qamd_status pq_train_centroids(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                               uint64_t chunk_size, uint32_t max_kmeans_threads, qamd_stop_fn stop, void *stop_user,
                               hipStream_t s, std::vector<float> &centroids, uint32_t *iterations, uint32_t *empties) {
    qamd_pq tmp;
    tmp.vp = *vp;
    tmp.chunk_size = chunk_size;
    tmp.count = vp->count;
    tmp.m = chunks_of(vp->dim, chunk_size);
    tmp.kmeans_workers = std::max<uint32_t>(1, max_kmeans_threads);
    const uint64_t dim = vp->dim, count = vp->count;
    if (count <= (uint64_t)kCentroids) {  // :290-297
        centroids.assign((size_t)kCentroids * dim, 0.0f);
        if (count && dim) {
            if (data_mem == QAMD_MEM_HOST) memcpy(centroids.data(), data, count * dim * 4);
            else QAMD_TRY(copy_out(centroids.data(), QAMD_MEM_HOST, data, count * dim * 4, s));
        }
    } else {
        QAMD_TRY(train_centroids(&tmp, data, data_mem, stop, stop_user, s));
        centroids = tmp.centroids_host;
    }
    if (iterations) *iterations = tmp.kmeans_iterations;
    if (empties) *empties = tmp.kmeans_empty_clusters;
    return QAMD_OK;
}

}  // namespace qamd

extern "C" qamd_status qamd_pq_find_centroids(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                                              uint64_t chunk_size, uint32_t max_kmeans_threads, qamd_stop_fn stop,
                                              void *stop_user, void *stream, float *centroids, uint32_t *iterations,
                                              uint32_t *empty_clusters) {
    if (!vp || !centroids) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (chunk_size == 0) return fail(QAMD_ERR_ARGUMENTS, "chunk_size must be > 0");
    if (vp->count > 0 && vp->dim > 0 && !data) return fail(QAMD_ERR_ARGUMENTS, "data is null");
    QAMD_ON_DEVICE(current_device());
    std::vector<float> cen;
    QAMD_TRY(qamd::pq_train_centroids(data, data_mem, vp, chunk_size, max_kmeans_threads, stop, stop_user, as_stream(stream), cen,
                                      iterations, empty_clusters));
    memcpy(centroids, cen.data(), cen.size() * sizeof(float));
    return QAMD_OK;
}
