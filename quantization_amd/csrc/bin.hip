// Binary quantizer on MI355X (gfx950): sign-bit packing and the XOR+popcount scan.
//
// Host side mirrors EncodedVectorsBin (quantization/src/encoded_vectors_binary.rs); the scan
// replaces impl_xor_popcnt_sse_uint{128,64,32} (quantization/cpp/sse.c:49-106), which the
// reference calls once per (query, row) pair.
//
// HBM layout: rows keep the reference's byte layout (bit i of a vector = byte i/8, bit i%8,
// identical for the u8 and u128 store types on little-endian, :193-208).  Device stride `ds`
// is the reference's row size for rows >= 8 bytes (always a multiple of 8; of 16 above 128
// dims, e.g. 128 B at dim 1024) and 4 bytes for the tiny rows (dim <= 32, 0..4 bytes).  Pad
// bits are zero in rows and query alike, so they never reach the popcount (:36-37).
//
// Scan mapping: as the u8 scan — G lanes read one row in 16-byte pieces, one wave-load covers
// 64/G consecutive rows — with v_xor + v_bcnt_u32_b32 instead of v_dot4.  All integer, exact;
// the f32 metric (:219-253) is exact for dim < 2^23.
//
// Many queries at once run on the matrix cores, where popcount(q AND v) is a dot product of 0/1 values: from 12 queries
// bin_gemm_rs_kernel (bits expanded to bytes in registers, int8 MFMA), from 129 queries on rows of 512 / 768 / 1024 / 1536
// bits bin_gemm_qs4_kernel (bits expanded once per row block to E2M1 nibbles in LDS, FP4 MFMA with exact f32 counts, the
// batch streamed past the block).  Both end in the reference's calculate_metric on the integer count: bit-identical.
#include <algorithm>
#include <memory>
#include <vector>

#include <mutex>

#include "batch_common.hpp"
#include "common.hpp"
#include "lists.hpp"
#include "multi_reduce.hpp"
#include "topk.hpp"
#include "topk_device.hpp"

#pragma clang fp contract(off)

using namespace qamd;

namespace {

constexpr int kBlock = 256;
constexpr int kScanBlock = 512;
constexpr uint64_t kRowPad = 1024;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
    u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ uint32_t xpop16(const uint4 &a, const uint4 &b, uint32_t acc) {
    acc += __popc(a.x ^ b.x);
    acc += __popc(a.y ^ b.y);
    acc += __popc(a.z ^ b.z);
    acc += __popc(a.w ^ b.w);
    return acc;
}

// Sum over the G (<= 16) adjacent lanes of a row group, result in every lane.  DPP only
// (quad_perm xor-1 / xor-2, row_half_mirror, row_mirror): four VALU adds for G = 16, instead
// of the ds_bpermute round trips __shfl_xor compiles to.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int G> __device__ __forceinline__ uint32_t group_sum(uint32_t v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "row group is at most one DPP row");
    if (G >= 2) v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    if (G >= 4) v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    if (G >= 8) v = dpp_add<0x141>(v);  // row_half_mirror
    if (G >= 16) v = dpp_add<0x140>(v); // row_mirror
    return v;
}

// encoded_vectors_binary.rs:237-252 calculate_metric
__device__ __forceinline__ float metric(uint32_t x, float dim_f, int is_dot, int invert) {
    const float xor_product = (float)x;
    const float zeros_count = dim_f - xor_product;
    const bool zx = (is_dot != 0) != (invert != 0);  // Dot,!invert and L1/L2,invert -> zeros - xor
    return zx ? zeros_count - xor_product : xor_product - zeros_count;
}

// Rows of ds >= 16 bytes (row_chunks = ds/16).  One wave per tile of (64/G)*UNROLL rows,
// non-persistent grid (same reasoning and measurements as u8_scan_kernel in u8.hip).
template <int G, int ITERS, int UNROLL, bool EXACT, bool FILTER>
__global__ __launch_bounds__(kScanBlock) void bin_scan_kernel(const uint4 *__restrict__ rows,
                                                             const uint4 *__restrict__ qbits, float dim_f,
                                                             int is_dot, int invert, uint32_t n_rows,
                                                             uint32_t row_chunks, float *__restrict__ out,
                                                             TopkFilter filt) {
    constexpr int RW = 64 / G;
    constexpr int TILE = RW * UNROLL;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, rslot = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) >> 6;
    const uint64_t base = wave * TILE;
    if (base >= n_rows) return;
    uint4 q[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t c = sub + it * G;
        if (EXACT) {
            q[it] = qbits[c];
        } else {
            const uint4 t = qbits[c < row_chunks ? c : row_chunks - 1];
            const bool in = c < row_chunks;
            q[it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
        }
    }
    uint4 v[UNROLL][ITERS];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const uint64_t row = base + u * RW + rslot;
        const uint4 *p = rows + row * row_chunks;
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint32_t c = sub + it * G;
            if (EXACT) {
                v[u][it] = ld_nt(p + c);
            } else {  // masked lane: read the row's last chunk, then xor against itself -> 0 bits
                const bool in = c < row_chunks;
                uint4 t = ld_nt(p + (in ? c : row_chunks - 1));
                v[u][it] = in ? t : q[it];
            }
        }
    }
    float mine = 0.0f;  // lane (rslot, sub = u % G) keeps row u's score: one coalesced nt store per G rows-groups
    uint32_t pivot = 0;
    if (FILTER) pivot = *filt.pivot_key;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        uint32_t acc = 0;
#pragma unroll
        for (int it = 0; it < ITERS; it++) acc = xpop16(v[u][it], q[it], acc);
        acc = group_sum<G>(acc);
        if (sub == (u % G)) mine = metric(acc, dim_f, is_dot, invert);
        if ((u % G) == G - 1 || u == UNROLL - 1) {
            const int first = (u / G) * G;
            const uint64_t row = base + (uint64_t)(first + sub) * RW + rslot;
            if (sub <= u - first && row < n_rows) {
                if (FILTER) topk_offer(filt, pivot, mine, (uint32_t)row);
                else __builtin_nontemporal_store(mine, out + row);
            }
        }
    }
}

// NQ queries per row read (qamd_bin_score_batch): the row's 16-byte pieces are loaded once and
// xor-popcounted against NQ query rows held in registers, so the HBM bytes per (query, row) pair fall
// from ds + 4 to ds / NQ + 4 and the scan turns from HBM-bound into popcount-bound.
// Lane (row slot, sub = j) keeps query j's score of the row: one store instruction per row slot
// writes NQ segments of 64/G consecutive floats.  G >= NQ.
// FILTER: no score is written; lane (row slot, sub = j) offers its row to query j's candidate lists
// (fused top-k of NQ queries in one pass: qamd_bin_topk_batch).
template <int G, int ITERS, int UNROLL, int NQ, bool EXACT, bool FILTER>
__global__ __launch_bounds__(kScanBlock) void bin_scan_multi_kernel(const uint4 *__restrict__ rows,
                                                                   const uint4 *__restrict__ qbits /* [NQ][q_stride] */,
                                                                   uint32_t q_stride, float dim_f, int is_dot, int invert,
                                                                   uint32_t n_rows, uint32_t row_chunks,
                                                                   float *__restrict__ out /* [NQ][out_pitch] */,
                                                                   uint64_t out_pitch, TopkFilterSlices slices) {
    static_assert(G >= NQ, "one lane of the row group per query");
    constexpr int RW = 64 / G;
    constexpr int TILE = RW * UNROLL;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, rslot = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) >> 6;
    const uint64_t base = wave * TILE;
    if (base >= n_rows) return;
    uint4 q[NQ][ITERS];
#pragma unroll
    for (int j = 0; j < NQ; j++) {
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint32_t c = sub + it * G;
            const bool in = EXACT || c < row_chunks;
            const uint4 t = qbits[(size_t)j * q_stride + (in ? c : row_chunks - 1)];
            q[j][it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
        }
    }
    uint4 v[UNROLL][ITERS];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const uint64_t row = base + u * RW + rslot;
        const uint4 *p = rows + row * row_chunks;
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint32_t c = sub + it * G;
            const bool in = EXACT || c < row_chunks;
            const uint4 t = ld_nt(p + (in ? c : row_chunks - 1));
            v[u][it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
        }
    }
    // lane (row slot, sub) ends up with the total of query multi_query_of<NQ>(sub) (multi_reduce.hpp):
    // the lanes whose sub is below NQ own one query each
    const int my_q = multi_query_of<NQ>(sub);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        uint32_t acc[NQ];
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            acc[j] = 0;
#pragma unroll
            for (int it = 0; it < ITERS; it++) acc[j] = xpop16(v[u][it], q[j][it], acc[j]);
        }
        const float mine = metric(multi_reduce<G, NQ>(acc, sub), dim_f, is_dot, invert);
        const uint64_t row = base + (uint64_t)u * RW + rslot;
        if (sub < NQ && row < n_rows) {
            if (FILTER) {
                const TopkFilter f = topk_filter_of(slices, (uint32_t)my_q);
                topk_offer(f, *f.pivot_key, mine, (uint32_t)row);  // the pivot is an L2-resident word, re-read per row slot
            } else {
                __builtin_nontemporal_store(mine, out + (uint64_t)my_q * out_pitch + row);
            }
        }
    }
}

// Single-launch top-k for small stores (topk.hpp small_topk; the scalar-u8 twin in u8.hip explains
// the structure): workgroup b scores rows [b * rows_per_wg, (b + 1) * rows_per_wg) with
// bin_scan_kernel's arithmetic; scores become 64-bit keys in the wave's LDS staging row and the
// wave / workgroup / last-arriver merges of topk_device.hpp keep the best k.  Exact under any
// number of ties (keys are distinct: score bits << 32 | row), which the sampled-pivot path is not
// good at for binary scores (few distinct values -> candidate lists overflow -> classic fallback).
template <int G, int ITERS, bool EXACT>
__global__ __launch_bounds__(1024) void bin_topk_small_kernel(const uint4 *__restrict__ rows,
                                                              const uint4 *__restrict__ qbits, float dim_f, int is_dot,
                                                              int invert, uint32_t n_rows, uint32_t row_chunks,
                                                              uint32_t rows_per_wg, SmallTopk p) {
    __shared__ unsigned long long lds[2 * kSmallTopkWaves][64];
    unsigned long long(*lists)[64] = lds;
    constexpr int RW = 64 / G;
    constexpr int UNROLL = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, rslot = lane / G;
    unsigned long long *stage = lds[kSmallTopkWaves + wave];
    uint4 q[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t c = sub + it * G;
        const bool in = EXACT || c < row_chunks;
        const uint4 t = qbits[in ? c : row_chunks - 1];
        q[it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
    }
    const uint64_t wg_base = (uint64_t)blockIdx.x * rows_per_wg;
    SmallTopkWave acc_list;
    for (uint32_t tile = wave * UNROLL; tile * RW < rows_per_wg; tile += kSmallTopkWaves * UNROLL) {
        uint4 v[UNROLL][ITERS];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint64_t row = wg_base + (uint64_t)(tile + u) * RW + rslot;
            const uint64_t rc = row < n_rows ? row : (uint64_t)n_rows - 1;
            const uint4 *src = rows + rc * row_chunks;
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const uint32_t c = sub + it * G;
                const bool in = EXACT || c < row_chunks;
                const uint4 t = ld_nt(src + (in ? c : row_chunks - 1));
                v[u][it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t local = (tile + u) * RW + rslot;
            const uint64_t row = wg_base + local;
            uint32_t acc = 0;
#pragma unroll
            for (int it = 0; it < ITERS; it++) acc = xpop16(v[u][it], q[it], acc);
            acc = group_sum<G>(acc);
            if (sub == 0) {
                unsigned long long key = ~0ull;
                if (local < rows_per_wg && row < n_rows)
                    key = ((unsigned long long)topk_ordered_bits(metric(acc, dim_f, is_dot, invert), p.largest != 0) << 32) |
                          (uint32_t)row;
                stage[acc_list.fill + rslot] = key;
            }
            acc_list.fill += RW;
            if (acc_list.fill == 64) small_topk_flush(acc_list, stage, lane);
        }
    }
    if (acc_list.fill) small_topk_flush(acc_list, stage, lane);
    small_topk_finish(acc_list.best, lists, p);
}

// Any row size, dword granularity: used for tiny rows (ds 4 or 8), very long rows and the
// random-access entry points.  ids == nullptr scans rows [0, n).
__global__ __launch_bounds__(kBlock) void bin_words_kernel(const uint32_t *__restrict__ rows,
                                                          const uint32_t *qbits, float dim_f, int is_dot,
                                                          int invert, const uint32_t *__restrict__ ids,
                                                          uint64_t n, uint32_t n_rows, uint32_t row_words,
                                                          float *__restrict__ out) {
    constexpr int G = 16, RW = 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, rslot = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * kBlock) >> 6;
    for (uint64_t base = wave * RW; base < n; base += n_waves * RW) {
        const uint64_t k = base + rslot;
        const uint32_t row = k < n ? (ids ? ids[k] : (uint32_t)k) : 0xFFFFFFFFu;
        const bool ok = row < n_rows;
        uint32_t acc = 0;
        if (ok) {
            const uint32_t *p = rows + (uint64_t)row * row_words;
            for (uint32_t w = sub; w < row_words; w += G) acc += __popc(p[w] ^ qbits[w]);
        }
        acc = group_sum<G>(acc);
        if (sub == 0 && k < n) out[k] = ok ? metric(acc, dim_f, is_dot, invert) : __builtin_nanf("");
    }
}

// Random access for bursts of pairs (lists.hpp): out[k] = metric(row ids[k] xor the query of pair k).
// Which query: `lists` == nullptr -> q_single for every pair (it may be a stored row: score_internal is
// the same metric on two stored rows, encoded_vectors_binary.rs:302-314); lists, list_rows == nullptr ->
// bit row l of a query batch for the pairs of list l; lists and list_rows -> stored row list_rows[l].
// VEC16 (ds % 16 == 0): 8 lanes per pair, 16-byte pieces, up to four in flight per lane; else 16 lanes
// per pair at dword granularity (rows of 4 and 8 bytes).
template <bool VEC16>
__global__ __launch_bounds__(kBlock) void bin_pairs_kernel(const uint32_t *__restrict__ rows, const uint32_t *q_single,
                                                          const uint8_t *__restrict__ q_batch, uint32_t q_stride,
                                                          const uint32_t *__restrict__ lists, uint32_t n_lists,
                                                          const uint32_t *__restrict__ list_rows, float dim_f, int is_dot,
                                                          int invert, const uint32_t *__restrict__ ids, uint64_t n,
                                                          uint32_t n_rows, uint32_t row_words, uint32_t pairs_per_block,
                                                          float *__restrict__ out) {
    constexpr int G = VEC16 ? 8 : 16, GROUPS = kBlock / G;
    __shared__ uint32_t first_list;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, group = threadIdx.x / G;
    const uint64_t p0 = (uint64_t)blockIdx.x * pairs_per_block;
    const uint64_t p1 = p0 + pairs_per_block < n ? p0 + pairs_per_block : n;
    uint32_t l = lists ? first_list_of_block(lists, n_lists, p0, &first_list) : 0u;  // (lists.hpp)
    for (uint64_t k = p0 + group; k < p1; k += GROUPS) {
        const uint32_t row = ids[k];
        bool ok = row < n_rows;
        const uint32_t *qp = q_single;
        if (lists) {
            l = advance_list(lists, n_lists, l, k);
            if (list_rows) {
                const uint32_t qr = list_rows[l];
                ok = ok && qr < n_rows;
                qp = rows + (uint64_t)(qr < n_rows ? qr : 0u) * row_words;
            } else {
                qp = reinterpret_cast<const uint32_t *>(q_batch + (size_t)l * q_stride);
            }
        }
        const uint32_t *p = rows + (uint64_t)(ok ? row : 0u) * row_words;
        uint32_t acc = 0;
        if (VEC16) {
            const uint32_t chunks = row_words / 4;
            const uint4 *p4 = reinterpret_cast<const uint4 *>(p), *q4 = reinterpret_cast<const uint4 *>(qp);
            if (chunks <= (uint32_t)G) {  // the common rows (<= 1024 bits): one 16-byte piece per lane
                if ((uint32_t)sub < chunks) acc = xpop16(p4[sub], q4[sub], acc);
            } else {
                for (uint32_t c0 = sub; c0 < chunks; c0 += 4 * G) {
                    uint4 v[4], qv[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t c = c0 + j * G, cc = c < chunks ? c : chunks - 1;
                        v[j] = p4[cc];
                        qv[j] = q4[cc];
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (c0 + j * G < chunks) acc = xpop16(v[j], qv[j], acc);
                }
            }
        } else {
            for (uint32_t w = sub; w < row_words; w += G) acc += __popc(p[w] ^ qp[w]);
        }
        acc = group_sum<G>(acc);
        if (sub == 0) out[k] = ok ? metric(acc, dim_f, is_dot, invert) : __builtin_nanf("");
    }
}

// encode_vector (:193-208): one wave per row, 64 elements per ballot; lane (step % 64) keeps
// the ballot of `step`, so after <= 64 steps every lane owns 8 output bytes (coalesced store).
__global__ __launch_bounds__(kBlock) void bin_encode_kernel(const float *__restrict__ data, uint64_t n_rows,
                                                           uint32_t dim, uint32_t row_words,
                                                           uint32_t *__restrict__ rows, uint64_t row0) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * kBlock) >> 6;
    const uint32_t steps = (dim + 63) / 64;
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const float *src = data + r * dim;
        uint32_t *dst = rows + (row0 + r) * row_words;
        unsigned long long mine = 0;
        for (uint32_t st = 0; st < steps; st++) {
            const uint32_t j = st * 64 + lane;
            const float v = j < dim ? src[j] : 0.0f;
            const unsigned long long b = __ballot(v > 0.0f);
            if ((st & 63u) == (uint32_t)lane) mine = b;
            if ((st & 63u) == 63u || st + 1 == steps) {
                const uint32_t w = ((st & ~63u) + lane) * 2;  // this lane's first dword
                if (w < row_words) dst[w] = (uint32_t)mine;
                if (w + 1 < row_words) dst[w + 1] = (uint32_t)(mine >> 32);
                mine = 0;
            }
        }
    }
}

// encode_vector, streaming form for stores whose rows are whole 128-bit words with no row
// padding (dim % 128 == 0): the batch is one flat stream of floats -> bits.  One wave per
// 16 KiB tile; wave-load j reads 64 consecutive float4 (nt), each lane turns its four signs
// into a nibble at bit 4*(lane%8), a 3-step DPP add over the 8-lane group (disjoint fields, so
// add == or) hands every lane of the group the finished dword, and lane 8g+t keeps the one of
// wave-load t (and t+8): after 16 loads a lane owns two dwords and the wave writes 2 x 256 B.
__global__ __launch_bounds__(kBlock) void bin_encode_flat_kernel(const float4 *__restrict__ d4, uint64_t n4,
                                                                uint32_t *__restrict__ words /* n4/8 dwords */) {
    const int lane = threadIdx.x & 63, t = lane & 7, g = lane >> 3;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t base = wave * 1024;
    if (base >= n4) return;
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint64_t i = base + j * 64 + lane;
        v[j] = i < n4 ? __builtin_bit_cast(float4, ld_nt(reinterpret_cast<const uint4 *>(d4) + i))
                      : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    uint32_t keep[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 16; j++) {
        uint32_t nib = (v[j].x > 0.0f ? 1u : 0u) | (v[j].y > 0.0f ? 2u : 0u) | (v[j].z > 0.0f ? 4u : 0u) |
                       (v[j].w > 0.0f ? 8u : 0u);
        const uint32_t dw = group_sum<8>(nib << (4 * t));
        if ((j & 7) == t) keep[j >> 3] = dw;
    }
    const uint64_t n_words = n4 / 8;
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const uint64_t w = wave * 128 + hf * 64 + t * 8 + g;  // wave-load (hf*8 + t), group g
        if (w < n_words) words[w] = keep[hf];
    }
}

int grid_for(uint64_t work_items, uint64_t per_block, int blocks_per_cu) {
    uint64_t want = (work_items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)device_info().cu_count * blocks_per_cu;
    if (want < 1) want = 1;
    return (int)(want > cap ? cap : want);
}

// get_storage_size * size_of (:99-116 u8, :152-159 u128, :210-213)
uint64_t row_bytes_of(uint64_t dim, int store) {
    if (store == QAMD_BITS_U128) return (dim / 128 + (dim % 128 != 0)) * 16;
    const uint64_t bytes_count = dim > 128 ? 16 : dim > 64 ? 8 : dim > 32 ? 4 : 1;
    const uint64_t bits = 8 * bytes_count;
    return (dim / bits + (dim % bits != 0)) * bytes_count;
}

uint64_t device_stride_of(uint64_t nb) { return nb >= 8 ? nb : 4; }

}  // namespace

struct qamd_bin {
    int device = 0;
    qamd_vector_parameters vp{};
    int store = 0;
    uint64_t count = 0;
    uint64_t nb = 0;  // reference row bytes
    uint64_t ds = 0;  // device row stride (bytes)
    DevBuf rows;      // [padded_rows][ds]
    // the batched top-k's pivot sample (rows hash(j) of the store, j < sample_count, then 512 zero rows);
    // gathered once on first use (count / 64 rows at most), immutable after
    mutable std::mutex sample_mu;
    mutable DevBuf sample_rows;
    mutable uint32_t sample_count = 0;
};

struct qamd_bin_query {
    int device = 0;
    uint64_t nb = 0, ds = 0, qdim_cap = 0;
    DevBuf buf;  // ds bytes (+ padding)
    ReadyEvent ready;  // the last encode_query
};

namespace {

qamd_status alloc_store(qamd_bin *h) {
    h->nb = row_bytes_of(h->vp.dim, h->store);
    h->ds = device_stride_of(h->nb);
    const uint64_t padded = round_up(h->count, kRowPad) + kRowPad;
    return h->rows.alloc(padded * h->ds, true);
}

template <int G, int ITERS, int UNROLL>
void launch_bin(const qamd_bin *h, const uint4 *qb, float *out, const TopkFilter *filt, hipStream_t s) {
    constexpr int TILE = (64 / G) * UNROLL;
    const uint32_t rc = (uint32_t)(h->ds / 16);
    const int is_dot = h->vp.distance_type == QAMD_DOT;
    const uint64_t waves = (h->count + TILE - 1) / TILE;
    const unsigned grid = (unsigned)((waves + kScanBlock / 64 - 1) / (kScanBlock / 64));
    const bool exact = rc == (uint32_t)(G * ITERS);
#define QAMD_BIN_GO(EX, FI)                                                                                  \
    hipLaunchKernelGGL((bin_scan_kernel<G, ITERS, UNROLL, EX, FI>), dim3(grid), dim3(kScanBlock), 0, s,      \
                       h->rows.as<uint4>(), qb, (float)h->vp.dim, is_dot, h->vp.invert, (uint32_t)h->count, rc, \
                       out, filt ? *filt : TopkFilter{})
    if (filt) {
        if (exact) QAMD_BIN_GO(true, true);
        else QAMD_BIN_GO(false, true);
    } else {
        if (exact) QAMD_BIN_GO(true, false);
        else QAMD_BIN_GO(false, false);
    }
#undef QAMD_BIN_GO
}

qamd_status words_launch(const qamd_bin *h, const uint32_t *qbits, const uint32_t *ids_dev, uint64_t n,
                         float *out_dev, hipStream_t s) {
    if (n == 0) return QAMD_OK;
    int grid = grid_for((n + 3) / 4, kBlock / 64, 8);
    hipLaunchKernelGGL(bin_words_kernel, dim3(grid), dim3(kBlock), 0, s, h->rows.as<uint32_t>(), qbits,
                       (float)h->vp.dim, (int)(h->vp.distance_type == QAMD_DOT), h->vp.invert, ids_dev, n,
                       (uint32_t)h->count, (uint32_t)(h->ds / 4), out_dev);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// One launch of bin_pairs_kernel (lists == nullptr: the single query bit row for every id).
qamd_status pairs_launch(const qamd_bin *h, const uint32_t *q_single, const uint8_t *q_batch, uint64_t q_stride,
                         const uint32_t *lists, uint32_t n_lists, const uint32_t *list_rows, const uint32_t *ids_dev,
                         uint64_t n, float *out_dev, hipStream_t s) {
    if (n == 0) return QAMD_OK;
    const bool vec16 = h->ds % 16 == 0;
    const uint32_t ppb = pairs_per_block(n, vec16 ? 32 : 16);  // lane groups per workgroup
    const unsigned grid = (unsigned)((n + ppb - 1) / ppb);
#define QAMD_BIN_PAIRS(V)                                                                                          \
    hipLaunchKernelGGL(bin_pairs_kernel<V>, dim3(grid), dim3(kBlock), 0, s, h->rows.as<uint32_t>(), q_single, q_batch, \
                       (uint32_t)q_stride, lists, n_lists, list_rows, (float)h->vp.dim,                            \
                       (int)(h->vp.distance_type == QAMD_DOT), h->vp.invert, ids_dev, n, (uint32_t)h->count,        \
                       (uint32_t)(h->ds / 4), ppb, out_dev)
    if (vec16) QAMD_BIN_PAIRS(true);
    else QAMD_BIN_PAIRS(false);
#undef QAMD_BIN_PAIRS
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// out[k] = metric(query bit row `qbits`, row ids[k]) for host or device ids / outputs: the body of
// score_ids and, with qbits = stored row i, of score_internal_ids.
qamd_status score_ids_any(const qamd_bin *h, const uint32_t *qbits, const uint32_t *ids, uint64_t n_ids, qamd_mem ids_mem,
                          float *out, qamd_mem out_mem, hipStream_t s) {
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
        QAMD_TRY(pairs_launch(h, qbits, nullptr, 0, nullptr, 0, nullptr, hs.dev, n_ids, reinterpret_cast<float *>(hs.dev + 1024), s));
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
    QAMD_TRY(pairs_launch(h, qbits, nullptr, 0, nullptr, 0, nullptr, ids_dev, n_ids, out_dev, s));
    if (out_mem == QAMD_MEM_HOST) QAMD_TRY(copy_out(out, QAMD_MEM_HOST, out_dev, n_ids * 4, s));
    else if (ids_mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));
    return QAMD_OK;
}

bool fused_capable(const qamd_bin *h) { return h->ds % 16 == 0 && h->ds / 16 <= 64; }

qamd_status scan_bits(const qamd_bin *h, const void *qbits_dev, float *out_dev, hipStream_t s,
                      const TopkFilter *filt = nullptr) {
    if (h->count == 0) return QAMD_OK;
    const uint32_t rc = (uint32_t)(h->ds / 16);
    if (!fused_capable(h)) return words_launch(h, static_cast<const uint32_t *>(qbits_dev), nullptr, h->count, out_dev, s);
    const uint4 *qb = static_cast<const uint4 *>(qbits_dev);
    if (rc == 1) launch_bin<1, 1, 4>(h, qb, out_dev, filt, s);
    else if (rc == 2) launch_bin<2, 1, 4>(h, qb, out_dev, filt, s);
    else if (rc <= 4) launch_bin<4, 1, 8>(h, qb, out_dev, filt, s);
    else if (rc <= 8) launch_bin<8, 1, 16>(h, qb, out_dev, filt, s);
    else if (rc <= 16) launch_bin<16, 1, 8>(h, qb, out_dev, filt, s);
    else if (rc <= 32) launch_bin<16, 2, 4>(h, qb, out_dev, filt, s);
    else if (rc <= 48) launch_bin<16, 3, 4>(h, qb, out_dev, filt, s);
    else launch_bin<16, 4, 2>(h, qb, out_dev, filt, s);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

qamd_status scan_into(const qamd_bin *h, const qamd_bin_query *q, float *out_dev, hipStream_t s,
                      const TopkFilter *filt = nullptr) {
    return scan_bits(h, q->buf.ptr, out_dev, s, filt);
}

template <int G, int ITERS>
qamd_status launch_bin_small(const qamd_bin *h, const uint4 *qb, const SmallTopkPlan &pl, const SmallTopk &p, hipStream_t s) {
    const uint32_t rc = (uint32_t)(h->ds / 16);
    const bool exact = rc == (uint32_t)(G * ITERS);
#define QAMD_BIN_SMALL(EX)                                                                                         \
    hipLaunchKernelGGL((bin_topk_small_kernel<G, ITERS, EX>), dim3(pl.workgroups), dim3(1024), 0, s, h->rows.as<uint4>(), \
                       qb, (float)h->vp.dim, (int)(h->vp.distance_type == QAMD_DOT), h->vp.invert, (uint32_t)h->count, rc, \
                       pl.rows_per_wg, p)
    if (exact) QAMD_BIN_SMALL(true);
    else QAMD_BIN_SMALL(false);
#undef QAMD_BIN_SMALL
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

int bin_small_group(uint32_t rc) { return rc == 1 ? 1 : rc == 2 ? 2 : rc <= 4 ? 4 : rc <= 8 ? 8 : rc <= 64 ? 16 : 0; }

qamd_status launch_small(const qamd_bin *h, const uint4 *qb, const SmallTopkPlan &pl, const SmallTopk &p, hipStream_t s) {
    const uint32_t rc = (uint32_t)(h->ds / 16);
    if (rc == 1) return launch_bin_small<1, 1>(h, qb, pl, p, s);
    if (rc == 2) return launch_bin_small<2, 1>(h, qb, pl, p, s);
    if (rc <= 4) return launch_bin_small<4, 1>(h, qb, pl, p, s);
    if (rc <= 8) return launch_bin_small<8, 1>(h, qb, pl, p, s);
    if (rc <= 16) return launch_bin_small<16, 1>(h, qb, pl, p, s);
    if (rc <= 32) return launch_bin_small<16, 2>(h, qb, pl, p, s);
    if (rc <= 48) return launch_bin_small<16, 3>(h, qb, pl, p, s);
    return launch_bin_small<16, 4>(h, qb, pl, p, s);
}

// The single-launch top-k when the store qualifies (<= 2M rows, k <= 64, 16-byte row pieces): false = not applicable.
bool bin_topk_small(const qamd_bin *h, const void *qbits_dev, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                    qamd_mem out_mem, hipStream_t s, qamd_status &st) {
    const int g = bin_small_group((uint32_t)(h->ds / 16));
    SmallTopkPlan plan;
    if (!g || !fused_capable(h)) return false;
    const uint32_t tile = 2 * (64 / g), least = (uint32_t)std::max<uint64_t>(16 * tile, (128 * 1024) / h->ds);
    if (!small_topk_plan(h->count, k, tile, least, plan)) return false;
    st = small_topk(plan, k, largest, out_ids, out_scores, out_mem, s, [&](const SmallTopk &p, hipStream_t stt) {
        return launch_small(h, static_cast<const uint4 *>(qbits_dev), plan, p, stt);
    });
    return true;
}

qamd_status check_query(const qamd_bin *h, const qamd_bin_query *q) {
    if (!h || !q) return fail(QAMD_ERR_ARGUMENTS, "null handle or query");
    if (q->nb != h->nb) return fail(QAMD_ERR_ARGUMENTS, "query has %llu bytes, rows have %llu",
                                    (unsigned long long)q->nb, (unsigned long long)h->nb);
    return QAMD_OK;
}

// Host rows at stride nb -> device stride ds (differs only for the tiny rows).
qamd_status upload_rows(qamd_bin *h, const uint8_t *rows, qamd_mem mem, hipStream_t s) {
    if (h->count == 0) return QAMD_OK;
    if (h->nb == h->ds) return copy_in(h->rows.ptr, rows, mem, h->count * h->nb, s);
    std::vector<uint8_t> host(h->count * h->nb), wide(h->count * h->ds, 0);
    if (h->nb) {
        if (mem == QAMD_MEM_DEVICE) QAMD_TRY(copy_out(host.data(), QAMD_MEM_HOST, rows, host.size(), s));
        else memcpy(host.data(), rows, host.size());
    }
    for (uint64_t r = 0; r < h->count; r++) memcpy(&wide[r * h->ds], &host[r * h->nb], h->nb);
    return copy_in(h->rows.ptr, wide.data(), QAMD_MEM_HOST, wide.size(), s);
}

// encode_vector (:193-208) for `nr` device-resident rows: rows r0 .. r0+nr of the store.
qamd_status launch_bin_encode(qamd_bin *h, const float *src, uint64_t nr, uint64_t r0, hipStream_t s) {
    const uint64_t dim = h->vp.dim;
    if (nr == 0 || dim == 0) return QAMD_OK;
    if (dim % 128 == 0 && h->ds * 8 == dim && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint64_t n4 = nr * dim / 4;
        const unsigned grid = (unsigned)((n4 + 1024 * (kBlock / 64) - 1) / (1024 * (kBlock / 64)));
        hipLaunchKernelGGL(bin_encode_flat_kernel, dim3(grid), dim3(kBlock), 0, s, reinterpret_cast<const float4 *>(src),
                           n4, h->rows.as<uint32_t>() + r0 * (h->ds / 4));
    } else {
        int grid = grid_for(nr, kBlock / 64, 8);
        hipLaunchKernelGGL(bin_encode_kernel, dim3(grid), dim3(kBlock), 0, s, src, nr, (uint32_t)dim,
                           (uint32_t)(h->ds / 4), h->rows.as<uint32_t>(), r0);
    }
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

}  // namespace

extern "C" {

uint64_t qamd_bin_quantized_vector_size(const qamd_vector_parameters *vp, qamd_bits_store store) {
    return row_bytes_of(vp->dim, store);
}

qamd_status qamd_bin_encode(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                            qamd_bits_store store, qamd_stop_fn stop, void *stop_user, void *stream,
                            qamd_bin **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    if (vp->count > 0 && vp->dim > 0 && !data) return fail(QAMD_ERR_ARGUMENTS, "data is null");
    QAMD_ON_DEVICE(current_device());
    hipStream_t s = as_stream(stream);
    std::unique_ptr<qamd_bin> h(new qamd_bin);
    h->device = current_device();
    h->vp = *vp;
    h->store = store;
    h->count = vp->count;
    QAMD_TRY(alloc_store(h.get()));
    const uint64_t dim = vp->dim, count = vp->count;
    if (count && dim) {
        // host rows are staged 256 MiB at a time; device rows are read in place, 8 GiB per launch
        const uint64_t batch_bytes = data_mem == QAMD_MEM_HOST ? (256ull << 20) : (8ull << 30);
        const uint64_t batch_rows = std::max<uint64_t>(1, std::min<uint64_t>(count, batch_bytes / (dim * 4)));
        DevBuf stage;
        if (data_mem == QAMD_MEM_HOST) QAMD_TRY(stage.alloc(batch_rows * dim * 4));
        for (uint64_t r0 = 0; r0 < count; r0 += batch_rows) {
            if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :174-176
            const uint64_t nr = std::min(batch_rows, count - r0);
            const float *src = data + r0 * dim;
            if (data_mem == QAMD_MEM_HOST) {
                QAMD_TRY(copy_in(stage.ptr, src, QAMD_MEM_HOST, nr * dim * 4, s));
                src = stage.as<float>();
            }
            QAMD_TRY(launch_bin_encode(h.get(), src, nr, r0, s));
            if (data_mem == QAMD_MEM_HOST || stop) QAMD_HIP(hipStreamSynchronize(s));
        }
        QAMD_HIP(hipStreamSynchronize(s));
    } else if (stop && count && stop(stop_user)) {
        return fail(QAMD_ERR_STOPPED, "Stopped");
    }
    *out = h.release();
    return QAMD_OK;
}

qamd_status qamd_bin_from_rows(const uint8_t *rows, qamd_mem rows_mem, const qamd_vector_parameters *vp,
                               qamd_bits_store store, void *stream, qamd_bin **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    QAMD_ON_DEVICE(current_device());
    std::unique_ptr<qamd_bin> h(new qamd_bin);
    h->device = current_device();
    h->vp = *vp;
    h->store = store;
    h->count = vp->count;
    QAMD_TRY(alloc_store(h.get()));
    if (h->count && h->nb && !rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_TRY(upload_rows(h.get(), rows, rows_mem, as_stream(stream)));
    QAMD_HIP(hipStreamSynchronize(as_stream(stream)));
    *out = h.release();
    return QAMD_OK;
}

// Rows [first_row, first_row + n_rows) as the reference's storage holds them (push_vector_data,
// encoded_storage.rs:17-25), in bounded pieces.
qamd_status qamd_bin_export_rows_range(const qamd_bin *h, uint64_t first_row, uint64_t n_rows, uint8_t *rows,
                                       qamd_mem rows_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (first_row > h->count || n_rows > h->count - first_row)
        return fail(QAMD_ERR_OUT_OF_RANGE, "rows [%llu, +%llu) out of range (count %llu)", (unsigned long long)first_row,
                    (unsigned long long)n_rows, (unsigned long long)h->count);
    if (n_rows == 0 || h->nb == 0) return QAMD_OK;
    if (!rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint8_t *src = h->rows.as<uint8_t>() + first_row * h->ds;
    if (h->nb == h->ds) return copy_out(rows, rows_mem, src, n_rows * h->nb, s);
    // rows of <= 32 dims are held at a 4-byte stride on the device: re-stride on the way out
    std::vector<uint8_t> wide(n_rows * h->ds), host(n_rows * h->nb);
    QAMD_TRY(copy_out(wide.data(), QAMD_MEM_HOST, src, wide.size(), s));
    for (uint64_t r = 0; r < n_rows; r++) memcpy(&host[r * h->nb], &wide[r * h->ds], h->nb);
    if (rows_mem == QAMD_MEM_HOST) memcpy(rows, host.data(), host.size());
    else QAMD_TRY(copy_in(rows, host.data(), QAMD_MEM_HOST, host.size(), s));
    return QAMD_OK;
}

qamd_status qamd_bin_export_rows(const qamd_bin *h, uint8_t *rows, qamd_mem rows_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return qamd_bin_export_rows_range(h, 0, h->count, rows, rows_mem, stream);
}

// save/load (:260-286): Metadata{vector_parameters} as serde_json + raw row bytes.
qamd_status qamd_bin_save(const qamd_bin *h, const char *data_path, const char *meta_path) {
    if (!h || !data_path || !meta_path) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::string js = "{\"vector_parameters\":" + vector_parameters_json(h->vp) + "}";
    make_parent_dirs(meta_path);
    if (!write_file(meta_path, js.data(), js.size())) return fail(QAMD_ERR_IO, "cannot write %s", meta_path);
    std::vector<uint8_t> rows(h->count * h->nb);
    QAMD_TRY(qamd_bin_export_rows(h, rows.data(), QAMD_MEM_HOST, nullptr));
    make_parent_dirs(data_path);
    if (!write_file(data_path, rows.data(), rows.size())) return fail(QAMD_ERR_IO, "cannot write %s", data_path);
    return QAMD_OK;
}

qamd_status qamd_bin_load(const char *data_path, const char *meta_path, const qamd_vector_parameters *vp,
                          qamd_bits_store store, qamd_bin **out) {
    if (!data_path || !meta_path || !vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    JsonValue root;
    QAMD_TRY(read_metadata(meta_path, root));
    qamd_vector_parameters file_vp{};
    {   // Metadata{vector_parameters} (encoded_vectors_binary.rs:21-24)
        std::string err;
        const JsonValue *vpj = json_field(root, "vector_parameters", err);
        if (!vpj || !parse_vector_parameters(*vpj, file_vp, err)) return fail(QAMD_ERR_IO, "%s: %s", meta_path, err.c_str());
    }
    std::string bytes;
    if (!read_file(data_path, bytes)) return fail(QAMD_ERR_IO, "cannot read %s", data_path);
    const uint64_t expected = row_bytes_of(vp->dim, store) * vp->count;  // :277-279
    if (bytes.size() != expected)
        return fail(QAMD_ERR_IO, "Loaded storage size %zu is not equal to expected size %llu", bytes.size(),
                    (unsigned long long)expected);
    qamd_vector_parameters eff = file_vp;  // metadata rules the metric, the caller's params the sizes
    eff.dim = vp->dim;
    eff.count = vp->count;
    return qamd_bin_from_rows(reinterpret_cast<const uint8_t *>(bytes.data()), QAMD_MEM_HOST, &eff, store, nullptr,
                              out);
}

qamd_status qamd_bin_encode_query(const qamd_bin *h, const float *query, uint64_t qdim, qamd_mem query_mem,
                                  void *stream, qamd_bin_query **query_io) {
    if (!h || !query_io || (!query && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint64_t nb = row_bytes_of(qdim, h->store), ds = device_stride_of(nb);
    qamd_bin_query *q = *query_io;
    std::unique_ptr<qamd_bin_query> fresh;
    if (!q) {
        fresh.reset(new qamd_bin_query);
        q = fresh.get();
        q->device = h->device;
    }
    if (q->ds != ds || q->qdim_cap < qdim || !q->buf.ptr) {
        QAMD_TRY(q->buf.alloc(round_up(ds, 16) + 16 + qdim * 4 + 16, true));  // bits | f32 staging
        q->qdim_cap = qdim;
        q->nb = nb;
        q->ds = ds;
    }
    // Always packed on the device (one implementation); a host query is uploaded first.
    if (qdim) {
        const float *q_dev = query;
        if (query_mem == QAMD_MEM_HOST) {
            float *stage = reinterpret_cast<float *>(q->buf.as<uint8_t>() + round_up(ds, 16) + 16);
            QAMD_TRY(copy_in(stage, query, QAMD_MEM_HOST, qdim * 4, s));
            q_dev = stage;
        }
        hipLaunchKernelGGL(bin_encode_kernel, dim3(1), dim3(kBlock), 0, s, q_dev, (uint64_t)1, (uint32_t)qdim,
                           (uint32_t)(ds / 4), q->buf.as<uint32_t>(), (uint64_t)0);
        QAMD_HIP(hipGetLastError());
    }
    QAMD_TRY(q->ready.record(s));
    if (fresh) *query_io = fresh.release();
    return QAMD_OK;
}

qamd_status qamd_bin_query_read(const qamd_bin_query *q, uint8_t *bits, uint64_t capacity, uint64_t *len) {
    if (!q) return fail(QAMD_ERR_ARGUMENTS, "null query");
    if (len) *len = q->nb;
    if (bits) {
        if (capacity < q->nb) return fail(QAMD_ERR_ARGUMENTS, "bits buffer too small");
        QAMD_ON_DEVICE(q->device);
        QAMD_TRY(q->ready.wait(nullptr));
        QAMD_TRY(copy_out(bits, QAMD_MEM_HOST, q->buf.ptr, q->nb, nullptr));
    }
    return QAMD_OK;
}

void qamd_bin_query_free(qamd_bin_query *q) { delete q; }

qamd_status qamd_bin_score_all(const qamd_bin *h, const qamd_bin_query *q, float *out, qamd_mem out_mem,
                               void *stream) {
    QAMD_TRY(check_query(h, q));
    if (h->count == 0) return QAMD_OK;
    if (!out) return fail(QAMD_ERR_ARGUMENTS, "out is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(q->ready.wait(s));
    if (out_mem == QAMD_MEM_DEVICE) return scan_into(h, q, out, s);
    float *tmp = nullptr;  // per-thread workspace: no hipMalloc / hipFree per query
    QAMD_TRY(thread_ws_acquire(WS_SCORES, h->count * 4, s, reinterpret_cast<void **>(&tmp)));
    qamd_status st = scan_into(h, q, tmp, s);
    if (st == QAMD_OK) st = copy_out(out, QAMD_MEM_HOST, tmp, h->count * 4, s);
    thread_ws_release(WS_SCORES, s, st == QAMD_OK);  // the download synchronised the stream
    return st;
}

qamd_status qamd_bin_score_ids(const qamd_bin *h, const qamd_bin_query *q, const uint32_t *ids, uint64_t n_ids,
                               qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_query(h, q));
    if (n_ids == 0) return QAMD_OK;
    if (!ids || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(q->ready.wait(s));
    return score_ids_any(h, q->buf.as<uint32_t>(), ids, n_ids, ids_mem, out, out_mem, s);
}

qamd_status qamd_bin_score_point(const qamd_bin *h, const qamd_bin_query *q, uint32_t i, float *out) {
    return qamd_bin_score_ids(h, q, &i, 1, QAMD_MEM_HOST, out, QAMD_MEM_HOST, nullptr);
}

// score_internal (:302-314) for one stored row against many: out[k] = score_internal(i, ids[k]).
qamd_status qamd_bin_score_internal_ids(const qamd_bin *h, uint32_t i, const uint32_t *ids, uint64_t n_ids,
                                        qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (n_ids == 0) return QAMD_OK;
    if (!ids || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (i >= h->count) return fail(QAMD_ERR_OUT_OF_RANGE, "row id %u out of range (count %llu)", i, (unsigned long long)h->count);
    QAMD_ON_DEVICE(h->device);
    return score_ids_any(h, h->rows.as<uint32_t>() + (uint64_t)i * (h->ds / 4), ids, n_ids, ids_mem, out, out_mem,
                         as_stream(stream));
}

qamd_status qamd_bin_score_internal(const qamd_bin *h, uint32_t i, uint32_t j, float *out) {
    if (!h || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (j >= h->count) return fail(QAMD_ERR_OUT_OF_RANGE, "row id out of range (count %llu)", (unsigned long long)h->count);
    return qamd_bin_score_internal_ids(h, i, &j, 1, QAMD_MEM_HOST, out, QAMD_MEM_HOST, nullptr);
}

// Many stored rows, each against its own id list, in one launch (lists.hpp).
qamd_status qamd_bin_score_internal_ids_batch(const qamd_bin *h, const uint32_t *rows, const uint32_t *list_offsets,
                                              uint32_t n_lists, const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem,
                                              float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (n_lists && !rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    return run_lists(list_offsets, n_lists, ids, n_ids, rows, lists_mem, out, out_mem, h->count, s, [&](const ListArgs &a) {
        return pairs_launch(h, nullptr, nullptr, 0, a.offsets, a.n_lists, a.rows, a.ids, a.n_pairs, a.out, s);
    });
}

qamd_status qamd_bin_topk(const qamd_bin *h, const qamd_bin_query *q, uint32_t k, int largest, uint32_t *out_ids,
                          float *out_scores, qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_query(h, q));
    if (k == 0) return QAMD_OK;
    if (!out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null output");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(q->ready.wait(s));
    {   // small stores: one launch, exact under any number of ties, no status read-back
        qamd_status st = QAMD_OK;
        if (bin_topk_small(h, q->buf.ptr, k, largest, out_ids, out_scores, out_mem, s, st)) return st;
    }
    if (!fused_capable(h)) {
        float *scores = nullptr;
        QAMD_TRY(thread_ws_acquire(WS_SCORES, std::max<uint64_t>(h->count, 1) * 4, s, reinterpret_cast<void **>(&scores)));
        qamd_status st = scan_into(h, q, scores, s);
        if (st == QAMD_OK) st = topk_finish(scores, h->count, k, largest, out_ids, out_scores, out_mem, s);
        thread_ws_release(WS_SCORES, s);
        return st;
    }
    FusedScan scan;
    scan.scan_scores = [&](float *scores, hipStream_t st) { return scan_into(h, q, scores, st); };
    scan.scan_filter = [&](const TopkFilter &f, hipStream_t st) { return scan_into(h, q, nullptr, st, &f); };
    scan.score_ids = [&](const uint32_t *ids, uint64_t n_ids, float *out, hipStream_t st) {
        return words_launch(h, q->buf.as<uint32_t>(), ids, n_ids, out, st);
    };
    return fused_topk(h->count, k, largest, out_ids, out_scores, out_mem, s, scan);
}

void qamd_bin_free(qamd_bin *h) { delete h; }

}  // extern "C"


// ============================================================================= many queries at once
// The caller's outer loop over queries (demos/src/ann_benchmark.rs:245-260).  score_batch reads every
// row ONCE for up to 8 queries (bin_scan_multi_kernel); topk_batch enqueues the per-query fused
// pipelines back to back (fused_topk_batch: one status read-back per 32 queries).
struct qamd_bin_query_batch {
    int device = 0;
    uint64_t nb = 0, ds = 0, n_queries = 0;
    uint64_t q_stride = 0;  // bytes between two queries' bit rows (multiple of 16)
    DevBuf bits;            // [n_queries][q_stride]
};

namespace {

template <int G, int ITERS, int UNROLL, int NQ>
void launch_bin_multi(const qamd_bin *h, const uint8_t *qbits, uint64_t q_stride, float *out, hipStream_t s,
                      const TopkFilterSlices *slices = nullptr) {
    constexpr int TILE = (64 / G) * UNROLL;
    const uint32_t rc = (uint32_t)(h->ds / 16);
    const uint64_t waves = (h->count + TILE - 1) / TILE;
    const unsigned grid = (unsigned)((waves + kScanBlock / 64 - 1) / (kScanBlock / 64));
    const bool exact = rc == (uint32_t)(G * ITERS);
#define QAMD_BIN_MULTI(EX, FI)                                                                                    \
    hipLaunchKernelGGL((bin_scan_multi_kernel<G, ITERS, UNROLL, NQ, EX, FI>), dim3(grid), dim3(kScanBlock), 0, s, \
                       h->rows.as<uint4>(), reinterpret_cast<const uint4 *>(qbits), (uint32_t)(q_stride / 16),    \
                       (float)h->vp.dim, (int)(h->vp.distance_type == QAMD_DOT), h->vp.invert, (uint32_t)h->count, \
                       rc, out, (uint64_t)h->count, slices ? *slices : TopkFilterSlices{})
    if (slices) {
        if (exact) QAMD_BIN_MULTI(true, true);
        else QAMD_BIN_MULTI(false, true);
    } else {
        if (exact) QAMD_BIN_MULTI(true, false);
        else QAMD_BIN_MULTI(false, false);
    }
#undef QAMD_BIN_MULTI
}

// Queries [q0, q0 + nq) with nq in {8, 4, 2}: rows of 8 .. 64 pieces (dims 1024 .. 8192 at the u128 granule).
template <int NQ> bool multi_step(const qamd_bin *h, const uint8_t *qbits, uint64_t q_stride, float *out, hipStream_t s,
                                  const TopkFilterSlices *slices = nullptr) {
    const uint32_t rc = (uint32_t)(h->ds / 16);
    if (rc < 8 || rc > 64 || h->ds % 16) return false;
    if (rc == 8) launch_bin_multi<8, 1, 8, NQ>(h, qbits, q_stride, out, s, slices);
    else if (rc <= 16) launch_bin_multi<16, 1, 4, NQ>(h, qbits, q_stride, out, s, slices);
    else if (rc <= 32) launch_bin_multi<16, 2, 2, NQ>(h, qbits, q_stride, out, s, slices);
    else if (NQ <= 4) launch_bin_multi<16, 4, 2, NQ>(h, qbits, q_stride, out, s, slices);
    else return false;
    return true;
}

}  // namespace

// ============================================================================= many queries on the matrix cores
// A binary store is a u8 store in disguise: bits as 0/1 bytes, a = popcount(q AND v) as the int8 contraction,
// xor = pop(q) + pop(v) - 2a, and the reference's metric (calculate_metric, :237-252)
//   zeros - xor = dim - 2 xor = (4a + (dim - 2 pop_q)) + (-2 pop_v)        (Dot, or L1/L2 inverted)
//   xor - zeros = 2 xor - dim = (-4a + (2 pop_q - dim)) + (2 pop_v)        (the other two cases)
// is the u8 epilogue (multiplier * s + q_offset) + v_offset with integer operands below 2^23: every f32 in it is
// exact, so the scores are the reference's bit for bit.  bin_gemm_rs_kernel is u8_gemm_rs_kernel's structure
// (csrc/u8_batch.hip: query tile resident in LDS, every wave streams its own 64 rows, no barrier after the
// set-up) with the rows kept as BITS up to the registers: lane (r, h) loads the 8 bytes [16 kb + 8h, +8) of row
// r per 128-bit K-block and expands 16 bits to 16 operand bytes (umul24(nibble, 0x00204081) & 0x01010101) right before
// the MFMAs that use them; the row's popcount falls out of the same loads.  HBM sees 1/8 of the bytes the
// contraction works on, so the kernel is matrix-pipe / VALU-bound from the first query tile on.
// The pre-filter starts the accumulators at -B_q and compares against the row's bound after the K loop (the
// row term is only known once the row has been read); the rest - pivots from a sample, wave-private candidate
// lists, scatter, emit - is batch_common.hpp's, as for u8.
namespace {

template <int MODE, bool LOW, int MI>  // MODE 0: scores out; 1 / 2: filter for the largest / smallest
__global__ __launch_bounds__(512) void bin_gemm_rs_kernel(const uint8_t *__restrict__ rows, uint32_t ds,
                                                         const uint8_t *__restrict__ qbits, uint32_t q_stride, float dim_f,
                                                         int zx, uint32_t n_rows, uint32_t n_queries, uint32_t q0,
                                                         float *__restrict__ out, uint64_t out_pitch, BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr int MJ = 2, TQ = 32 * MI, CHUNK = 64;
    constexpr bool FILTER = MODE != 0, LARGEST = MODE == 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t nkb = __builtin_amdgcn_readfirstlane(ds / 16);  // 128-bit K-blocks per row
    const uint32_t PA = nkb * 128 + 16;                            // LDS pitch of a query (one byte per bit)
    const float multiplier = zx ? 4.0f : -4.0f;
    float *q_off_s = reinterpret_cast<float *>(lds_raw + (size_t)TQ * PA);  // [64]
    float *pivot_s = q_off_s + 64;                                          // [64]
    int *bq_s = reinterpret_cast<int *>(pivot_s + 64);                      // [64]
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(bq_s + 64) + wave;
    if (FILTER && lane == 0) *wcount_s = 0;
    auto expand = [](uint32_t bits16) {  // 16 bits -> 16 bytes of 0 / 1
        // nibble * (1 + 2^7 + 2^14 + 2^21) puts bit j at position 8j; both factors fit the full-rate 24-bit
        // multiply (v_mul_u32_u24: 4 cycles; v_mul_lo_u32 is quarter rate and made this the kernel's bound)
        v4i v;
        v.x = (int)(__umul24(bits16 & 0xFu, 0x00204081u) & 0x01010101u);
        v.y = (int)(__umul24((bits16 >> 4) & 0xFu, 0x00204081u) & 0x01010101u);
        v.z = (int)(__umul24((bits16 >> 8) & 0xFu, 0x00204081u) & 0x01010101u);
        v.w = (int)(__umul24((bits16 >> 12) & 0xFu, 0x00204081u) & 0x01010101u);
        return v;
    };
    // the query tile as bytes (queries past the batch: zero), its offsets and integer bounds
    for (uint32_t idx = t; idx < (uint32_t)TQ * nkb * 8; idx += 512) {  // 16 bits each
        const uint32_t q = idx / (nkb * 8), piece = idx % (nkb * 8);
        uint32_t bits16 = 0;
        if (q0 + q < n_queries) bits16 = *reinterpret_cast<const uint16_t *>(qbits + (uint64_t)(q0 + q) * q_stride + piece * 2);
        *reinterpret_cast<v4i *>(lds_raw + q * PA + piece * 16) = expand(bits16);
    }
    if (t < TQ) {
        uint32_t pq = 0;
        if (q0 + t < n_queries)
            for (uint32_t w = 0; w < nkb * 4; w++) pq += __popc(*reinterpret_cast<const uint32_t *>(qbits + (uint64_t)(q0 + t) * q_stride + w * 4));
        const float qo = zx ? dim_f - 2.0f * (float)pq : 2.0f * (float)pq - dim_f;
        q_off_s[t] = qo;
        if (FILTER) {
            const float pv = q0 + t < n_queries ? filt.pivot_scores[q0 + t] : (LARGEST ? __builtin_huge_valf() : -__builtin_huge_valf());
            pivot_s[t] = pv;
            int bq = pp_bound<LOW>(pv - qo, fabsf(pv) + fabsf(qo), multiplier, 1);
            if (__builtin_isinf(pv)) bq = ((pv > 0.0f) == LARGEST) == LOW ? -(int)kPpLim : (int)kPpLim;
            bq_s[t] = bq;
        }
    }
    __syncthreads();

    const uint32_t n_chunks = (n_rows + CHUNK - 1) / CHUNK, stride = gridDim.x * 8;
    const uint8_t *a_base = lds_raw + r * PA + 64 * h;
    uint2 cur[MJ][8];  // up to 8 K-blocks in registers at a time (rows of up to 1024 bits per pass)
    for (uint32_t chunk = blockIdx.x * 8 + wave; chunk < n_chunks; chunk += stride) {
        const uint64_t row0 = (uint64_t)chunk * CHUNK;
        v16i acc[MI][MJ];
        uint32_t pop[MJ] = {0, 0};
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                v4i b4 = {0, 0, 0, 0};
                if (FILTER) b4 = *reinterpret_cast<const v4i *>(bq_s + i * 32 + 8 * gq + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int jj = 0; jj < MJ; jj++) acc[i][jj][4 * gq + e] = -b4[e];
            }
        for (uint32_t kb0 = 0; kb0 < nkb; kb0 += 8) {  // the store's rows are padded: reads past a row stay in bounds
            const uint32_t nk = nkb - kb0 < 8 ? nkb - kb0 : 8;
#pragma unroll
            for (int jj = 0; jj < MJ; jj++) {
                const uint8_t *p = rows + (row0 + jj * 32 + r) * ds + (size_t)kb0 * 16 + 8 * h;
#pragma unroll
                for (int kb = 0; kb < 8; kb++)
                    cur[jj][kb] = (uint32_t)kb < nk ? *reinterpret_cast<const uint2 *>(p + 16 * kb) : make_uint2(0, 0);
            }
#pragma unroll
            for (int kb = 0; kb < 8; kb++) {
                if ((uint32_t)kb >= nk) break;
#pragma unroll
                for (int jj = 0; jj < MJ; jj++) pop[jj] += __popc(cur[jj][kb].x) + __popc(cur[jj][kb].y);
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    v4i bf[MJ];
#pragma unroll
                    for (int jj = 0; jj < MJ; jj++) {
                        const uint32_t word = (x < 2) ? cur[jj][kb].x : cur[jj][kb].y;
                        bf[jj] = expand((word >> (16 * (x & 1))) & 0xFFFFu);
                    }
#pragma unroll
                    for (int i = 0; i < MI; i++) {
                        const v4i a = *reinterpret_cast<const v4i *>(a_base + (uint32_t)i * 32u * PA + (kb0 + kb) * 128 + 16 * x);
#pragma unroll
                        for (int jj = 0; jj < MJ; jj++) acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf[jj], acc[i][jj], 0, 0, 0);
                    }
                }
            }
        }
        // ---- epilogue: rows on the lanes (r; the two halves h hold the two halves of every K-block's bits)
        uint4 *wave_list = FILTER ? filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * 8 + wave) * filt.wave_cap : nullptr;
#pragma unroll
        for (int jj = 0; jj < MJ; jj++) {
            const uint32_t pv = pop[jj] + (uint32_t)__shfl_xor((int)pop[jj], 32);
            const uint64_t row = row0 + jj * 32 + r;
            const bool row_ok = row < n_rows;
            const float v_off = zx ? -2.0f * (float)pv : 2.0f * (float)pv;
            int br = 0;
            if (FILTER) br = row_ok ? pp_bound<LOW>(-v_off, fabsf(v_off), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
#pragma unroll
            for (int i = 0; i < MI; i++) {
                if (FILTER) {  // "some accumulator of the tile may pass": the largest is >= the row bound (LOW: the smallest <)
                    int ext = acc[i][jj][0];
#pragma unroll
                    for (int e = 1; e < 15; e += 2)
                        ext = LOW ? min(min(ext, acc[i][jj][e]), acc[i][jj][e + 1]) : max(max(ext, acc[i][jj][e]), acc[i][jj][e + 1]);
                    ext = LOW ? min(ext, acc[i][jj][15]) : max(ext, acc[i][jj][15]);
                    if (!__builtin_amdgcn_readfirstlane(__ballot(LOW ? ext < br : ext >= br) != 0)) continue;
                }
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const uint32_t ql = i * 32 + 8 * gq + 4 * h;
                    const float4 qo4 = *reinterpret_cast<const float4 *>(q_off_s + ql);
                    const float qo[4] = {qo4.x, qo4.y, qo4.z, qo4.w};
                    if (!FILTER) {
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float sc = (multiplier * (float)acc[i][jj][4 * gq + e] + qo[e]) + v_off;
                            const uint32_t q = q0 + ql + e;
                            if (row_ok && q < n_queries) out[(uint64_t)q * out_pitch + row] = sc;
                        }
                    } else {
                        const v4i bq4 = *reinterpret_cast<const v4i *>(bq_s + ql);
                        const float4 pv4 = *reinterpret_cast<const float4 *>(pivot_s + ql);
                        const float pvt[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int av = acc[i][jj][4 * gq + e];
                            if (LOW ? av < br : av >= br) {  // may pass: the exact f32 comparison decides
                                const float sc = (multiplier * (float)(av + bq4[e]) + qo[e]) + v_off;
                                const float d = LARGEST ? sc - pvt[e] : pvt[e] - sc;
                                if (d >= 0.0f) {
                                    const uint32_t pos = atomicAdd(wcount_s, 1u);
                                    if (pos < filt.wave_cap)
                                        wave_list[pos] = make_uint4(topk_ordered_bits(sc, LARGEST), (uint32_t)row, q0 + ql + e, 0u);
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    if (FILTER && lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = *wcount_s;
}

// ------------------------------------------------------------------------------------------
// Many queries on the FP4 matrix cores (round 3).  popcount(q AND v) is a dot product of 0/1 values, and 0 and 1 are
// exact in every MFMA input format: as E2M1 nibbles (0b0010 = 1.0) through v_mfma_scale_f32_16x16x128_f8f6f4 (both
// scales 2^0) a k-step covers 128 BITS per instruction and the part runs it at 8.85 POP/s and 2.35 GHz in a bare loop
// (the int8 instructions: 3.6-4.4 POP/s at 1.85-2.2 GHz); the f32 accumulators hold the exact counts (< 2^24).  What made
// bin_gemm_rs_kernel vector-ALU-bound was expanding every row's bits once per 64-query tile in every wave; here the
// structure is u8_gemm_qs16_kernel's (csrc/u8_batch.hip): a workgroup expands a block of 128 rows ONCE into LDS
// (nibbles: 4 x the row bytes, pitch = whole 256-byte bank rows, 16-byte chunks XOR (row & 15)) and streams the whole
// batch past it - up to 2048 queries per launch, their nibble image prepared once per call in fragment order
// (bin_frag4_kernel: per 16 queries and 128-bit k-step one 1 KiB piece, lane (i, g) = bits [128 s + 32 g, +32) of query
// i).  Same epilogue arithmetic as bin_gemm_rs_kernel: the scores are the reference's bit for bit.  Rows of 512 / 768 / 1024 /
// 1536 bits.
constexpr uint64_t kQs4Slice = 2048, kQs4MinQueries = 129;  // queries per launch of bin_gemm_qs4_kernel; from where it is used
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

// 32 bits -> 32 E2M1 nibbles (1.0 = 0b0010 where set).  Rows and queries go through the same function and a k-step sums
// over all 32 places of a dword's chunk, so ANY fixed placement of the bits inside the chunk does: dword p of the result
// holds bit pairs p of the four bytes (bits 2 p, 2 p + 1 of byte j at nibbles 2 j, 2 j + 1).  Per output dword one shift,
// one mask - four byte-sized selectors 0..3 - and one v_perm_b32 that looks each selector up in the four-byte table
// {0x00, 0x02, 0x20, 0x22}: 11 vector-ALU operations per 32 bits.  (Round 3 spread every nibble of every byte with 24-bit
// multiplies: ~38 operations, 1.1 us of the 3.3 us a 128-row block cost.)
__device__ __forceinline__ uint4 nib32(uint32_t w) {
    constexpr uint32_t kPairs = 0x03030303u, kTable = 0x22200200u;
    return make_uint4(__builtin_amdgcn_perm(0u, kTable, w & kPairs), __builtin_amdgcn_perm(0u, kTable, (w >> 2) & kPairs),
                      __builtin_amdgcn_perm(0u, kTable, (w >> 4) & kPairs), __builtin_amdgcn_perm(0u, kTable, (w >> 6) & kPairs));
}

// out[(t16 * nsteps + s) * 64 + lane] = nibbles of bits [128 s + 32 g, +32) of query 16 t16 + i (lane = 16 g + i); also the
// query's offset and, given its pivot, the integer bound of the pre-filter (as bin_gemm_rs_kernel computes them per tile)
template <bool LOW>
__global__ __launch_bounds__(256) void bin_frag4_kernel(const uint8_t *__restrict__ qbits, uint32_t q_stride, uint32_t n_queries,
                                                       uint32_t q_pad, uint32_t nsteps, float dim_f, int zx, int largest,
                                                       const float *__restrict__ pivots, uint4 *__restrict__ out,
                                                       float *__restrict__ q_off, int *__restrict__ bq) {
    const uint64_t total = (uint64_t)(q_pad / 16) * nsteps * 64;
    for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
        const uint32_t lane = (uint32_t)(idx & 63u);
        const uint64_t ts = idx >> 6;
        const uint32_t sx = (uint32_t)(ts % nsteps), t16 = (uint32_t)(ts / nsteps);
        const uint32_t q = 16u * t16 + (lane & 15u), g = lane >> 4;
        uint32_t w = 0;
        if (q < n_queries) w = *reinterpret_cast<const uint32_t *>(qbits + (uint64_t)q * q_stride + 16u * sx + 4u * g);
        out[idx] = nib32(w);
    }
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q < q_pad) {
        uint32_t pq = 0;
        if (q < n_queries)
            for (uint32_t w = 0; w < nsteps * 4; w++) pq += __popc(*reinterpret_cast<const uint32_t *>(qbits + (uint64_t)q * q_stride + w * 4));
        const float qo = zx ? dim_f - 2.0f * (float)pq : 2.0f * (float)pq - dim_f;
        const float multiplier = zx ? 4.0f : -4.0f;
        const bool lg = largest != 0;
        const float pv = q < n_queries ? pivots[q] : (lg ? __builtin_huge_valf() : -__builtin_huge_valf());
        int b = pp_bound<LOW>(pv - qo, fabsf(pv) + fabsf(qo), multiplier, 1);
        if (__builtin_isinf(pv)) b = ((pv > 0.0f) == lg) == LOW ? -(int)kPpLim : (int)kPpLim;
        q_off[q] = qo;
        bq[q] = b;
    }
}

// IT: 16-query tiles per wave and chunk - 4 (64 queries), 2 for batches of up to 256 queries, 1 up to 128 (a chunk for every wave)
// JT: 16-row tiles per block - 8 (128 rows; rows of up to 1024 bits), 6 (96 rows of 1536 bits: two 72 KiB slabs)
template <int MODE, bool LOW, int IT, int JT>  // MODE 1 / 2: filter for the largest / smallest scores
__global__ __launch_bounds__(512) void bin_gemm_qs4_kernel(const uint8_t *__restrict__ rows, uint32_t ds,
                                                          const uint4 *__restrict__ qfrag, const float *__restrict__ q_offsets,
                                                          const int *__restrict__ bq_all, int zx, uint32_t n_rows,
                                                          uint32_t n_queries, BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr int JH = JT / 2, QS_ROWS = 16 * JT, CQ = 16 * IT, MAXSTEPS = JT == 8 ? 8 : 12;
    constexpr bool LARGEST = MODE == 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t i16 = (uint32_t)lane & 15u, g4 = (uint32_t)lane >> 4;
    const uint32_t nsteps = __builtin_amdgcn_readfirstlane(ds / 16);  // k-steps of 128 bits
    const uint32_t PA = ((ds * 4 + 255) / 256) * 256;                 // LDS pitch of a row's nibble image: whole 256-byte bank rows
                                                                      // (768-bit rows: 384 bytes of nibbles on a 512-byte pitch; the
                                                                      // places past the image are only multiplied with zero nibbles)
    const float multiplier = zx ? 4.0f : -4.0f;
    const uint32_t n_blocks = (n_rows + QS_ROWS - 1) / QS_ROWS;
    const uint32_t live_chunks = (n_queries + CQ - 1) / CQ;
    // TWO row blocks in LDS (a block of 128 rows is only 16 KiB of bits: the cost of changing blocks is latency, and with a
    // single slab it was 3.9 us per block whatever the batch): the next block's bits are requested into 8 registers when
    // this block starts, expanded into the other slab when this block's chunks are done - one barrier per block
    const uint32_t SLAB = QS_ROWS * PA;
    float *voff_s = reinterpret_cast<float *>(lds_raw + 2 * (size_t)SLAB);  // [2][128]
    int *br_s = reinterpret_cast<int *>(voff_s + 2 * QS_ROWS);              // [2][128]
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(br_s + 2 * QS_ROWS) + wave;
    int *bq_s = reinterpret_cast<int *>(br_s + 2 * QS_ROWS) + 16;           // [CQ * live_chunks]
    // (expanding through a 256-entry byte -> nibbles table in LDS instead of the multiplies was measured: no faster)
    if (lane == 0) *wcount_s = 0;
    // the "always" / "never" sentinels of the bounds (+-2^29) are brought to +-2^23: still far beyond any count, and the
    // accumulators (count - bound, f32) stay exact integers
    for (uint32_t i = t; i < CQ * live_chunks; i += 512) bq_s[i] = max(min(bq_all[i], 1 << 23), -(1 << 23));

    v4i Q0[IT], Q1[IT], Q2[IT];
    auto load_step = [&](v4i(&a)[IT], uint32_t c, uint32_t j) {
        const uint4 *p = qfrag + ((uint64_t)(IT * c) * nsteps + j) * 64 + lane;
#pragma unroll
        for (int it = 0; it < IT; it++) {
            const uint4 v = p[(uint64_t)it * nsteps * 64];
            a[it] = v4i{(int)v.x, (int)v.y, (int)v.z, (int)v.w};
        }
    };
    // Small batches (IT < 4: at most one chunk per wave): the wave's query fragments for all (at most 8) k-steps stay in
    // registers for the whole launch.  Streamed, a chunk of 16 or 32 queries is 8 short k-steps behind an L2 round trip
    // each - a 4 us floor per row block whatever the batch.
    constexpr bool QREG = IT < 4;
    v4i Qr[QREG ? MAXSTEPS : 1][IT];
    if (QREG) {
#pragma unroll
        for (int j = 0; j < MAXSTEPS; j++)
#pragma unroll
            for (int it = 0; it < IT; it++) Qr[j][it] = v4i{0, 0, 0, 0};
        if ((uint32_t)wave < live_chunks) {
#pragma unroll
            for (int j = 0; j < MAXSTEPS; j++)
                if ((uint32_t)j < nsteps) load_step(Qr[j], (uint32_t)wave, (uint32_t)j);
        }
    }
    const uint32_t b_row = i16 * PA, b_gi = (g4 ^ i16) * 16u;
    // Row-block fill: the block is 128 * ds contiguous bytes of the store; thread t takes the 16-byte (128-bit) pieces
    // t, t + 512, ... (ds / 16 * 128 / 512 = ds / 64 of them: 2 at 1024 bits), requested before the barrier that frees the
    // LDS rows, expanded and written after it: piece pc of row r = nibble chunks 4 pc .. 4 pc + 3, each at place
    // chunk ^ (r & 15).  The row's popcount (its offset in the score) is the sum over its ds / 16 pieces, which sit in
    // adjacent lanes.
    const uint32_t per = ds / 16;                        // 128-bit pieces per row: 4, 6, 8 or 12
    const uint32_t sper = per <= 4 ? 4u : per <= 8 ? 8u : 16u;  // lanes per row (a power of two: the row's popcount is a butterfly sum)
    const uint32_t n_pieces = __builtin_amdgcn_readfirstlane(QS_ROWS * sper / 512);  // rounds: rows * sper lanes / 512 threads
    constexpr int MAXP = JT == 8 ? 2 : 3;  // 128 rows x 8 lanes / 512; 96 x 16 / 512
    // TWO register sets: a block's bits are requested one block before they are expanded.  `vmcnt` retires in order and
    // the K loop waits for its query fragments at every k-step, so a request in front of a K loop stalls it for an HBM
    // round trip (measured: a 3.9 us floor per block); the request is placed behind the last query loads of the wave's
    // last chunk instead, and the data is not needed before the END of the following block.
    // (Round 4: up to FOUR register sets, the request four blocks ahead.  With two, a workgroup had at most two 16 KiB blocks in
    // flight - 32 KiB per CU against the ~64 KiB that 8 TB/s x ~2 us of HBM latency ask for: a launch with the K loop cut
    // out still took 3.2 ms for the 6.4 GB of a 50M x 1024-bit store, 2 TB/s, and that was the floor of every small batch.)
    // The streamed-query forms (IT = 4) and the 96-row form with IT = 2 sit at the 256-register limit and keep two sets:
    // their blocks last long enough (MFMA-bound) for one block of look-ahead.
    constexpr int DEPTH = (IT == 4 || (IT == 2 && JT == 6)) ? 2 : 4;
    uint4 st[DEPTH][MAXP];
    auto fill_request = [&](uint4(&st)[MAXP], uint32_t blk) {
        const uint8_t *p = rows + (uint64_t)blk * QS_ROWS * ds;
#pragma unroll
        for (int i = 0; i < MAXP; i++) {
            const uint32_t ii = (uint32_t)i < n_pieces ? (uint32_t)i : n_pieces - 1;  // wave-uniform
            const uint32_t slot = (uint32_t)t + 512u * ii, row = slot / sper, pc = slot % sper;
            st[i] = pc < per ? ld_nt(reinterpret_cast<const uint4 *>(p + (size_t)row * ds + (size_t)pc * 16)) : make_uint4(0, 0, 0, 0);
        }
    };
    auto fill_write = [&](const uint4(&st)[MAXP], uint64_t row0, uint32_t par) {
        uint8_t *slab = lds_raw + par * SLAB;
#pragma unroll
        for (int i = 0; i < MAXP; i++) {
            if ((uint32_t)i >= n_pieces) break;
            const uint32_t slot = (uint32_t)t + 512u * i, row = slot / sper, pc = slot % sper;
            const uint32_t w[4] = {st[i].x, st[i].y, st[i].z, st[i].w};
            if (pc < per) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t chunk = 4u * pc + k;
                    *reinterpret_cast<uint4 *>(slab + row * PA + ((chunk ^ (row & 15u)) * 16u)) = nib32(w[k]);
                }
            }
            // popcount of the row: pieces of a row are in `per` adjacent lanes of this round (per divides 64)
            uint32_t pop = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
            for (uint32_t d = 1; d < sper; d <<= 1) pop += (uint32_t)__shfl_xor((int)pop, (int)d);  // (lanes past the row's pieces hold zeros)
            if (pc == 0) {
                const bool ok = row0 + row < n_rows;
                const float v_off = zx ? -2.0f * (float)pop : 2.0f * (float)pop;
                voff_s[par * QS_ROWS + row] = v_off;
                br_s[par * QS_ROWS + row] = ok ? pp_bound<LOW>(-v_off, fabsf(v_off), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
            }
        }
    };
    const uint32_t my_first = wave;
    const uint32_t first_blk = blockIdx.x < n_blocks ? blockIdx.x : 0u;
    auto clamp_blk = [&](uint32_t blk) { return blk < n_blocks ? blk : first_blk; };  // past the end: a harmless re-read
    // set (i mod DEPTH) holds the bits of the workgroup's i-th block from its request until they are expanded
#pragma unroll
    for (int d = 0; d < DEPTH; d++) fill_request(st[d], clamp_blk(first_blk + (uint32_t)d * gridDim.x));
    fill_write(st[0], (uint64_t)first_blk * QS_ROWS, 0u);
    if (!QREG && my_first < live_chunks) {
        load_step(Q0, my_first, 0);
        load_step(Q1, my_first, 1);
    }
    __syncthreads();
    uint4 *wave_list = filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * 8 + wave) * filt.wave_cap;

    // one row block: slab `par` holds it, st_use the next block's bits (requested three blocks ago), st_req takes the bits of
    // the block after that
    auto block_body = [&](uint32_t blk, uint32_t par, const uint4(&st_use)[MAXP], uint4(&st_req)[MAXP]) {
        const uint64_t row0 = (uint64_t)blk * QS_ROWS;
        const uint32_t next_blk = clamp_blk(blk + gridDim.x), next2_blk = clamp_blk(blk + (uint32_t)DEPTH * gridDim.x);
        bool requested = false;
        const uint8_t *slab = lds_raw + par * SLAB;
        const float *voff_cur = voff_s + par * QS_ROWS;
        const int *br_cur = br_s + par * QS_ROWS;
        for (uint32_t c = wave; c < live_chunks; c += 8) {
            const uint32_t c_next = c + 8 < live_chunks ? c + 8 : my_first;
            v4f_t acc[IT][JT];
#pragma unroll
            for (int it = 0; it < IT; it++) {
                const v4i bq4 = *reinterpret_cast<const v4i *>(bq_s + CQ * c + 16 * it + 4 * g4);
#pragma unroll
                for (int jt = 0; jt < JT; jt++)
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[it][jt][e] = -(float)bq4[e];  // |bound| <= dim or the 2^29 sentinel: exact
            }
            auto compute = [&](const v4i(&a)[IT], uint32_t j) {
                uint32_t lane_addr = b_row + (b_gi ^ ((j & 3u) * 64u)) + (j >> 2) * 256u;
                asm volatile("" : "+v"(lane_addr));
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    v4i bf[JH];
#pragma unroll
                    for (int j4 = 0; j4 < JH; j4++)
                        bf[j4] = *reinterpret_cast<const v4i *>(slab + lane_addr + (uint32_t)(JH * hf + j4) * 16u * PA);
#pragma unroll
                    for (int it = 0; it < IT; it++) {
                        const v8i_t a8 = {a[it].x, a[it].y, a[it].z, a[it].w, 0, 0, 0, 0};
#pragma unroll
                        for (int j4 = 0; j4 < JH; j4++) {
                            const v8i_t b8 = {bf[j4].x, bf[j4].y, bf[j4].z, bf[j4].w, 0, 0, 0, 0};
                            acc[it][JH * hf + j4] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[it][JH * hf + j4], 4, 4, 0, 127, 0, 127);
                        }
                    }
                }
            };
            auto request = [&](v4i(&a)[IT], uint32_t j) {
                if (j < nsteps) load_step(a, c, j);
                else load_step(a, c_next, j - nsteps);
            };
            if (((c >> 3) + ((uint32_t)wave >> 2)) & 1u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(0);
            uint32_t j = 0;
            if (QREG) {
#pragma unroll
                for (int jj = 0; jj < MAXSTEPS; jj++)
                    if ((uint32_t)jj < nsteps) compute(Qr[jj], (uint32_t)jj);
                j = nsteps;
            }
            for (; !QREG && j + 2 < nsteps; j += 3) {
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
            const uint32_t left = QREG ? 0u : nsteps - j;
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
            if (c + 8 >= live_chunks) {  // this wave's last chunk of the block: no wait for younger loads follows before the block ends
                fill_request(st_req, next2_blk);
                requested = true;
            }
            // ---- epilogue: lane (i16, g4) holds, per (it, jt), queries CQ c + 16 it + 4 g4 + e against row 16 jt + i16
            uint32_t c_e = c, lane_e = (uint32_t)lane;
            asm volatile("" : "+s"(c_e), "+v"(lane_e));
            const uint32_t i_e = lane_e & 15u, g_e = lane_e >> 4;
            // "some accumulator may pass": acc = count - query bound; the row bound decides (as in bin_gemm_rs_kernel)
            bool any_lane = false;
#pragma unroll
            for (int jt = 0; jt < JT; jt++) {
                const float brf = (float)br_cur[jt * 16 + i_e];
                float ext = acc[0][jt][0];
#pragma unroll
                for (int it = 0; it < IT; it++)
#pragma unroll
                    for (int e = 0; e < 4; e++) ext = LOW ? fminf(ext, acc[it][jt][e]) : fmaxf(ext, acc[it][jt][e]);
                any_lane |= LOW ? ext < brf : ext >= brf;
            }
            if (__builtin_amdgcn_readfirstlane(__ballot(any_lane) != 0)) {
#pragma unroll
                for (int jt = 0; jt < JT; jt++) {
                    __builtin_amdgcn_sched_barrier(0);
                    const uint64_t row = row0 + jt * 16 + i_e;
                    const int br = br_cur[jt * 16 + i_e];
                    const float v_off = voff_cur[jt * 16 + i_e];
#pragma unroll
                    for (int it = 0; it < IT; it++) {
                        const uint32_t q = CQ * c_e + 16 * it + 4 * g_e;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int av = (int)acc[it][jt][e];
                            if (LOW ? av < br : av >= br) {  // may pass: the exact f32 comparison decides
                                const float qo = q_offsets[q + e];
                                const float pvt = filt.pivot_scores[q + e];
                                const float sc = (multiplier * (float)(av + bq_s[q + e]) + qo) + v_off;
                                const float d = LARGEST ? sc - pvt : pvt - sc;
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
        }
        if (!requested) fill_request(st_req, next2_blk);  // a wave without chunks
        fill_write(st_use, (uint64_t)next_blk * QS_ROWS, par ^ 1u);  // the other slab: last read a block ago, a barrier has passed since
        __syncthreads();
    };
    for (uint32_t blk = blockIdx.x; blk < n_blocks; blk += 4 * gridDim.x) {  // block i: expands set (i + 1) mod DEPTH, refills set i mod DEPTH
        block_body(blk, 0u, st[1 % DEPTH], st[0]);
        if (blk + gridDim.x >= n_blocks) break;
        block_body(blk + gridDim.x, 1u, st[2 % DEPTH], st[1 % DEPTH]);
        if (blk + 2 * gridDim.x >= n_blocks) break;
        block_body(blk + 2 * gridDim.x, 0u, st[3 % DEPTH], st[2 % DEPTH]);
        if (blk + 3 * gridDim.x >= n_blocks) break;
        block_body(blk + 3 * gridDim.x, 1u, st[0], st[3 % DEPTH]);
    }
    if (lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * 8 + wave] = *wcount_s;
}

// ------------------------------------------------------------------------------------------
// FP4 matrix cores, ROW-STREAMING form (round 4): batches whose whole nibble image fits in LDS - 256 queries of 1024 bits
// are 128 KiB.  bin_gemm_qs4_kernel keeps ROWS in LDS and pays per 128-row block an expansion phase, an epilogue phase and
// a barrier during which the matrix pipes idle (timeline, profiles/r04_bin_batch.txt: at 256 queries the K loop is 35 % of
// a block, at 128 queries less; no amount of look-ahead on the row loads moved it).  Here the roles are swapped, as in
// bin_gemm_rs_kernel / u8_gemm_rs_kernel: the QUERIES are resident (fragment order: the A operand of k-step s of a
// 16-query tile is one lane-linear 1 KiB piece, conflict-free by construction), every wave streams its OWN rows - 32 per
// trip, loaded two trips ahead, expanded to nibbles ONCE per row into 64 registers, then multiplied with every query tile -
// and after the set-up there is no barrier: each SIMD's two waves drift apart and fill each other's vector-ALU phases with
// MFMAs.  A lane owns NS consecutive dwords of its row (lane (i, g): dwords [NS g, NS g + NS) of row i: whole 16-byte
// loads, every line fetched once); k-step s multiplies dword NS g + s of the row with the same dword of the query, which is
// where the staging copy puts it (any pairing of K positions is the same dot product).  Arithmetic, bounds and candidate
// lists are bin_gemm_qs4_kernel's.
constexpr uint32_t kRs4MinQueries = 12;  // (the int8 matrix-core form; the fp4 form takes 3 and 5+ queries)
constexpr uint32_t kRs4MaxPasses = 4;  // (measured: profiles/r04_bin_batch.txt; four passes tie with the query-streaming form)
inline uint32_t rs4_max_queries(uint64_t ds) {  // whole 32-query tile pairs whose nibble image + bounds fit the CU's LDS
    return (uint32_t)((160 * 1024 - 1024) / (ds * 4 + 4) / 32 * 32);
}
// Waves per workgroup: 12 (three per SIMD: 166 registers at NS = 8) where the registers allow, else 8 - a third wave covers
// more of the other two's vector-ALU phases with MFMAs.
constexpr int rs4_waves(int ns) { return ns <= 8 ? 12 : 8; }
template <int MODE, bool LOW, int NS>  // MODE 1 / 2: filter for the largest / smallest scores; NS = k-steps = ds / 16
__global__ __launch_bounds__(64 * rs4_waves(NS)) void bin_gemm_rs4_kernel(const uint8_t *__restrict__ rows, const uint4 *__restrict__ qfrag,
                                                          const float *__restrict__ q_offsets, const int *__restrict__ bq_all,
                                                          int zx, uint32_t n_rows, uint32_t n_tiles /* even */, BatchFilter filt) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds_raw[];
    constexpr int RT = 2, QP = 2, DS = 16 * NS, WAVES = rs4_waves(NS), THREADS = 64 * WAVES;
    constexpr bool LARGEST = MODE == 1;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t i16 = (uint32_t)lane & 15u, g4 = (uint32_t)lane >> 4;
    const float multiplier = zx ? 4.0f : -4.0f;
    uint4 *img = reinterpret_cast<uint4 *>(lds_raw);                                      // [n_tiles][NS][64] x 16 B
    float *nbq_s = reinterpret_cast<float *>(lds_raw + (size_t)n_tiles * NS * 1024);      // [16 n_tiles]: MINUS the query bounds
    uint32_t *wcount_s = reinterpret_cast<uint32_t *>(nbq_s + 16 * n_tiles) + wave;
    if (lane == 0) *wcount_s = 0;
    for (uint32_t idx = t; idx < n_tiles * NS * 64; idx += THREADS) {  // position (tile, s, lane (i, g)) <- the query's dword NS g + s
        const uint32_t tile = idx / (NS * 64), rem = idx % (NS * 64), sx = rem >> 6, i = rem & 15u, g = (rem >> 4) & 3u;
        const uint32_t d = NS * g + sx;
        img[idx] = qfrag[((uint64_t)tile * NS + d / 4) * 64 + 16 * (d % 4) + i];
    }
    // (the "always" / "never" sentinels of the bounds, +-2^29, are brought to +-2^23: far beyond any count, and exact in f32)
    // stored negated and as f32: a tile's accumulators START at -bound, and the first MFMA of a chain takes these four
    // values as its C operand straight from the LDS read - no conversion, no copy per tile
    for (uint32_t i = t; i < 16 * n_tiles; i += THREADS) nbq_s[i] = -(float)max(min(bq_all[i], 1 << 23), -(1 << 23));
    __syncthreads();

    const uint32_t n_chunks = (n_rows + 16 * RT - 1) / (16 * RT), stride = gridDim.x * WAVES;
    uint32_t raw[RT][NS], raw2[RT][NS];  // the rows of this trip and of the next; the trip after that is requested into `raw`
                                         // as soon as it has been expanded (two trips = 8 KiB per wave in flight)
    auto load_rows = [&](uint32_t (&raw)[RT][NS], uint32_t chunk) {  // rows past the store are its zero padding
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const uint8_t *p = rows + ((uint64_t)chunk * (16 * RT) + rt * 16 + i16) * DS + 4 * NS * g4;
            if (NS % 4 == 0) {
#pragma unroll
                for (int v = 0; v < NS / 4; v++) {
                    const uint4 x = reinterpret_cast<const uint4 *>(p)[v];
                    raw[rt][4 * v] = x.x, raw[rt][4 * v + 1] = x.y, raw[rt][4 * v + 2] = x.z, raw[rt][4 * v + 3] = x.w;
                }
            } else {
#pragma unroll
                for (int v = 0; v < NS / 2; v++) {
                    const uint2 x = reinterpret_cast<const uint2 *>(p)[v];
                    raw[rt][2 * v] = x.x, raw[rt][2 * v + 1] = x.y;
                }
            }
        }
    };
    uint4 *wave_list = filt.wave_cand + (uint64_t)(filt.wave_base + blockIdx.x * WAVES + wave) * filt.wave_cap;
    const uint32_t first = blockIdx.x * WAVES + wave;
    auto clamp_chunk = [&](uint32_t c) { return c < n_chunks ? c : (first < n_chunks ? first : 0u); };  // past the end: a harmless re-read
    load_rows(raw, clamp_chunk(first));
    load_rows(raw2, clamp_chunk(first + stride));
    auto trip = [&](uint32_t chunk, uint32_t (&raw)[RT][NS]) {
        v8i_t B[RT][NS];
        uint32_t pop[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            pop[rt] = 0;
#pragma unroll
            for (int sx = 0; sx < NS; sx++) {
                const uint4 nb = nib32(raw[rt][sx]);
                B[rt][sx] = v8i_t{(int)nb.x, (int)nb.y, (int)nb.z, (int)nb.w, 0, 0, 0, 0};
                pop[rt] += __popc(raw[rt][sx]);
            }
        }
        const uint64_t row0 = (uint64_t)chunk * (16 * RT);
        load_rows(raw, clamp_chunk(chunk + 2 * stride));  // two trips ahead, behind this trip's MFMAs
        float v_off[RT], brf[RT];
        int br[RT];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {  // the row's popcount: its four lanes (i, 0..3)
            uint32_t pv = pop[rt] + (uint32_t)__shfl_xor((int)pop[rt], 16);
            pv += (uint32_t)__shfl_xor((int)pv, 32);
            v_off[rt] = zx ? -2.0f * (float)pv : 2.0f * (float)pv;
            const bool ok = row0 + rt * 16 + i16 < n_rows;
            br[rt] = ok ? pp_bound<LOW>(-v_off[rt], fabsf(v_off[rt]), multiplier, 0) : (LOW ? -(int)kPpLim : (int)kPpLim);
            brf[rt] = (float)br[rt];
        }
        for (uint32_t qt = 0; qt < n_tiles; qt += QP) {
            v4f_t acc[QP][RT];
            const uint4 *a_base = img + (size_t)qt * NS * 64 + lane;
#pragma unroll
            for (int sx = 0; sx < NS; sx++) {
#pragma unroll
                for (int qp = 0; qp < QP; qp++) {
                    const uint4 a = a_base[(qp * NS + sx) * 64];
                    const v8i_t a8 = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
                    const v4f_t start = *reinterpret_cast<const v4f_t *>(nbq_s + 16 * (qt + qp) + 4 * g4);  // (used at sx == 0 only)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++)
                        acc[qp][rt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, B[rt][sx], sx == 0 ? start : acc[qp][rt], 4, 4, 0, 127, 0, 127);
                }
            }
            // ---- epilogue: lane (i, g) holds queries 16 (qt + qp) + 4 g + e against row 16 rt + i of the trip
            bool any_lane = false;
#pragma unroll
            for (int qp = 0; qp < QP; qp++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const float ext = LOW ? fminf(fminf(acc[qp][rt][0], acc[qp][rt][1]), fminf(acc[qp][rt][2], acc[qp][rt][3]))
                                          : fmaxf(fmaxf(acc[qp][rt][0], acc[qp][rt][1]), fmaxf(acc[qp][rt][2], acc[qp][rt][3]));
                    any_lane |= LOW ? ext < brf[rt] : ext >= brf[rt];
                }
            if (__builtin_amdgcn_readfirstlane(__ballot(any_lane) != 0)) {
#pragma unroll
                for (int qp = 0; qp < QP; qp++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const uint32_t q = 16 * (qt + qp) + 4 * g4;
                        const uint64_t row = row0 + rt * 16 + i16;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const int av = (int)acc[qp][rt][e];
                            if (LOW ? av < br[rt] : av >= br[rt]) {  // may pass: the exact f32 comparison decides
                                const float qo = q_offsets[q + e];
                                const float pvt = filt.pivot_scores[q + e];
                                const float sc = (multiplier * (float)(av - (int)nbq_s[q + e]) + qo) + v_off[rt];
                                const float d = LARGEST ? sc - pvt : pvt - sc;
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
    };
    for (uint32_t chunk = first; chunk < n_chunks; chunk += 2 * stride) {
        trip(chunk, raw);
        if (chunk + stride >= n_chunks) break;
        trip(chunk + stride, raw2);
    }
    if (lane == 0) filt.wave_counts[filt.wave_base + blockIdx.x * WAVES + wave] = *wcount_s;
}

// Sample rows for the pivots (rows only; the golden-ratio scatter of topk.hip), then `pad` zero rows.
__global__ __launch_bounds__(256) void bin_gather_rows_kernel(const uint4 *__restrict__ rows, uint32_t row_chunks, uint64_t n_rows,
                                                             uint32_t n, uint32_t pad, uint4 *__restrict__ out) {
    const uint64_t total = (uint64_t)(n + pad) * row_chunks;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
        const uint32_t j = (uint32_t)(i / row_chunks), c = (uint32_t)(i % row_chunks);
        if (j >= n) {
            out[i] = make_uint4(0, 0, 0, 0);
            continue;
        }
        const unsigned long long hsh = (unsigned long long)j * 0x9E3779B97F4A7C15ull;
        const uint64_t src = ((hsh >> 32) * n_rows) >> 32;
        out[i] = rows[src * row_chunks + c];
    }
}

}  // namespace


namespace {

// Query fragments per workgroup tile that fit in LDS for this row length (0: none).
inline int bin_mfma_frags(uint64_t ds, uint64_t n_queries = ~0ull) {
    const uint64_t nkb = ds / 16;
    if (n_queries > 32 && (size_t)64 * (nkb * 128 + 16) + 2048 <= 160 * 1024) return 2;
    if ((size_t)32 * (nkb * 128 + 16) + 2048 <= 160 * 1024) return 1;
    return 0;
}

template <int MODE>
qamd_status launch_bin_gemm(const qamd_bin *h, const qamd_bin_query_batch *b, const uint8_t *rows, uint64_t n_rows, uint32_t q0,
                            int mi, float *out, uint64_t out_pitch, const BatchFilter &filt, hipStream_t s) {
    const bool zx = (h->vp.distance_type == QAMD_DOT) != (h->vp.invert != 0);
    const bool low = MODE != 0 && (!zx) != (MODE == 2);  // multiplier = zx ? +4 : -4
    const size_t lds = (size_t)32 * mi * ((h->ds / 16) * 128 + 16) + 3 * 64 * 4 + 64;
    const uint32_t grid = (uint32_t)std::max(1, device_info().cu_count / 8) * 8;
#define QAMD_BIN_GEMM(M_, LOW_, MI_)                                                                                        \
    do {                                                                                                                   \
        QAMD_LDS_OPT_IN((&bin_gemm_rs_kernel<M_, LOW_, MI_>), 160 * 1024); \
        hipLaunchKernelGGL((bin_gemm_rs_kernel<M_, LOW_, MI_>), dim3(grid), dim3(512), lds, s, rows, (uint32_t)h->ds,      \
                           b->bits.as<uint8_t>(), (uint32_t)b->q_stride, (float)h->vp.dim, zx ? 1 : 0, (uint32_t)n_rows,   \
                           (uint32_t)b->n_queries, q0, out, out_pitch, filt);                                              \
    } while (0)
    constexpr int M = MODE == 0 ? 1 : MODE;
    if (MODE == 0) {
        if (mi == 2) QAMD_BIN_GEMM(0, false, 2);
        else QAMD_BIN_GEMM(0, false, 1);
    } else if (low) {
        if (mi == 2) QAMD_BIN_GEMM(M, true, 2);
        else QAMD_BIN_GEMM(M, true, 1);
    } else {
        if (mi == 2) QAMD_BIN_GEMM(M, false, 2);
        else QAMD_BIN_GEMM(M, false, 1);
    }
#undef QAMD_BIN_GEMM
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// The batch on the matrix cores: per-query pivot from a cached row sample, one filtering pass per query tile
// (32 or 64 queries), candidates scattered to per-query lists, one emit.  status[q] != 0: query q must be redone
// by the exact path (its list over- or underflowed: heavy ties, an unlucky sample).
qamd_status bin_topk_batch_mfma(const qamd_bin *h, const qamd_bin_query_batch *b, uint32_t k, int largest, uint32_t *ids_dev,
                                float *sc_dev, std::vector<uint32_t> &status, hipStream_t s) {
    const uint64_t Q = b->n_queries, n = h->count;
    const int mi = bin_mfma_frags(h->ds, Q);
    const uint32_t TQ = 32u * (uint32_t)mi;
    const uint64_t q_pad = round_up(Q, 64);
    const double want = std::max<double>(Q <= 128 ? 1024.0 : 512.0, 3.0 * k);
    const double s_min = n < (1u << 20) ? 2048.0 : (double)kTopkSample;
    const double s_mem_cap = std::max<double>(s_min, std::floor((double)(6ull << 30) / (4.0 * (double)Q) / 256.0) * 256.0);
    const uint32_t S = (uint32_t)std::min<double>(std::min<double>(524288.0, s_mem_cap),
                                                  std::max<double>(s_min, round_up((uint64_t)(8.0 * (double)n / want), 256)));
    const double target = std::max<double>(want, 8.0 * (double)n / (double)S);
    const uint32_t r = (uint32_t)std::ceil((double)S * target / (double)n);
    if (r > 64) {  // (tiny stores are not routed here)
        std::fill(status.begin(), status.end(), 1u);
        return QAMD_OK;
    }
    // the pivot sample of the store: gathered once per handle, every call uses a prefix (see u8_batch.hip)
    const uint32_t rows_all = (uint32_t)std::min<double>(524288.0, std::max<double>(s_min, (double)round_up((uint64_t)(8.0 * (double)n / 512.0), 256)));
    const uint32_t rc = (uint32_t)(h->ds / 16);
    {
        std::lock_guard<std::mutex> lk(h->sample_mu);
        if (h->sample_count < rows_all) {
            DevBuf c;
            QAMD_TRY(c.alloc((uint64_t)(rows_all + 512) * h->ds));
            hipLaunchKernelGGL(bin_gather_rows_kernel, dim3(device_info().cu_count * 8), dim3(256), 0, s, h->rows.as<uint4>(), rc, n,
                               rows_all, 512u, c.as<uint4>());
            QAMD_HIP(hipGetLastError());
            QAMD_HIP(hipStreamSynchronize(s));
            h->sample_rows = std::move(c);
            h->sample_count = rows_all;
        }
    }
    // rows of 512 / 1024 / 2048 bits, enough queries: the FP4 query-streaming form (QAMD_BIN4=0 / QAMD_BIN4_MIN: developer A/B)
    static const char *e4 = dev_env("QAMD_BIN4"), *e4min = dev_env("QAMD_BIN4_MIN");
    static const char *ers4 = dev_env("QAMD_BIN_RS4");  // developer A/B: 0 = without the row-streaming fp4 form
    const bool fp4_shape = (h->ds == 64 || h->ds == 96 || h->ds == 128 || h->ds == 192) && !(e4 && e4[0] == '0');
    // the whole batch's nibble image in LDS: the row-streaming fp4 form (one pass, no barriers); larger batches: query-streaming
    // A batch of up to kRs4MaxPasses LDS images goes through that form in as many passes over the rows, the queries spread evenly
    // (50M x 1024 bits: 289 queries = two passes of 160: 6.4 ms against the query-streaming form's 10.1; 576 = 2 x 288: 10.4 / 15.7)
    static const char *ers4p = dev_env("QAMD_BIN_RS4_PASSES");  // developer A/B: the most passes the row-streaming form may take
    const uint32_t rs4_cap = rs4_max_queries(h->ds), rs4_most = ers4p ? (uint32_t)atoi(ers4p) : kRs4MaxPasses;
    const uint32_t rs4_passes = rs4_cap ? (uint32_t)((round_up(Q, 32) + rs4_cap - 1) / rs4_cap) : 0u;
    const bool rs4 = fp4_shape && !(ers4 && ers4[0] == '0') && rs4_passes >= 1 && rs4_passes <= rs4_most;
    const uint32_t rs4_per_pass = rs4 ? (uint32_t)round_up((Q + rs4_passes - 1) / rs4_passes, 32) : 0u;  // queries per pass (whole tile pairs)
    const bool qs4 = rs4 || (fp4_shape && Q >= (e4min ? (uint64_t)atoll(e4min) : kQs4MinQueries));
    const uint32_t rs4_lists = pp_waves_per_launch() / 8 * (uint32_t)rs4_waves((int)(h->ds / 16));  // one list per wave of a launch
    const uint32_t n_lists = rs4 ? rs4_passes * rs4_lists
                                 : pp_waves_per_launch() * (qs4 ? (uint32_t)((Q + kQs4Slice - 1) / kQs4Slice) : 1u);
    const double per_wave = 2.0 * target * (double)std::min<uint64_t>(Q, qs4 ? kQs4Slice : TQ) / (double)pp_waves_per_launch();
    // (at least 1024 slots: queries of one batch can be near-duplicates, and then a passing row appends to every
    // query's list at once - 64 entries in one wave's list per such row)
    const uint32_t wave_cap = (uint32_t)std::min<double>(1u << 20, std::max<double>(1024.0, 16.0 * per_wave));
    size_t arena_bytes = 0;
    auto reserve = [&](size_t bytes) {
        const size_t off = arena_bytes;
        arena_bytes += round_up(std::max<size_t>(bytes, 16), 256);
        return off;
    };
    const size_t o_pivots = reserve(q_pad * 4), o_counters = reserve(q_pad * kCounterStride * 4), o_wcounts = reserve((size_t)n_lists * 4);
    const size_t zero_bytes = arena_bytes;
    const size_t o_scores = reserve(Q * (uint64_t)S * 4), o_cand = reserve(Q * (uint64_t)kBatchCap * 8), o_status = reserve((Q + 1) * 4),
                 o_wcand = reserve((uint64_t)n_lists * wave_cap * sizeof(uint4));
    StreamBuf arena;
    QAMD_TRY(arena.alloc(arena_bytes, s));
    char *base = arena.as<char>();
    QAMD_HIP(hipMemsetAsync(base, 0, zero_bytes, s));
    float *pivots = reinterpret_cast<float *>(base + o_pivots);
    uint32_t *counters = reinterpret_cast<uint32_t *>(base + o_counters);
    uint32_t *wave_counts = reinterpret_cast<uint32_t *>(base + o_wcounts);
    float *s_scores = reinterpret_cast<float *>(base + o_scores);
    unsigned long long *cand = reinterpret_cast<unsigned long long *>(base + o_cand);
    uint4 *wave_cand = reinterpret_cast<uint4 *>(base + o_wcand);
    const HostScratch hs = Q + 1 <= 2048 ? host_scratch() : HostScratch{};
    uint32_t *status_dev = hs.dev ? hs.dev : reinterpret_cast<uint32_t *>(base + o_status);
    uint32_t *overflow_dev = status_dev + Q;
    if (hs.host) hs.host[Q] = 0;
    else QAMD_HIP(hipMemsetAsync(overflow_dev, 0, 4, s));
    // sample scores of every query, then every query's pivot
    for (uint32_t q0 = 0; q0 < Q; q0 += TQ)
        QAMD_TRY(launch_bin_gemm<0>(h, b, h->sample_rows.as<uint8_t>(), S, q0, mi, s_scores, S, BatchFilter{}, s));
    hipLaunchKernelGGL(batch_pivot_kernel, dim3((unsigned)q_pad), dim3(1024), 0, s, s_scores, S, (uint64_t)S, r, largest, (uint32_t)Q,
                       pivots, counters);
    BatchFilter f{};
    f.pivot_scores = pivots;
    f.counters = counters;
    f.candidates = cand;
    f.largest = largest;
    f.wave_cap = wave_cap;
    f.wave_cand = wave_cand;
    f.wave_counts = wave_counts;
    if (qs4) {  // ONE pass over the rows for up to 2048 queries (bin_gemm_qs4_kernel), then one scatter
        const bool zx = (h->vp.distance_type == QAMD_DOT) != (h->vp.invert != 0);
        const bool low = (!zx) != (largest == 0);  // multiplier = zx ? +4 : -4; MODE 2 (smallest) flips
        const uint32_t nsteps = (uint32_t)(h->ds / 16);
        StreamBuf prep;
        const size_t frag_bytes = (size_t)q_pad * h->ds * 4;
        QAMD_TRY(prep.alloc(frag_bytes + q_pad * 8, s));
        uint4 *frag = prep.as<uint4>();
        float *q_off = reinterpret_cast<float *>(prep.as<char>() + frag_bytes);
        int *bq = reinterpret_cast<int *>(q_off + q_pad);
        const unsigned pgrid = (unsigned)std::max<uint64_t>((q_pad + 255) / 256, std::min<uint64_t>(2048, (frag_bytes / 16 + 255) / 256));
        if (low)
            hipLaunchKernelGGL(bin_frag4_kernel<true>, dim3(pgrid), dim3(256), 0, s, b->bits.as<uint8_t>(), (uint32_t)b->q_stride, (uint32_t)Q,
                               (uint32_t)q_pad, nsteps, (float)h->vp.dim, zx ? 1 : 0, largest, pivots, frag, q_off, bq);
        else
            hipLaunchKernelGGL(bin_frag4_kernel<false>, dim3(pgrid), dim3(256), 0, s, b->bits.as<uint8_t>(), (uint32_t)b->q_stride, (uint32_t)Q,
                               (uint32_t)q_pad, nsteps, (float)h->vp.dim, zx ? 1 : 0, largest, pivots, frag, q_off, bq);
        const size_t qs_rows = h->ds > 128 ? 96 : 128;
        const size_t lds = 2 * qs_rows * round_up(h->ds * 4, 256) + 4 * qs_rows * 4 + 64 + kQs4Slice * 4;
        const uint32_t grid = (uint32_t)std::max(1, device_info().cu_count / 8) * 8;
        if (rs4)
        for (uint32_t pass = 0; pass < rs4_passes; pass++) {
            const uint64_t q_base = (uint64_t)pass * rs4_per_pass;
            if (q_base >= Q) break;
            const uint32_t nq = (uint32_t)std::min<uint64_t>(rs4_per_pass, Q - q_base);
            const uint32_t n_tiles = (uint32_t)(round_up(nq, 32) / 16);
            const size_t lds4 = (size_t)n_tiles * h->ds * 64 + (size_t)n_tiles * 64 + 64;
            BatchFilter fs = f;
            fs.pivot_scores = pivots + q_base;
            fs.query_base = (uint32_t)q_base;
            fs.wave_base = pass * rs4_lists;
#define QAMD_RS4_NS(M_, LOW_, NS_)                                                                                          \
    do {                                                                                                                   \
        QAMD_LDS_OPT_IN((&bin_gemm_rs4_kernel<M_, LOW_, NS_>), 160 * 1024);                                                 \
        hipLaunchKernelGGL((bin_gemm_rs4_kernel<M_, LOW_, NS_>), dim3(grid), dim3(64 * rs4_waves(NS_)), lds4, s, h->rows.as<uint8_t>(), \
                           frag + (q_base / 16) * nsteps * 64, q_off + q_base, bq + q_base, zx ? 1 : 0, (uint32_t)n, n_tiles, fs); \
    } while (0)
#define QAMD_RS4(M_, LOW_)                                                                                                  \
    do {                                                                                                                   \
        switch (h->ds) {                                                                                                   \
            case 64: QAMD_RS4_NS(M_, LOW_, 4); break;                                                                      \
            case 96: QAMD_RS4_NS(M_, LOW_, 6); break;                                                                      \
            case 128: QAMD_RS4_NS(M_, LOW_, 8); break;                                                                     \
            default: QAMD_RS4_NS(M_, LOW_, 12); break;                                                                     \
        }                                                                                                                  \
    } while (0)
            if (largest) {
                if (low) QAMD_RS4(1, true);
                else QAMD_RS4(1, false);
            } else {
                if (low) QAMD_RS4(2, true);
                else QAMD_RS4(2, false);
            }
#undef QAMD_RS4
#undef QAMD_RS4_NS
            QAMD_HIP(hipGetLastError());
        } else
        for (uint64_t q_base = 0; q_base < Q; q_base += kQs4Slice) {
            const uint32_t nq = (uint32_t)std::min<uint64_t>(kQs4Slice, Q - q_base);
            BatchFilter fs = f;
            fs.pivot_scores = pivots + q_base;
            fs.query_base = (uint32_t)q_base;
            fs.wave_base = (uint32_t)(q_base / kQs4Slice) * pp_waves_per_launch();
#define QAMD_QS4_JT(M_, LOW_, IT_, JT_)                                                                                     \
    do {                                                                                                                   \
        QAMD_LDS_OPT_IN((&bin_gemm_qs4_kernel<M_, LOW_, IT_, JT_>), 160 * 1024); \
        hipLaunchKernelGGL((bin_gemm_qs4_kernel<M_, LOW_, IT_, JT_>), dim3(grid), dim3(512), lds, s, h->rows.as<uint8_t>(), \
                           (uint32_t)h->ds, frag + (q_base / 16) * nsteps * 64, q_off + q_base, bq + q_base, zx ? 1 : 0,   \
                           (uint32_t)n, nq, fs);                                                                           \
    } while (0)
#define QAMD_QS4_IT(M_, LOW_, IT_)                                                                                          \
    do {                                                                                                                   \
        if (h->ds > 128) QAMD_QS4_JT(M_, LOW_, IT_, 6);                                                                    \
        else QAMD_QS4_JT(M_, LOW_, IT_, 8);                                                                                \
    } while (0)
#define QAMD_QS4(M_, LOW_)                                                                                                  \
    do {                                                                                                                   \
        if (Q <= 128) QAMD_QS4_IT(M_, LOW_, 1);                                                                            \
        else if (Q <= 256) QAMD_QS4_IT(M_, LOW_, 2);                                                                       \
        else QAMD_QS4_IT(M_, LOW_, 4);                                                                                     \
    } while (0)
            if (largest) {
                if (low) QAMD_QS4(1, true);
                else QAMD_QS4(1, false);
            } else {
                if (low) QAMD_QS4(2, true);
                else QAMD_QS4(2, false);
            }
#undef QAMD_QS4
#undef QAMD_QS4_IT
#undef QAMD_QS4_JT
            QAMD_HIP(hipGetLastError());
        }
        if (Q <= 4096)
            hipLaunchKernelGGL(wave_scatter_grouped_kernel, dim3(std::min<uint32_t>(n_lists, 128)), dim3(1024), (size_t)Q * 8, s, wave_cand,
                               wave_counts, wave_cap, n_lists, (uint32_t)Q, counters, cand, overflow_dev);
        else
            hipLaunchKernelGGL(wave_scatter_kernel, dim3(n_lists), dim3(256), 0, s, wave_cand, wave_counts, wave_cap, counters, cand,
                               overflow_dev);
    } else
    for (uint32_t q0 = 0; q0 < Q; q0 += TQ) {  // one pass over the rows per query tile; its lists are scattered right away
        if (largest) QAMD_TRY(launch_bin_gemm<1>(h, b, h->rows.as<uint8_t>(), n, q0, mi, nullptr, 0, f, s));
        else QAMD_TRY(launch_bin_gemm<2>(h, b, h->rows.as<uint8_t>(), n, q0, mi, nullptr, 0, f, s));
        if (Q <= 4096)
            hipLaunchKernelGGL(wave_scatter_grouped_kernel, dim3(std::min<uint32_t>(n_lists, 128)), dim3(1024), (size_t)Q * 8, s, wave_cand,
                               wave_counts, wave_cap, n_lists, (uint32_t)Q, counters, cand, overflow_dev);
        else
            hipLaunchKernelGGL(wave_scatter_kernel, dim3(n_lists), dim3(256), 0, s, wave_cand, wave_counts, wave_cap, counters, cand,
                               overflow_dev);
    }
    if (k <= kSmallTopkMaxK)
        hipLaunchKernelGGL(batch_emit_wave_kernel, dim3((unsigned)Q), dim3(1024), 0, s, cand, counters, n, k, largest, ids_dev, sc_dev,
                           status_dev);
    else
        hipLaunchKernelGGL(batch_emit_kernel, dim3((unsigned)Q), dim3(1024), 0, s, cand, counters, n, k, largest, ids_dev, sc_dev,
                           status_dev);
    QAMD_HIP(hipGetLastError());
    uint32_t overflow = 0;
    if (hs.host) {
        QAMD_HIP(hipStreamSynchronize(s));
        std::copy(hs.host, hs.host + Q, status.begin());
        overflow = hs.host[Q];
    } else {
        std::vector<uint32_t> back(Q + 1);
        QAMD_TRY(copy_out(back.data(), QAMD_MEM_HOST, status_dev, (Q + 1) * 4, s));
        std::copy(back.begin(), back.begin() + Q, status.begin());
        overflow = back[Q];
    }
    if (overflow) std::fill(status.begin(), status.end(), 1u);
    return QAMD_OK;
}

}  // namespace


extern "C" {

qamd_status qamd_bin_encode_query_batch(const qamd_bin *h, const float *queries, uint64_t n_queries, uint64_t qdim,
                                        qamd_mem queries_mem, void *stream, qamd_bin_query_batch **batch_io) {
    if (!h || !batch_io || (!queries && n_queries && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint64_t nb = row_bytes_of(qdim, h->store), ds = device_stride_of(nb);
    if (n_queries && nb != h->nb)
        return fail(QAMD_ERR_ARGUMENTS, "queries have %llu bytes, rows have %llu", (unsigned long long)nb,
                    (unsigned long long)h->nb);
    qamd_bin_query_batch *b = *batch_io;
    std::unique_ptr<qamd_bin_query_batch> fresh;
    if (!b) {
        fresh.reset(new qamd_bin_query_batch);
        b = fresh.get();
        b->device = h->device;
    }
    const uint64_t q_stride = round_up(ds, 16);
    const size_t need = std::max<size_t>(q_stride * n_queries, 16);
    if (b->bits.bytes < need || b->q_stride != q_stride) QAMD_TRY(b->bits.alloc(need, true));
    b->nb = nb;
    b->ds = ds;
    b->q_stride = q_stride;
    b->n_queries = n_queries;
    if (n_queries && qdim) {
        DevBuf qtmp;
        const void *qd = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(queries, queries_mem, n_queries * qdim * 4, qtmp, s, &qd, &staged));
        // the row encoder with one "row" per query and the batch's stride (:288-291 is encode_vector itself)
        int grid = grid_for(n_queries, kBlock / 64, 8);
        hipLaunchKernelGGL(bin_encode_kernel, dim3(grid), dim3(kBlock), 0, s, static_cast<const float *>(qd), n_queries,
                           (uint32_t)qdim, (uint32_t)(q_stride / 4), b->bits.as<uint32_t>(), (uint64_t)0);
        QAMD_HIP(hipGetLastError());
        if (staged) QAMD_HIP(hipStreamSynchronize(s));
    }
    if (fresh) *batch_io = fresh.release();
    return QAMD_OK;
}

void qamd_bin_query_batch_free(qamd_bin_query_batch *b) { delete b; }

static qamd_status bin_check_batch(const qamd_bin *h, const qamd_bin_query_batch *b) {
    if (!h || !b) return fail(QAMD_ERR_ARGUMENTS, "null handle or query batch");
    if (b->n_queries && b->nb != h->nb)
        return fail(QAMD_ERR_ARGUMENTS, "queries have %llu bytes, rows have %llu", (unsigned long long)b->nb,
                    (unsigned long long)h->nb);
    return QAMD_OK;
}

// Many (query, id list) pairs in one launch (lists.hpp): out[p] = score_point(query l, ids[p]).
qamd_status qamd_bin_score_ids_batch(const qamd_bin *h, const qamd_bin_query_batch *b, const uint32_t *list_offsets,
                                     uint32_t n_lists, const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem, float *out,
                                     qamd_mem out_mem, void *stream) {
    QAMD_TRY(bin_check_batch(h, b));
    if (n_lists > b->n_queries)
        return fail(QAMD_ERR_ARGUMENTS, "%u lists, but the batch holds %llu queries", n_lists, (unsigned long long)b->n_queries);
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    return run_lists(list_offsets, n_lists, ids, n_ids, nullptr, lists_mem, out, out_mem, h->count, s, [&](const ListArgs &a) {
        return pairs_launch(h, nullptr, b->bits.as<uint8_t>(), b->q_stride, a.offsets, a.n_lists, nullptr, a.ids, a.n_pairs,
                            a.out, s);
    });
}

qamd_status qamd_bin_score_batch(const qamd_bin *h, const qamd_bin_query_batch *b, float *out, qamd_mem out_mem,
                                 void *stream) {
    QAMD_TRY(bin_check_batch(h, b));
    if (h->count == 0 || b->n_queries == 0) return QAMD_OK;
    if (!out) return fail(QAMD_ERR_ARGUMENTS, "out is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    StreamBuf tmp;
    float *out_dev = out;
    if (out_mem == QAMD_MEM_HOST) {
        QAMD_TRY(tmp.alloc(b->n_queries * h->count * 4, s));
        out_dev = tmp.as<float>();
    }
    const uint8_t *bits = b->bits.as<uint8_t>();
    uint64_t q = 0;
    // 5 queries and more: tiles of 32 / 64 queries on the matrix cores (same scores bit for bit: every f32
    // of the epilogue is an exact integer); the row bits are read once per tile instead of once per 8 queries
    // (measured at 50M x 1024: one 32-query tile 2.28 ms whatever the batch; the vector-ALU passes 1.81 ms for 4
    // queries, 2.6 for 8, and from there 2.6 ms per 8)
    if (b->n_queries >= 5 && h->count >= 4096 && fused_capable(h) && bin_mfma_frags(h->ds) != 0 && h->vp.dim >= 64) {
        const int mi = bin_mfma_frags(h->ds, b->n_queries);
        for (; q < b->n_queries; q += 32 * mi)  // (a partly filled last tile computes zero queries and stores nothing for them)
            QAMD_TRY(launch_bin_gemm<0>(h, b, h->rows.as<uint8_t>(), h->count, (uint32_t)q, mi, out_dev, h->count, BatchFilter{}, s));
    }
    while (q < b->n_queries) {
        const uint64_t left = b->n_queries - q;
        const uint8_t *qb = bits + q * b->q_stride;
        float *o = out_dev + q * h->count;
        if (left >= 8 && multi_step<8>(h, qb, b->q_stride, o, s)) q += 8;
        else if (left >= 4 && multi_step<4>(h, qb, b->q_stride, o, s)) q += 4;
        else if (left >= 2 && multi_step<2>(h, qb, b->q_stride, o, s)) q += 2;
        else {
            QAMD_TRY(scan_bits(h, qb, o, s));
            q += 1;
        }
    }
    QAMD_HIP(hipGetLastError());
    if (out_mem == QAMD_MEM_HOST) return copy_out(out, QAMD_MEM_HOST, out_dev, b->n_queries * h->count * 4, s);
    return QAMD_OK;
}

qamd_status qamd_bin_topk_batch(const qamd_bin *h, const qamd_bin_query_batch *b, uint32_t k, int largest,
                                uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    QAMD_TRY(bin_check_batch(h, b));
    if (k == 0 || b->n_queries == 0) return QAMD_OK;
    if (!out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null output");
    QAMD_ON_DEVICE(h->device);
    const uint8_t *bits = b->bits.as<uint8_t>();
    const uint64_t qs = b->q_stride;
    BatchScan scan;
    scan.filter_capable = fused_capable(h);
    scan.scan_scores = [&](uint32_t q, float *scores, hipStream_t st) { return scan_bits(h, bits + q * qs, scores, st); };
    scan.scan_filter = [&](uint32_t q, const TopkFilter &f, hipStream_t st) {
        return scan_bits(h, bits + q * qs, nullptr, st, &f);
    };
    scan.score_ids = [&](uint32_t q, const uint32_t *ids, uint64_t n_ids, float *out, hipStream_t st) {
        return words_launch(h, reinterpret_cast<const uint32_t *>(bits + q * qs), ids, n_ids, out, st);
    };
    scan.topk_small = [&](uint32_t q, uint32_t *ids, float *sc, hipStream_t st, qamd_status &status) {
        return bin_topk_small(h, bits + (uint64_t)q * qs, k, largest, ids, sc, QAMD_MEM_DEVICE, st, status);
    };
    // up to 8 queries share one pass over the rows (the filtering form of bin_scan_multi_kernel)
    scan.scan_filter_multi = [&](uint32_t q, uint32_t left, const TopkFilterSlices &sl, hipStream_t st, qamd_status &status) -> uint32_t {
        const uint8_t *qb = bits + (uint64_t)q * qs;
        uint32_t took = 0;
        if (left >= 8 && multi_step<8>(h, qb, qs, nullptr, st, &sl)) took = 8;
        else if (left >= 4 && multi_step<4>(h, qb, qs, nullptr, st, &sl)) took = 4;
        else if (left >= 2 && multi_step<2>(h, qb, qs, nullptr, st, &sl)) took = 2;
        if (took && hipGetLastError() != hipSuccess) status = fail(QAMD_ERR_DEVICE, "binary multi-query filter launch failed");
        return took;
    };
    // 12 queries and more on stores of 32k rows and more (below that the filtering vector-ALU passes, 0.22 ms per
    // query at 50M rows, are cheaper than one matrix-core pass): bin_gemm_rs_kernel; queries whose
    // candidate list over- or underflowed there (heavy ties at small dims) go through the path below one by one
    const uint64_t Q = b->n_queries;
    hipStream_t s = as_stream(stream);
    static const char *emin = dev_env("QAMD_BIN_MFMA_MIN");  // developer A/B: the smallest batch the matrix cores take
    // rows of 512 / 768 / 1024 / 1536 bits: a pass of the row-streaming fp4 kernel costs 1.4-1.5 ms per 50M x 1024 whatever the batch,
    // the vector-ALU scan 1.04 / 1.98 / 1.26 / 2.2 / 1.8 / 3.7 ms at 2 / 3 / 4 / 5 / 8 / 11 queries (profiles/r04_bin_batch.txt)
    const bool fp4_rows = h->ds == 64 || h->ds == 96 || h->ds == 128 || h->ds == 192;
    const bool enough = emin ? Q >= (uint64_t)atoll(emin) : (fp4_rows ? (Q == 3 || Q >= 5) : Q >= kRs4MinQueries);
    if (enough && h->count >= 32768 && k <= 1024 && fused_capable(h) && bin_mfma_frags(h->ds) != 0 && h->vp.dim >= 64) {
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
        QAMD_TRY(bin_topk_batch_mfma(h, b, k, largest, ids_dev, sc_dev, status, s));
        for (uint64_t q = 0; q < Q; q++) {
            if (!status[q]) continue;
            BatchScan one;
            one.filter_capable = scan.filter_capable;
            one.scan_scores = [&, q](uint32_t, float *scores, hipStream_t st) { return scan.scan_scores((uint32_t)q, scores, st); };
            one.scan_filter = [&, q](uint32_t, const TopkFilter &f, hipStream_t st) { return scan.scan_filter((uint32_t)q, f, st); };
            one.score_ids = [&, q](uint32_t, const uint32_t *ids, uint64_t n_ids, float *out, hipStream_t st) {
                return scan.score_ids((uint32_t)q, ids, n_ids, out, st);
            };
            one.topk_small = [&, q](uint32_t, uint32_t *ids, float *sc, hipStream_t st, qamd_status &status1) {
                return scan.topk_small((uint32_t)q, ids, sc, st, status1);
            };
            QAMD_TRY(fused_topk_batch(h->count, 1, k, largest, ids_dev + q * k, sc_dev + q * k, QAMD_MEM_DEVICE, s, one));
        }
        if (out_mem == QAMD_MEM_HOST) {
            QAMD_TRY(copy_out(out_ids, QAMD_MEM_HOST, ids_dev, Q * k * 4, s));
            QAMD_TRY(copy_out(out_scores, QAMD_MEM_HOST, sc_dev, Q * k * 4, s));
        } else {
            QAMD_HIP(hipStreamSynchronize(s));
        }
        return QAMD_OK;
    }
    return fused_topk_batch(h->count, (uint32_t)b->n_queries, k, largest, out_ids, out_scores, out_mem, s, scan);
}

}  // extern "C"


// ============================================================================= streaming encode
// EncodedVectorsBin::encode (:165-191) walks its iterator ONCE, pushing one packed row per vector
// (EncodedStorageBuilder::push_vector_data, encoded_storage.rs:17-25); here in bounded batches.
struct qamd_bin_encoder {
    int device = 0;
    hipStream_t stream = nullptr;
    qamd_stop_fn stop = nullptr;
    void *stop_user = nullptr;
    std::unique_ptr<qamd_bin> h;
    uint64_t pushed = 0;
    DevBuf stage;
};

extern "C" {

qamd_status qamd_bin_encoder_begin(const qamd_vector_parameters *vp, qamd_bits_store store, qamd_stop_fn stop,
                                   void *stop_user, void *stream, qamd_bin_encoder **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    QAMD_ON_DEVICE(current_device());
    std::unique_ptr<qamd_bin_encoder> e(new qamd_bin_encoder);
    e->device = current_device();
    e->stream = as_stream(stream);
    e->stop = stop;
    e->stop_user = stop_user;
    e->h.reset(new qamd_bin);
    e->h->device = e->device;
    e->h->vp = *vp;
    e->h->store = store;
    e->h->count = vp->count;
    QAMD_TRY(alloc_store(e->h.get()));
    *out = e.release();
    return QAMD_OK;
}

qamd_status qamd_bin_encoder_push(qamd_bin_encoder *e, const float *batch, uint64_t n_rows, qamd_mem batch_mem) {
    if (!e || (!batch && n_rows && e->h->vp.dim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (e->stop && e->stop(e->stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :174-176
    if (e->pushed + n_rows > e->h->count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)(e->pushed + n_rows), (unsigned long long)e->h->count);
    QAMD_ON_DEVICE(e->device);
    const uint64_t dim = e->h->vp.dim;
    const uint64_t piece_rows = std::max<uint64_t>(1, (256ull << 20) / std::max<uint64_t>(dim * 4, 1));
    for (uint64_t r = 0; r < n_rows && dim; r += piece_rows) {
        const uint64_t nr = std::min(piece_rows, n_rows - r);
        const void *src = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(batch + r * dim, batch_mem, nr * dim * 4, e->stage, e->stream, &src, &staged));
        QAMD_TRY(launch_bin_encode(e->h.get(), static_cast<const float *>(src), nr, e->pushed + r, e->stream));
        if (staged) QAMD_HIP(hipStreamSynchronize(e->stream));  // the staging buffer is reused
    }
    e->pushed += n_rows;
    return QAMD_OK;
}

qamd_status qamd_bin_encoder_finish(qamd_bin_encoder *e, qamd_bin **out) {
    if (!e || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_bin_encoder> own(e);
    if (e->pushed != e->h->count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)e->pushed, (unsigned long long)e->h->count);
    QAMD_ON_DEVICE(e->device);
    QAMD_HIP(hipStreamSynchronize(e->stream));
    *out = e->h.release();
    return QAMD_OK;
}

void qamd_bin_encoder_abort(qamd_bin_encoder *e) {
    if (!e) return;
    DeviceGuard g(e->device);
    (void)hipStreamSynchronize(e->stream);
    delete e;
}

}  // extern "C"

#ifdef QAMD_DEV
// Developer-only accessors for the tuning harness (tune.hip): libquantization_amd_dev.so only.
extern "C" __attribute__((visibility("default"))) const void *qamd_dev_bin_rows(const qamd_bin *h) { return h->rows.ptr; }
extern "C" __attribute__((visibility("default"))) const void *qamd_dev_bin_query_ptr(const qamd_bin_query *q) {
    return q->buf.ptr;
}
#endif
