// Scalar u8 quantizer on MI355X (gfx950): encode, query encode, and the query-vs-store scan.
//
// Host class mirrors EncodedVectorsU8 (quantization/src/encoded_vectors_u8.rs); the kernels
// replace the per-pair SIMD FFI (quantization/cpp/avx2.c:25-122) by whole-store launches.
//
// HBM layout (library-owned copy).  The reference row is AoS [vector_offset f32][codes]
// with stride actual_dim+4 (772 B at dim 768: only 4-byte aligned).  On device it is split
// into  codes[count_padded][actual_dim]  (rows 16-byte aligned, contiguous) and
// offsets[count] f32.  Same 772 algorithmic bytes per scored row; every code load is an
// aligned 16-byte `global_load_dwordx4`.
//
// Scan mapping (HBM-bound integer work, no MFMA): a row is read by G = min(16, pow2(chunks))
// adjacent lanes, 16 B per lane per iteration, so one wave-load covers 64/G consecutive rows
// = fully used 128-B lines.  The query's 16-byte chunks live in VGPRs for the whole kernel.
// v_dot4_u32_u8 accumulates in 32-bit integers (exact), a log2(G)-step cross-lane add
// finishes the row, and the f32 epilogue is evaluated in the reference's order
// ((multiplier*s) + q.offset) + vector_offset  (encoded_vectors_u8.rs:347), no FMA.
//
// Exactness: codes <= 127 => the integer sum is what every reference kernel computes; the
// reference converts lane sums to f32 before adding them (avx2.c:59-62), which is the same
// number whenever the sum is < 2^24 (always for actual_dim <= 1040).  Above that the GPU
// returns the exact integer rounded ONCE to f32 (= the reference's scalar path,
// encoded_vectors_u8.rs:158), or — mode QAMD_U8_LANES_AVX2 — reproduces avx2.c's 8-lane f32
// summation bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

#include "common.hpp"
#include "lists.hpp"
#include "multi_reduce.hpp"
#include "topk.hpp"
#include "topk_device.hpp"
#include "u8_internal.hpp"

#pragma clang fp contract(off)

using namespace qamd;

namespace {

constexpr int kBlock = 256;
constexpr int kScanBlock = 512;  // scan workgroup (tuning sweep: 512 and 1024 tie, 256 is ~1% behind)
constexpr uint64_t kRowPad = 1024;  // rows are padded so that every wave tile is in bounds
constexpr uint64_t kQuantileSample = 100000;  // QUANTILE_SAMPLE_SIZE, quantile.rs:3

enum Epilogue : int { EPI_POINT = 0, EPI_INTERNAL = 1 };

// ------------------------------------------------------------------------------ device helpers
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Streamed-once 16-byte loads: `global_load_dwordx4 ... nt`.
__device__ __forceinline__ uint4 ld_nt(const uint4 *p) {
    u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float4 ld_nt(const float4 *p) {
    f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ uint32_t dot16(const uint4 &a, const uint4 &b, uint32_t acc) {
    acc = __builtin_amdgcn_udot4(a.x, b.x, acc, false);
    acc = __builtin_amdgcn_udot4(a.y, b.y, acc, false);
    acc = __builtin_amdgcn_udot4(a.z, b.z, acc, false);
    acc = __builtin_amdgcn_udot4(a.w, b.w, acc, false);
    return acc;
}

__device__ __forceinline__ uint32_t sad16(const uint4 &a, const uint4 &b, uint32_t acc) {
    acc = __builtin_amdgcn_sad_u8(a.x, b.x, acc);
    acc = __builtin_amdgcn_sad_u8(a.y, b.y, acc);
    acc = __builtin_amdgcn_sad_u8(a.z, b.z, acc);
    acc = __builtin_amdgcn_sad_u8(a.w, b.w, acc);
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

// encoded_vectors_u8.rs:347 / :409.  s is the exact integer pair sum.
__device__ __forceinline__ float epilogue(float multiplier, uint32_t s, float q_off, float v_off,
                                          float diff, int mode) {
    float sf = (float)(int32_t)s;
    float ms = multiplier * sf;
    if (mode == EPI_POINT) return (ms + q_off) + v_off;
    return ms + ((q_off + v_off) - diff);
}

// encoded_vectors_u8.rs:234-237  ((v-offset)/alpha).clamp(0,127) as u8
__device__ __forceinline__ uint32_t f32_to_u8(float v, float alpha, float offset) {
    float x = (v - offset) / alpha;  // IEEE division (hipcc: correctly rounded f32 divide)
    x = (x < 0.0f) ? 0.0f : x;
    x = (x > 127.0f) ? 127.0f : x;
    return (x != x) ? 0u : (uint32_t)x;
}

// f32_to_u8 for four values without the division, for all but ~6 values in 100 000.
// x = RN((v - offset) / alpha) and y = RN((v - offset) * RN(1 / alpha)) differ by at most
// |x| * (2^-23 + 2^-24) < 2.3e-5 for |x| <= 128 (three roundings of relative size 2^-24), so whenever y
// lies further than 2^-15 = 3.05e-5 from every integer -- or anywhere outside (-2^-15, 128 + 2^-15),
// where the clamp decides -- trunc(clamp(x)) == trunc(clamp(y)).  If any of the four values is inside
// one of those bands (or NaN: both tests fail), all four take the reference's division: ONE branch per
// float4, taken by 1.5 % of the waves.  The host selects this path only for a normal alpha in
// 2^-60 .. 2^60 (1 / alpha exact to half an ulp, no overflow).  Codes are the reference's, bit for bit
// (tests/test_gpu_u8.py: boundary sweep).  Returns the four codes packed into a dword.
__device__ __forceinline__ bool quant_decided(float y) {
    constexpr float kBand = 3.0517578125e-05f;  // 2^-15
    const float fr = __builtin_amdgcn_fractf(y);
    return (__builtin_fabsf(fr - 0.5f) <= 0.5f - kBand) || (__builtin_fabsf(y - 64.0f) >= 64.0f + kBand);
}
__device__ __forceinline__ uint32_t quant_code(float y) {  // y is not NaN here
    return (uint32_t)__builtin_amdgcn_fmed3f(y, 0.0f, 127.0f);
}
__device__ __forceinline__ uint32_t f32x4_to_u8x4_fast(const float4 &f, float alpha, float offset, float r_alpha) {
    const float y0 = (f.x - offset) * r_alpha, y1 = (f.y - offset) * r_alpha, y2 = (f.z - offset) * r_alpha,
                y3 = (f.w - offset) * r_alpha;
    if (quant_decided(y0) && quant_decided(y1) && quant_decided(y2) && quant_decided(y3))
        return quant_code(y0) | (quant_code(y1) << 8) | (quant_code(y2) << 16) | (quant_code(y3) << 24);
    return f32_to_u8(f.x, alpha, offset) | (f32_to_u8(f.y, alpha, offset) << 8) | (f32_to_u8(f.z, alpha, offset) << 16) |
           (f32_to_u8(f.w, alpha, offset) << 24);
}

// ------------------------------------------------------------------------------ scan kernel
// One wave covers ONE tile of (64/G)*UNROLL consecutive rows and exits; the grid is one wave
// per tile.  Measured on MI355X (10M x 768, tools/tune_u8.py): this non-persistent form
// streams at 6.5-6.7 TB/s, a persistent grid-stride loop over the same body at 5.9-6.4 —
// workgroups are dispatched in order, so the resident waves always cover one compact, moving
// window of the store.  ITERS = ceil(row_chunks / G).
// EXACT: row_chunks == G*ITERS, so no lane is ever past the row end.  Otherwise loads stay
// unconditional (clamped address + select) so that they still issue back to back.
// FILTER: fused top-k mode — no score is written; rows at least as good as the pivot are
// appended to the candidate buffer (topk_device.hpp).
template <int G, int ITERS, int UNROLL, bool IS_L1, bool EXACT, bool FILTER>
__global__ __launch_bounds__(kScanBlock) void u8_scan_kernel(
    const uint4 *__restrict__ codes, const float *__restrict__ offsets,
    const uint4 *__restrict__ qcodes, const float *__restrict__ q_off_p, float multiplier,
    uint32_t n_rows, uint32_t row_chunks, float *__restrict__ out, TopkFilter filt) {
    constexpr int RW = 64 / G;
    constexpr int TILE = RW * UNROLL;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G;
    const int rslot = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) >> 6;
    const uint64_t base = wave * TILE;
    if (base >= n_rows) return;

    uint4 v[UNROLL][ITERS];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const uint64_t row = base + u * RW + rslot;
        const uint4 *p = codes + row * row_chunks;
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint32_t c = sub + it * G;
            if (EXACT) {
                v[u][it] = ld_nt(p + c);
            } else {
                const uint32_t cc = c < row_chunks ? c : row_chunks - 1;
                uint4 t = ld_nt(p + cc);
                const bool in = c < row_chunks;
                v[u][it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
            }
        }
    }
    uint4 q[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t c = sub + it * G;
        if (EXACT) {
            q[it] = qcodes[c];  // unconditional: no branch around the load
        } else {
            const uint4 t = qcodes[c < row_chunks ? c : row_chunks - 1];
            const bool in = c < row_chunks;
            q[it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
        }
    }
    const float q_off = *q_off_p;
    // Score stores: lane (rslot, sub = u % G) keeps row u's score, so that each group of G
    // tiles rows leaves the wave as ONE store of (64/G)*G = 64 consecutive floats at most —
    // whole 64-byte segments, which is what makes the nontemporal store a win (+2.8 %);
    // nt stores of the 16-byte per-row-group pieces were 15 % slower than plain ones.
    constexpr int NGROUPS = (UNROLL + G - 1) / G;
    float v_off[NGROUPS];
#pragma unroll
    for (int g = 0; g < NGROUPS; g++)  // offsets[] is padded like codes[]: no guard needed
        v_off[g] = offsets[base + (uint64_t)(g * G + sub) * RW + rslot];
    float mine = 0.0f;
    uint32_t pivot = 0;
    if (FILTER) pivot = *filt.pivot_key;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        uint32_t acc = 0;
#pragma unroll
        for (int it = 0; it < ITERS; it++)
            acc = IS_L1 ? sad16(v[u][it], q[it], acc) : dot16(v[u][it], q[it], acc);
        acc = group_sum<G>(acc);
        if (sub == (u % G)) mine = epilogue(multiplier, acc, q_off, v_off[u / G], 0.0f, EPI_POINT);
        if ((u % G) == G - 1 || u == UNROLL - 1) {
            constexpr int dummy = 0;
            (void)dummy;
            const int first = (u / G) * G;
            const uint64_t row = base + (uint64_t)(first + sub) * RW + rslot;
            if (sub <= u - first && row < n_rows) {
                if (FILTER) topk_offer(filt, pivot, mine, (uint32_t)row);
                else __builtin_nontemporal_store(mine, out + row);
            }
        }
    }
}

// NQ (2, 4, 8) queries per row read on the vector ALU -- the route of qamd_u8_topk_batch /
// score_batch for a handful of queries over a large store, where the matrix-core kernel is bound
// by what its LDS-DMA path moves (4.6 TB/s of row bytes: 1.67 ms per 10M x 768 whatever the batch)
// while this one streams the rows like the single-query scan: the row's 16-byte pieces are loaded
// once (nt) and v_dot4'd against NQ query rows held in registers (NQ * ITERS * 4 VGPRs).  Lane
// (row slot, sub = j) keeps query j's score of the row.  Same integer sum and the same f32
// epilogue as u8_scan_kernel => the same score bits.  G >= NQ.
// FILTER: no score is written; the lane offers its row to query j's candidate lists (slices).
template <int G, int ITERS, int UNROLL, int NQ, bool IS_L1, bool EXACT, bool FILTER>
__global__ __launch_bounds__(kScanBlock) void u8_scan_multi_kernel(
    const uint4 *__restrict__ codes, const float *__restrict__ offsets, const uint8_t *__restrict__ qcodes,
    uint32_t q_pitch, const float *__restrict__ q_offs, uint32_t nq_valid /* <= NQ: queries really there */,
    float multiplier, uint32_t n_rows, uint32_t row_chunks, float *__restrict__ out /* [NQ][out_pitch] */,
    uint64_t out_pitch, TopkFilterSlices slices) {
    static_assert(G >= NQ, "one lane of the row group per query");
    constexpr int RW = 64 / G;
    constexpr int TILE = RW * UNROLL;
    // A wave walks kMultiTiles consecutive tiles: the NQ query rows (NQ * ITERS 16-byte loads per lane)
    // are fetched once per wave, so they must be amortised over more rows than one tile holds
    // (with one tile per wave the query loads were as many bytes as the rows themselves).
    constexpr int TPW = NQ;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, rslot = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) >> 6;
    if (wave * TILE * TPW >= n_rows) return;
    uint4 q[NQ][ITERS];
#pragma unroll
    for (int j = 0; j < NQ; j++) {  // an unused slot (3 queries in a 4-wide pass) re-reads the last query
        const uint4 *qj = reinterpret_cast<const uint4 *>(qcodes + (size_t)((uint32_t)j < nq_valid ? j : nq_valid - 1) * q_pitch);
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const uint32_t c = sub + it * G;
            const bool in = EXACT || c < row_chunks;
            const uint4 t = qj[in ? c : row_chunks - 1];
            q[j][it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
        }
    }
    // lane (row slot, sub) ends up with the total of query multi_query_of<NQ>(sub) (multi_reduce.hpp)
    const int my_q = multi_query_of<NQ>(sub);
    const bool owner = sub < NQ && (uint32_t)my_q < nq_valid;
    const float q_off = q_offs[owner ? my_q : 0];
    for (int tile = 0; tile < TPW; tile++) {
        const uint64_t base = (wave * TPW + tile) * TILE;
        if (base >= n_rows) break;
        uint4 v[UNROLL][ITERS];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint4 *p = codes + (base + u * RW + rslot) * row_chunks;  // rows are padded: no guard
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const uint32_t c = sub + it * G;
                const bool in = EXACT || c < row_chunks;
                const uint4 t = ld_nt(p + (in ? c : row_chunks - 1));
                v[u][it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint64_t row = base + (uint64_t)u * RW + rslot;
            const float v_off = offsets[row];  // padded like codes[]
            uint32_t acc[NQ];
#pragma unroll
            for (int j = 0; j < NQ; j++) {
                acc[j] = 0;
#pragma unroll
                for (int it = 0; it < ITERS; it++)
                    acc[j] = IS_L1 ? sad16(v[u][it], q[j][it], acc[j]) : dot16(v[u][it], q[j][it], acc[j]);
            }
            const float mine = epilogue(multiplier, multi_reduce<G, NQ>(acc, sub), q_off, v_off, 0.0f, EPI_POINT);
            if (owner && row < n_rows) {
                if (FILTER) {
                    const TopkFilter f = topk_filter_of(slices, (uint32_t)my_q);
                    topk_offer(f, *f.pivot_key, mine, (uint32_t)row);
                } else {
                    __builtin_nontemporal_store(mine, out + (uint64_t)my_q * out_pitch + row);
                }
            }
        }
    }
}

// Single-launch top-k for small stores (topk.hpp small_topk): workgroup b scores rows
// [b * rows_per_wg, (b + 1) * rows_per_wg) exactly like u8_scan_kernel (same loads, same integer
// sum, same f32 epilogue => the same score bits), but a score never goes to HBM: it becomes a
// 64-bit key (order-preserving score bits << 32 | row) in the wave's LDS staging row, and the wave /
// workgroup / last-arriver merges of topk_device.hpp keep the best k.  100k x 768: one launch
// replaces the sample / pivot / filtering-scan / sort chain (66 us).
// The body, given the query's pieces in registers (q) and its offset:
template <int G, int ITERS, bool IS_L1, bool EXACT>
__device__ __forceinline__ void u8_topk_small_body(const uint4 *__restrict__ codes, const float *__restrict__ offsets,
                                                   const uint4 (&q)[ITERS], float q_off, float multiplier, uint32_t n_rows,
                                                   uint32_t row_chunks, uint32_t rows_per_wg, const SmallTopk &p,
                                                   unsigned long long (*lds)[64]) {
    unsigned long long(*lists)[64] = lds;
    constexpr int RW = 64 / G;
    constexpr int UNROLL = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, rslot = lane / G;
    unsigned long long *stage = lds[kSmallTopkWaves + wave];
    const uint64_t wg_base = (uint64_t)blockIdx.x * rows_per_wg;
    SmallTopkWave acc_list;
    for (uint32_t tile = wave * UNROLL; tile * RW < rows_per_wg; tile += kSmallTopkWaves * UNROLL) {
        uint4 v[UNROLL][ITERS];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            // rows past the workgroup's range or the store still load from a valid address
            const uint64_t row = wg_base + (uint64_t)(tile + u) * RW + rslot;
            const uint64_t rc = row < n_rows ? row : (uint64_t)n_rows - 1;
            const uint4 *src = codes + rc * row_chunks;
#pragma unroll
            for (int it = 0; it < ITERS; it++) {
                const uint32_t c = sub + it * G;
                const uint4 t = ld_nt(src + ((EXACT || c < row_chunks) ? c : row_chunks - 1));
                const bool in = EXACT || c < row_chunks;
                v[u][it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t local = (tile + u) * RW + rslot;
            const uint64_t row = wg_base + local;
            uint32_t acc = 0;
#pragma unroll
            for (int it = 0; it < ITERS; it++) acc = IS_L1 ? sad16(v[u][it], q[it], acc) : dot16(v[u][it], q[it], acc);
            acc = group_sum<G>(acc);
            if (sub == 0) {
                unsigned long long key = ~0ull;
                if (local < rows_per_wg && row < n_rows) {
                    const float sc = epilogue(multiplier, acc, q_off, offsets[row], 0.0f, EPI_POINT);
                    key = ((unsigned long long)topk_ordered_bits(sc, p.largest != 0) << 32) | (uint32_t)row;
                }
                stage[acc_list.fill + rslot] = key;
            }
            acc_list.fill += RW;
            if (acc_list.fill == 64) small_topk_flush(acc_list, stage, lane);
        }
    }
    if (acc_list.fill) small_topk_flush(acc_list, stage, lane);
    small_topk_finish(acc_list.best, lists, p);
}

template <int G, int ITERS, bool IS_L1, bool EXACT>
__global__ __launch_bounds__(1024) void u8_topk_small_kernel(
    const uint4 *__restrict__ codes, const float *__restrict__ offsets, const uint4 *__restrict__ qcodes,
    const float *__restrict__ q_off_p, float multiplier, uint32_t n_rows, uint32_t row_chunks, uint32_t rows_per_wg,
    SmallTopk p) {
    __shared__ unsigned long long lds[2 * kSmallTopkWaves][64];  // [0, 16): tournament lists, [16, 32): staging rows
    const int lane = threadIdx.x & 63, sub = lane % G;
    uint4 q[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t c = sub + it * G;
        const uint4 t = qcodes[(EXACT || c < row_chunks) ? c : row_chunks - 1];
        const bool in = EXACT || c < row_chunks;
        q[it] = make_uint4(in ? t.x : 0, in ? t.y : 0, in ? t.z : 0, in ? t.w : 0);
    }
    u8_topk_small_body<G, ITERS, IS_L1, EXACT>(codes, offsets, q, *q_off_p, multiplier, n_rows, row_chunks, rows_per_wg, p, lds);
}

// encode_query FOLDED INTO the single-launch top-k: one launch per search on a small store.  The f32
// query arrives BY VALUE in the kernel arguments (<= 896 values: the argument block is 4 KiB), so no
// copy call, no mapped-memory read over PCIe and no second, dependent launch stands in front of the
// scan.  Every workgroup quantises the query itself -- thread t makes dword t of the codes with the
// encoder's own f32_to_u8 and padding (encoded_vectors_u8.rs:290-329), the code sums are exact
// integers (127^2 * 896 < 2^24), the offset is formed in the reference's order -- keeps the codes in
// LDS, and goes on as u8_topk_small_kernel: same codes and offset => the same score bits.  Workgroup 0
// also leaves offset + codes in the query object's buffer, so after this launch the object is an
// ordinary encoded query for whoever uses it next.
constexpr uint32_t kFusedQueryDims = 896;
struct QueryByValue {
    float v[kFusedQueryDims];
};
template <int G, int ITERS, bool IS_L1, bool EXACT>
__global__ __launch_bounds__(1024) void u8_topk_small_fused_kernel(
    const uint4 *__restrict__ codes, const float *__restrict__ offsets, const QueryByValue qv, uint32_t qdim,
    uint32_t actual_dim, float alpha, float offset, int distance, int invert, uint8_t *__restrict__ qbuf, float multiplier,
    uint32_t n_rows, uint32_t row_chunks, uint32_t rows_per_wg, SmallTopk p) {
    __shared__ unsigned long long lds[2 * kSmallTopkWaves][64];
    __shared__ __attribute__((aligned(16))) uint32_t qc_lds[kFusedQueryDims / 4];
    __shared__ uint32_t qsum[2];
    const uint32_t t = threadIdx.x;
    const int lane = threadIdx.x & 63, sub = lane % G;
    if (t < 2) qsum[t] = 0;
    __syncthreads();
    const float placeholder = (distance == QAMD_DOT) ? 0.0f : offset;
    const uint32_t pad_code = f32_to_u8(placeholder, alpha, offset);
    uint32_t s1 = 0, s2 = 0;
    if (t < actual_dim / 4) {
        uint32_t packed = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const uint32_t j = 4 * t + b;
            const uint32_t c = j < qdim ? f32_to_u8(qv.v[j], alpha, offset) : pad_code;
            packed |= c << (8 * b);
            s1 += c;
            s2 += c * c;
        }
        qc_lds[t] = packed;
    }
    if (t < 256) {  // actual_dim / 4 <= 224: the first four waves hold everything
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
        }
        if (lane == 0) {
            atomicAdd(&qsum[0], s1);
            atomicAdd(&qsum[1], s2);
        }
    }
    __syncthreads();
    float q_off;
    if (distance == QAMD_DOT) q_off = (float)qsum[0] * alpha * offset;
    else if (distance == QAMD_L1) q_off = 0.0f;
    else q_off = (float)qsum[1] * alpha * alpha;
    if (invert) q_off = -q_off;
    if (blockIdx.x == 0) {  // the query object becomes an encoded query (layout: encode_query_kernel)
        if (t == 0) *reinterpret_cast<float *>(qbuf) = q_off;
        if (t < actual_dim / 4) reinterpret_cast<uint32_t *>(qbuf + 16)[t] = qc_lds[t];
    }
    uint4 q[ITERS];
    const uint4 *qc4 = reinterpret_cast<const uint4 *>(qc_lds);
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const uint32_t c = sub + it * G;
        const uint4 v = qc4[(EXACT || c < row_chunks) ? c : row_chunks - 1];
        const bool in = EXACT || c < row_chunks;
        q[it] = make_uint4(in ? v.x : 0, in ? v.y : 0, in ? v.z : 0, in ? v.w : 0);
    }
    u8_topk_small_body<G, ITERS, IS_L1, EXACT>(codes, offsets, q, q_off, multiplier, n_rows, row_chunks, rows_per_wg, p, lds);
}

// Generic dims (row_chunks > 16*8): runtime chunk loop, query re-read through L1/L2.
template <bool IS_L1>
__global__ __launch_bounds__(kBlock) void u8_scan_generic_kernel(
    const uint4 *__restrict__ codes, const float *__restrict__ offsets,
    const uint4 *__restrict__ qcodes, const float *__restrict__ q_off_p, float multiplier,
    uint32_t n_rows, uint32_t row_chunks, float *__restrict__ out) {
    constexpr int G = 16, RW = 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, rslot = lane / G;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * kBlock) >> 6;
    const float q_off = *q_off_p;
    for (uint64_t base = (uint64_t)wave * RW; base < n_rows; base += (uint64_t)n_waves * RW) {
        const uint64_t row = base + rslot;
        const uint4 *p = codes + row * row_chunks;
        uint32_t acc = 0;
        for (uint32_t c = sub; c < row_chunks; c += G) {
            uint4 v = ld_nt(p + c);
            uint4 qv = qcodes[c];
            acc = IS_L1 ? sad16(v, qv, acc) : dot16(v, qv, acc);
        }
        acc = group_sum<G>(acc);
        if (sub == 0 && row < n_rows)
            out[row] = epilogue(multiplier, acc, q_off, offsets[row], 0.0f, EPI_POINT);
    }
}

// avx2.c lane-exact variant (any dim): lane k of the reference's 8 x i32 accumulator gets
// byte pairs p with p % 8 == k of every 32-byte block (avx2.c:41-45) and bytes 2k,2k+1 of a
// 16-byte tail (:49-58).  Within a 16-byte chunk c (two per block) dword j holds pairs
// 2j, 2j+1 of that chunk, i.e. lanes (2j + 8*(c&1)... ) -> since a block has 16 pairs and
// lane = pair % 8, both chunks of a block map dword j to lanes 2j and 2j+1.  So 8 integer
// accumulators: acc[2j] += lo-half products of dword j, acc[2j+1] += hi-half products.
// Each is converted to f32 and summed ((l0+l4)+(l2+l6))+((l1+l5)+(l3+l7)) (HSUM256_PS).
template <bool DUMMY>
__global__ __launch_bounds__(kBlock) void u8_scan_avx2_lanes_kernel(
    const uint4 *__restrict__ codes, const float *__restrict__ offsets,
    const uint4 *__restrict__ qcodes, const float *__restrict__ q_off_p, float multiplier,
    uint32_t n_rows, uint32_t row_chunks, float *__restrict__ out) {
    constexpr int G = 16, RW = 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, rslot = lane / G;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * kBlock) >> 6;
    const float q_off = *q_off_p;
    for (uint64_t base = (uint64_t)wave * RW; base < n_rows; base += (uint64_t)n_waves * RW) {
        const uint64_t row = base + rslot;
        const uint4 *p = codes + row * row_chunks;
        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t c = sub; c < row_chunks; c += G) {
            uint4 v = ld_nt(p + c);
            uint4 qv = qcodes[c];
            const uint32_t vd[4] = {v.x, v.y, v.z, v.w};
            const uint32_t qd[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[2 * j] = __builtin_amdgcn_udot4(vd[j], qd[j] & 0x0000FFFFu, acc[2 * j], false);
                acc[2 * j + 1] = __builtin_amdgcn_udot4(vd[j], qd[j] & 0xFFFF0000u, acc[2 * j + 1], false);
            }
        }
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; k++) f[k] = (float)(int32_t)group_sum<G>(acc[k]);
        float a0 = f[0] + f[4], a1 = f[1] + f[5], a2 = f[2] + f[6], a3 = f[3] + f[7];
        float s = (a0 + a2) + (a1 + a3);
        if (sub == 0 && row < n_rows) out[row] = (multiplier * s + q_off) + offsets[row];
    }
}

// Random access: out[k] = score(query of pair k, ids[k]).  One 16-lane group per pair; ITERS =
// ceil(row_chunks / 16) is a template parameter so that all of a row's 16-byte pieces (and the query's)
// are in flight at once with no clamped duplicate loads (ITERS = 0: any row length, four-deep batches).
// Which query a pair is scored against:
//   lists == nullptr            one query for every pair (score_ids): q_single / q_off_single, which
//                               may point into the store itself (score_internal: "query" = row i);
//   lists, list_rows == nullptr query l of a batch for the pairs of list l (score_ids_batch);
//   lists, list_rows            stored row list_rows[l] for the pairs of list l (score_internal_ids_batch).
// A workgroup takes `pairs_per_block` consecutive pairs; with lists it finds the list of its first pair
// by ONE binary search (thread 0, through LDS) and every group then walks forward from there as its
// pair index grows -- a binary search per pair was ten dependent loads in front of every row fetch
// (1M random pairs: 0.26 ms; this form: profiles/r03_bursts.jsonl).
// Same integer sum and the same f32 epilogue as the scan => the same score bits.
template <bool IS_L1, int ITERS>
__global__ __launch_bounds__(kBlock) void u8_score_pairs_kernel(
    const uint4 *__restrict__ codes, const float *__restrict__ offsets, const uint4 *q_single, const float *q_off_single,
    const uint8_t *__restrict__ q_batch, uint32_t q_pitch, const float *__restrict__ q_offs,
    const uint32_t *__restrict__ lists, uint32_t n_lists, const uint32_t *__restrict__ list_rows, float multiplier,
    float diff, int mode, const uint32_t *__restrict__ ids, uint64_t n_ids, uint32_t n_rows, uint32_t row_chunks,
    uint32_t pairs_per_block, float *__restrict__ out) {
    constexpr int G = 16, GROUPS = kBlock / G;
    __shared__ uint32_t first_list;
    const int lane = threadIdx.x & 63;
    const int sub = lane % G, group = threadIdx.x / G;
    const uint64_t p0 = (uint64_t)blockIdx.x * pairs_per_block;
    const uint64_t p1 = p0 + pairs_per_block < n_ids ? p0 + pairs_per_block : n_ids;
    uint32_t l = lists ? first_list_of_block(lists, n_lists, p0, &first_list) : 0u;
    const float q_off_one = lists ? 0.0f : *q_off_single;
    for (uint64_t k = p0 + group; k < p1; k += GROUPS) {
        const uint32_t row = ids[k];
        bool ok = row < n_rows;
        const uint4 *qp = q_single;
        float q_off = q_off_one;
        if (lists) {
            l = advance_list(lists, n_lists, l, k);  // the pair index only grows: a step or two
            if (list_rows) {
                const uint32_t qr = list_rows[l];
                ok = ok && qr < n_rows;
                const uint32_t qc = qr < n_rows ? qr : 0u;
                qp = codes + (uint64_t)qc * row_chunks;
                q_off = offsets[qc];
            } else {
                qp = reinterpret_cast<const uint4 *>(q_batch + (size_t)l * q_pitch);
                q_off = q_offs[l];
            }
        }
        const uint4 *p = codes + (uint64_t)(ok ? row : 0u) * row_chunks;
        uint32_t acc = 0;
        if (ITERS > 0) {
            uint4 v[ITERS > 0 ? ITERS : 1], qv[ITERS > 0 ? ITERS : 1];
#pragma unroll
            for (int j = 0; j < ITERS; j++) {
                const uint32_t c = sub + j * G, cc = c < row_chunks ? c : row_chunks - 1;  // only the last piece can be clamped
                v[j] = p[cc];
                qv[j] = qp[cc];
            }
#pragma unroll
            for (int j = 0; j < ITERS; j++)
                if (j + 1 < ITERS || sub + j * G < row_chunks) acc = IS_L1 ? sad16(v[j], qv[j], acc) : dot16(v[j], qv[j], acc);
        } else {
            for (uint32_t c0 = sub; c0 < row_chunks; c0 += 4 * G) {
                uint4 v[4], qv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {  // unconditional loads (clamped address): all four issue back to back
                    const uint32_t c = c0 + j * G, cc = c < row_chunks ? c : row_chunks - 1;
                    v[j] = p[cc];
                    qv[j] = qp[cc];
                }
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (c0 + j * G < row_chunks) acc = IS_L1 ? sad16(v[j], qv[j], acc) : dot16(v[j], qv[j], acc);
            }
        }
        acc = group_sum<G>(acc);
        if (sub == 0) out[k] = ok ? epilogue(multiplier, acc, q_off, offsets[row], diff, mode) : __builtin_nanf("");
    }
}

// ------------------------------------------------------------------------------ encode kernels
// Pass 1: global min / max (quantile.rs:5-19).  `value < min` / `value > max` ignore NaN,
// as fminf/fmaxf do.  Per-block partials; the host folds them.
__global__ __launch_bounds__(kBlock) void minmax_kernel(const float *__restrict__ data, uint64_t n,
                                                       float *__restrict__ partial /* 2 per block */) {
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    const uint64_t n4 = ((reinterpret_cast<uintptr_t>(data) & 15) == 0) ? n / 4 : 0;
    const float4 *d4 = reinterpret_cast<const float4 *>(data);
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
        float4 v = ld_nt(d4 + i);
        mn = v.x < mn ? v.x : mn; mx = v.x > mx ? v.x : mx;
        mn = v.y < mn ? v.y : mn; mx = v.y > mx ? v.y : mx;
        mn = v.z < mn ? v.z : mn; mx = v.z > mx ? v.z : mx;
        mn = v.w < mn ? v.w : mn; mx = v.w > mx ? v.w : mx;
    }
    for (uint64_t i = n4 * 4 + (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float v = data[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        float o = __shfl_xor(mn, m, 64);
        mn = o < mn ? o : mn;
        o = __shfl_xor(mx, m, 64);
        mx = o > mx ? o : mx;
    }
    __shared__ float smn[kBlock / 64], smx[kBlock / 64];
    if ((threadIdx.x & 63) == 0) {
        smn[threadIdx.x >> 6] = mn;
        smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) {
            mn = smn[w] < mn ? smn[w] : mn;
            mx = smx[w] > mx ? smx[w] : mx;
        }
        partial[2 * blockIdx.x] = mn;
        partial[2 * blockIdx.x + 1] = mx;
    }
}

// Pass 1, streaming form: one 16 KiB tile per wave (non-persistent, nt loads — the shape that
// reads HBM fastest, see u8_scan_kernel), per-workgroup min/max folded into 64 sharded slots
// with integer atomics on order-preserving keys, so that a whole store needs ONE read-back.
// NaNs are skipped exactly like the reference's `value < min` / `value > max`.
__device__ __forceinline__ uint32_t f32_order_key(float f) {
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int kMinmaxSlots = 64, kMinmaxSlotStride = 64;  // one slot pair per 256-byte line
__global__ __launch_bounds__(kScanBlock) void minmax_stream_kernel(const float4 *__restrict__ d4, uint64_t n4,
                                                                  uint32_t *__restrict__ slots) {
    const uint64_t wave = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) >> 6;
    const uint64_t base = wave * 1024 + (threadIdx.x & 63);
    if (wave * 1024 >= n4) return;
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
    const float qnan = __uint_as_float(0x7FC00000u);
    float4 v[16];
    if (wave * 1024 + 1024 <= n4) {
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = ld_nt(d4 + base + j * 64);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t i = base + j * 64;
            v[j] = i < n4 ? ld_nt(d4 + i) : make_float4(qnan, qnan, qnan, qnan);  // NaN: skipped by both folds
        }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        mn = v[j].x < mn ? v[j].x : mn; mx = v[j].x > mx ? v[j].x : mx;
        mn = v[j].y < mn ? v[j].y : mn; mx = v[j].y > mx ? v[j].y : mx;
        mn = v[j].z < mn ? v[j].z : mn; mx = v[j].z > mx ? v[j].z : mx;
        mn = v[j].w < mn ? v[j].w : mn; mx = v[j].w > mx ? v[j].w : mx;
    }
    // wave fold: DPP inside each 16-lane row, then the four row results through readlane
    float o;
    o = dpp_f32<0xB1>(mn); mn = o < mn ? o : mn;   o = dpp_f32<0xB1>(mx); mx = o > mx ? o : mx;
    o = dpp_f32<0x4E>(mn); mn = o < mn ? o : mn;   o = dpp_f32<0x4E>(mx); mx = o > mx ? o : mx;
    o = dpp_f32<0x141>(mn); mn = o < mn ? o : mn;  o = dpp_f32<0x141>(mx); mx = o > mx ? o : mx;
    o = dpp_f32<0x140>(mn); mn = o < mn ? o : mn;  o = dpp_f32<0x140>(mx); mx = o > mx ? o : mx;
    float wmn = mn, wmx = mx;
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mn), r));
        const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mx), r));
        wmn = a < wmn ? a : wmn;
        wmx = c > wmx ? c : wmx;
    }
    if ((threadIdx.x & 63) == 0) {
        // The slot only ever moves outwards, so a stale read can cost a redundant atomic but never
        // lose an update; after the first few tiles almost no wave needs the atomic at all.
        uint32_t *slot = slots + (wave & (kMinmaxSlots - 1)) * kMinmaxSlotStride;
        const uint32_t kmn = f32_order_key(wmn), kmx = f32_order_key(wmx);
        if (kmn < __builtin_nontemporal_load(slot)) atomicMin(slot, kmn);
        if (kmx > __builtin_nontemporal_load(slot + 1)) atomicMax(slot + 1, kmx);
    }
}

// Pass 2, streaming form (dim % 4 == 0, 16-byte aligned input): 16 lanes per row, 4 rows per
// wave, one wave per tile.  Lane `sub` converts float4 #(sub + 16*it) of its row into one dword
// of codes: every wave-load is 4 x 256 contiguous bytes and up to four are in flight per lane;
// the row sums finish with DPP row adds.  Same arithmetic as quantize_kernel.
// Measured (2M x 768, profiles/r03_quantize_shapes.txt): 1.36-1.37 ms = 5.6 TB/s of read + written bytes with the
// exact division AND with the division-free conversion -- the kernel is bound by the mixed read/write stream, not
// by the vector ALU; plain instead of nt stores: the same; a lane owning four consecutive float4 (one 16-byte
// store per lane, loads at a 64-byte lane stride): 3.6 ms.
template <int ITERS /* float4 per lane, 0 = any dim (loop of 4-deep batches) */, bool FAST /* f32x4_to_u8x4_fast */>
__global__ __launch_bounds__(kScanBlock) void quantize16_kernel(
    const float *__restrict__ data, uint64_t n_rows, uint32_t dim, uint32_t actual_dim, float alpha,
    float offset, int distance, int invert, uint32_t *__restrict__ codes32, float *__restrict__ offsets,
    uint64_t row0) {
    const float r_alpha = 1.0f / alpha;
    const int lane = threadIdx.x & 63, sub = lane & 15, rslot = lane >> 4;
    const uint64_t wave = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) >> 6;
    if (wave * 4 >= n_rows) return;
    const uint64_t r = wave * 4 + rslot;
    const bool row_ok = r < n_rows;
    const uint64_t rc = row_ok ? r : n_rows - 1;  // clamped: loads stay in bounds, stores are masked
    const float *src = data + rc * dim;
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    const uint32_t dwords = actual_dim / 4, f4 = dim / 4;
    uint32_t *dst = codes32 + (row0 + rc) * dwords;
    const float placeholder = (distance == QAMD_DOT) ? 0.0f : offset;
    const uint32_t pad_code = f32_to_u8(placeholder, alpha, offset);
    const uint32_t pad_dword = pad_code * 0x01010101u;
    uint32_t s1 = 0, s2 = 0;
    constexpr int DEPTH = ITERS ? ITERS : 4;
    for (uint32_t d0 = sub; d0 < dwords; d0 += 16 * DEPTH) {
        float4 f[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const uint32_t d = d0 + 16 * j;
            f[j] = d < f4 ? ld_nt(src4 + d) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const uint32_t d = d0 + 16 * j;
            if (d >= dwords) break;
            uint32_t packed;
            if (d < f4) {
                if (FAST) {
                    packed = f32x4_to_u8x4_fast(f[j], alpha, offset, r_alpha);
                } else {
                    packed = f32_to_u8(f[j].x, alpha, offset) | (f32_to_u8(f[j].y, alpha, offset) << 8) |
                             (f32_to_u8(f[j].z, alpha, offset) << 16) | (f32_to_u8(f[j].w, alpha, offset) << 24);
                }
                s1 = __builtin_amdgcn_udot4(packed, 0x01010101u, s1, false);  // sum of the four codes
                s2 = __builtin_amdgcn_udot4(packed, packed, s2, false);       // sum of their squares
            } else {
                packed = pad_dword;
                s1 += 4 * pad_code;
                s2 += 4 * pad_code * pad_code;
            }
            if (row_ok) __builtin_nontemporal_store(packed, dst + d);  // 16 lanes x 4 B = one 64-byte segment
        }
        if (ITERS) break;
    }
    s1 = group_sum<16>(s1);
    s2 = group_sum<16>(s2);
    if (sub == 0 && row_ok) {
        float vo;
        const float D = (float)actual_dim;
        if (distance == QAMD_DOT) {
            float sum = (float)s1;
            if (s1 >= (1u << 24)) {
                sum = 0.0f;
                for (uint32_t j = 0; j < actual_dim; j++)
                    sum += (float)(j < dim ? f32_to_u8(src[j], alpha, offset) : pad_code);
            }
            vo = D * offset * offset + sum * alpha * offset;
        } else if (distance == QAMD_L1) {
            vo = 0.0f;
        } else {
            float sum = (float)s2;
            if (s2 >= (1u << 24)) {
                sum = 0.0f;
                for (uint32_t j = 0; j < actual_dim; j++) {
                    const float c = (float)(j < dim ? f32_to_u8(src[j], alpha, offset) : pad_code);
                    sum += c * c;
                }
            }
            vo = D * offset * offset + sum * alpha * alpha;
        }
        offsets[row0 + r] = invert ? -vo : vo;
    }
}

// Pass 2: one wave per row (encoded_vectors_u8.rs:73-118).  Lane l quantizes elements
// 4l..4l+3 of each 256-element slab into one dword of codes; the row's code sums are exact
// integers.  vector_offset is the reference's sequential f32 sum: identical to the integer
// sum while that is < 2^24; beyond it (only possible for sum of squares at actual_dim > 1040)
// lane 0 replays the sequential f32 loop.
__global__ __launch_bounds__(kBlock) void quantize_kernel(
    const float *__restrict__ data, uint64_t n_rows, uint32_t dim, uint32_t actual_dim, float alpha,
    float offset, int distance, int invert, uint32_t *__restrict__ codes32 /* row stride actual_dim/4 */,
    float *__restrict__ offsets, uint64_t row0) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint64_t n_waves = ((uint64_t)gridDim.x * kBlock) >> 6;
    const float placeholder = (distance == QAMD_DOT) ? 0.0f : offset;
    const uint32_t pad_code = f32_to_u8(placeholder, alpha, offset);
    const uint32_t dwords = actual_dim / 4;
    for (uint64_t r = wave; r < n_rows; r += n_waves) {
        const float *src = data + r * dim;
        uint32_t *dst = codes32 + (row0 + r) * dwords;
        uint32_t s1 = 0, s2 = 0;
        const bool vec4 = (dim % 4 == 0) && ((reinterpret_cast<uintptr_t>(data) & 15) == 0);
        for (uint32_t d = lane; d < dwords; d += 64) {
            uint32_t packed = 0;
            if (vec4 && d * 4 < dim) {  // whole dword inside the row: one aligned 16-byte load
                const float4 f = ld_nt(reinterpret_cast<const float4 *>(src) + d);
                const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t c = f32_to_u8(fv[b], alpha, offset);
                    packed |= c << (8 * b);
                    s1 += c;
                    s2 += c * c;
                }
            } else {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    uint32_t j = d * 4 + b;
                    uint32_t c = j < dim ? f32_to_u8(src[j], alpha, offset) : pad_code;
                    packed |= c << (8 * b);
                    s1 += c;
                    s2 += c * c;
                }
            }
            dst[d] = packed;
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
        }
        if (lane == 0) {
            float vo;
            const float D = (float)actual_dim;
            if (distance == QAMD_DOT) {
                float s = (float)s1;  // 127*actual_dim < 2^24 for every dim < 132k
                if (s1 >= (1u << 24)) {
                    s = 0.0f;
                    for (uint32_t j = 0; j < actual_dim; j++)
                        s += (float)(j < dim ? f32_to_u8(src[j], alpha, offset) : pad_code);
                }
                vo = D * offset * offset + s * alpha * offset;
            } else if (distance == QAMD_L1) {
                vo = 0.0f;
            } else {
                float s = (float)s2;
                if (s2 >= (1u << 24)) {
                    s = 0.0f;
                    for (uint32_t j = 0; j < actual_dim; j++) {
                        float c = (float)(j < dim ? f32_to_u8(src[j], alpha, offset) : pad_code);
                        s += c * c;
                    }
                }
                vo = D * offset * offset + s * alpha * alpha;
            }
            offsets[row0 + r] = invert ? -vo : vo;
        }
    }
}

// encode_query on device (encoded_vectors_u8.rs:290-329): one wave.
// qbuf: [0] offset f32, [16..] codes.
__global__ __launch_bounds__(64) void encode_query_kernel(const float *__restrict__ query,
                                                         uint32_t qdim, uint32_t actual_dim,
                                                         float alpha, float offset, int distance,
                                                         int invert, uint8_t *__restrict__ qbuf) {
    const int lane = threadIdx.x;
    const float placeholder = (distance == QAMD_DOT) ? 0.0f : offset;
    const uint32_t pad_code = f32_to_u8(placeholder, alpha, offset);
    uint8_t *codes = qbuf + 16;
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t j = lane; j < actual_dim; j += 64) {
        uint32_t c = j < qdim ? f32_to_u8(query[j], alpha, offset) : pad_code;
        codes[j] = (uint8_t)c;
        s1 += c;
        s2 += c * c;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if (lane == 0) {
        float off;
        if (distance == QAMD_DOT) {
            float s = (float)s1;
            if (s1 >= (1u << 24)) {
                s = 0.0f;
                for (uint32_t j = 0; j < actual_dim; j++)
                    s += (float)(j < qdim ? f32_to_u8(query[j], alpha, offset) : pad_code);
            }
            off = s * alpha * offset;
        } else if (distance == QAMD_L1) {
            off = 0.0f;
        } else {
            float s = (float)s2;
            if (s2 >= (1u << 24)) {
                s = 0.0f;
                for (uint32_t j = 0; j < actual_dim; j++) {
                    float c = (float)(j < qdim ? f32_to_u8(query[j], alpha, offset) : pad_code);
                    s += c * c;
                }
            }
            off = s * alpha * alpha;
        }
        *reinterpret_cast<float *>(qbuf) = invert ? -off : off;
    }
}

// Batch form of encode_query_kernel: one wave per query, separate code / offset arrays
// (the layout the multi-query MFMA path reads, u8_batch.hip).
__global__ __launch_bounds__(64) void encode_queries_kernel(const float *__restrict__ queries, uint32_t qdim,
                                                           uint32_t actual_dim, float alpha, float offset,
                                                           int distance, int invert, uint8_t *__restrict__ codes_out,
                                                           uint32_t code_pitch, float *__restrict__ offsets_out) {
    const int lane = threadIdx.x;
    const float *query = queries + (size_t)blockIdx.x * qdim;
    uint8_t *codes = codes_out + (size_t)blockIdx.x * code_pitch;
    const float placeholder = (distance == QAMD_DOT) ? 0.0f : offset;
    const uint32_t pad_code = f32_to_u8(placeholder, alpha, offset);
    uint32_t s1 = 0, s2 = 0;
    for (uint32_t j = lane; j < actual_dim; j += 64) {
        uint32_t c = j < qdim ? f32_to_u8(query[j], alpha, offset) : pad_code;
        codes[j] = (uint8_t)c;
        s1 += c;
        s2 += c * c;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if (lane == 0) {
        float off;
        if (distance == QAMD_DOT) {
            float s = (float)s1;
            if (s1 >= (1u << 24)) {
                s = 0.0f;
                for (uint32_t j = 0; j < actual_dim; j++)
                    s += (float)(j < qdim ? f32_to_u8(query[j], alpha, offset) : pad_code);
            }
            off = s * alpha * offset;
        } else if (distance == QAMD_L1) {
            off = 0.0f;
        } else {
            float s = (float)s2;
            if (s2 >= (1u << 24)) {
                s = 0.0f;
                for (uint32_t j = 0; j < actual_dim; j++) {
                    float c = (float)(j < qdim ? f32_to_u8(query[j], alpha, offset) : pad_code);
                    s += c * c;
                }
            }
            off = s * alpha * alpha;
        }
        offsets_out[blockIdx.x] = invert ? -off : off;
    }
}

// Reference-format rows <-> device layout (encoded_storage.rs:27-31, row stride actual_dim+4).
__global__ __launch_bounds__(kBlock) void split_rows_kernel(const uint32_t *__restrict__ rows32,
                                                           uint64_t n_rows, uint32_t row_dwords,
                                                           uint32_t *__restrict__ codes32,
                                                           float *__restrict__ offsets) {
    const uint64_t total = n_rows * row_dwords;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        uint64_t r = t / row_dwords;
        uint32_t k = (uint32_t)(t - r * row_dwords);
        uint32_t w = rows32[t];
        if (k == 0)
            offsets[r] = __uint_as_float(w);
        else
            codes32[r * (row_dwords - 1) + (k - 1)] = w;
    }
}

__global__ __launch_bounds__(kBlock) void join_rows_kernel(const uint32_t *__restrict__ codes32,
                                                          const float *__restrict__ offsets,
                                                          uint64_t n_rows, uint32_t row_dwords,
                                                          uint32_t *__restrict__ rows32) {
    const uint64_t total = n_rows * row_dwords;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += stride) {
        uint64_t r = t / row_dwords;
        uint32_t k = (uint32_t)(t - r * row_dwords);
        rows32[t] = k == 0 ? __float_as_uint(offsets[r]) : codes32[r * (row_dwords - 1) + (k - 1)];
    }
}

int grid_for(uint64_t work_items, uint64_t per_block, int blocks_per_cu) {
    uint64_t want = (work_items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)device_info().cu_count * blocks_per_cu;
    if (want < 1) want = 1;
    return (int)(want > cap ? cap : want);
}

// ------------------------------------------------------------------------------ host side
float host_multiplier(float alpha, int distance, int invert) {  // :119-128
    float m = distance == QAMD_DOT ? alpha * alpha : distance == QAMD_L1 ? alpha : -2.0f * alpha * alpha;
    return invert ? -m : m;
}

uint64_t actual_dim_of(uint64_t dim) { return dim + (16 - dim % 16) % 16; }  // :257-259

}  // namespace

// ------------------------------------------------------------------------------ handles
// struct qamd_u8 / qamd_u8_query: u8_internal.hpp

namespace {

qamd_status alloc_store(qamd_u8 *h) {
    h->padded_rows = round_up(h->count, kRowPad) + kRowPad;
    h->row_chunks = (uint32_t)(h->meta.actual_dim / 16);
    // every builder (encode, from_rows, load) writes all `count` rows; only the padding is zeroed
    QAMD_TRY(h->codes.alloc_zero_tail(h->padded_rows * h->meta.actual_dim, h->count * h->meta.actual_dim));
    QAMD_TRY(h->offsets.alloc_zero_tail(h->padded_rows * sizeof(float), h->count * sizeof(float)));
    return QAMD_OK;
}

template <bool IS_L1> struct ScanLaunch {
    template <int G, int ITERS, int UNROLL>
    static void go(const qamd_u8 *h, const uint4 *qc, const float *qo, float *out, const TopkFilter *filt,
                   hipStream_t s) {
        constexpr int TILE = (64 / G) * UNROLL;
        const uint64_t waves = (h->count + TILE - 1) / TILE;  // one wave per tile
        const unsigned grid = (unsigned)((waves + kScanBlock / 64 - 1) / (kScanBlock / 64));
        const bool exact = h->row_chunks == (uint32_t)(G * ITERS);
#define QAMD_U8_GO(EX, FI)                                                                                  \
    hipLaunchKernelGGL((u8_scan_kernel<G, ITERS, UNROLL, IS_L1, EX, FI>), dim3(grid), dim3(kScanBlock), 0, s, \
                       h->codes.as<uint4>(), h->offsets.as<float>(), qc, qo, h->meta.multiplier,            \
                       (uint32_t)h->count, h->row_chunks, out, filt ? *filt : TopkFilter{})
        if (filt) {
            if (exact) QAMD_U8_GO(true, true);
            else QAMD_U8_GO(false, true);
        } else {
            if (exact) QAMD_U8_GO(true, false);
            else QAMD_U8_GO(false, false);
        }
#undef QAMD_U8_GO
    }
};

// Returns false when the store's row size has no templated kernel (caller uses the generic one).
template <bool IS_L1>
bool launch_scan(const qamd_u8 *h, const uint4 *qc, const float *qo, float *out, const TopkFilter *filt,
                 hipStream_t s) {
    using L = ScanLaunch<IS_L1>;
    const uint32_t rc = h->row_chunks;
    if (rc == 1) return L::template go<1, 1, 4>(h, qc, qo, out, filt, s), true;
    if (rc == 2) return L::template go<2, 1, 4>(h, qc, qo, out, filt, s), true;
    if (rc <= 4) return L::template go<4, 1, 8>(h, qc, qo, out, filt, s), true;
    if (rc <= 8) return L::template go<8, 1, 8>(h, qc, qo, out, filt, s), true;
    switch ((rc + 15) / 16) {
        case 1: return L::template go<16, 1, 8>(h, qc, qo, out, filt, s), true;
        case 2: return L::template go<16, 2, 4>(h, qc, qo, out, filt, s), true;
        case 3: return L::template go<16, 3, 4>(h, qc, qo, out, filt, s), true;
        case 4: return L::template go<16, 4, 2>(h, qc, qo, out, filt, s), true;
        case 5: return L::template go<16, 5, 2>(h, qc, qo, out, filt, s), true;
        case 6: return L::template go<16, 6, 2>(h, qc, qo, out, filt, s), true;
        case 7: return L::template go<16, 7, 2>(h, qc, qo, out, filt, s), true;
        case 8: return L::template go<16, 8, 2>(h, qc, qo, out, filt, s), true;
        default: break;
    }
    return false;
}

template <bool IS_L1>
void launch_scan_generic(const qamd_u8 *h, const uint4 *qc, const float *qo, float *out, hipStream_t s) {
    uint64_t waves = (h->count + 3) / 4;  // one wave per 4-row tile (non-persistent, see u8_scan_kernel)
    unsigned grid = (unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64));
    hipLaunchKernelGGL((u8_scan_generic_kernel<IS_L1>), dim3(grid), dim3(kBlock), 0, s,
                       h->codes.as<uint4>(), h->offsets.as<float>(), qc, qo, h->meta.multiplier,
                       (uint32_t)h->count, h->row_chunks, out);
}

template <bool IS_L1> struct SmallLaunch {
    template <int G, int ITERS>
    static qamd_status go(const qamd_u8 *h, const uint4 *qc, const float *qo, const FusedQuery *fq, const SmallTopkPlan &pl,
                          const SmallTopk &p, hipStream_t s) {
        const bool exact = h->row_chunks == (uint32_t)(G * ITERS);
        if constexpr (G * ITERS * 16 > (int)kFusedQueryDims + 15 * 16) {
            if (fq) return fail(QAMD_ERR_ARGUMENTS, "fused query does not fit the kernel arguments");
        } else if (fq) {
            QueryByValue qv;
            memcpy(qv.v, fq->values, (size_t)fq->qdim * 4);
            const qamd_vector_parameters &vp = h->meta.vector_parameters;
#define QAMD_U8_SMALL_FUSED(EX)                                                                                        \
    hipLaunchKernelGGL((u8_topk_small_fused_kernel<G, ITERS, IS_L1, EX>), dim3(pl.workgroups), dim3(1024), 0, s,          \
                       h->codes.as<uint4>(), h->offsets.as<float>(), qv, fq->qdim, (uint32_t)h->meta.actual_dim,         \
                       h->meta.alpha, h->meta.offset, vp.distance_type, vp.invert, fq->qbuf, h->meta.multiplier,           \
                       (uint32_t)h->count, h->row_chunks, pl.rows_per_wg, p)
            if (exact) QAMD_U8_SMALL_FUSED(true);
            else QAMD_U8_SMALL_FUSED(false);
#undef QAMD_U8_SMALL_FUSED
            QAMD_HIP(hipGetLastError());
            return QAMD_OK;
        }
#define QAMD_U8_SMALL(EX)                                                                                       \
    hipLaunchKernelGGL((u8_topk_small_kernel<G, ITERS, IS_L1, EX>), dim3(pl.workgroups), dim3(1024), 0, s,       \
                       h->codes.as<uint4>(), h->offsets.as<float>(), qc, qo, h->meta.multiplier, (uint32_t)h->count, \
                       h->row_chunks, pl.rows_per_wg, p)
        if (exact) QAMD_U8_SMALL(true);
        else QAMD_U8_SMALL(false);
#undef QAMD_U8_SMALL
        QAMD_HIP(hipGetLastError());
        return QAMD_OK;
    }
};

// Row group of the small top-k kernel for this store's row size; 0 = no templated shape.
int small_group(uint32_t rc) { return rc == 1 ? 1 : rc == 2 ? 2 : rc <= 4 ? 4 : rc <= 8 ? 8 : rc <= 128 ? 16 : 0; }

template <bool IS_L1>
qamd_status launch_small(const qamd_u8 *h, const uint4 *qc, const float *qo, const FusedQuery *fq, const SmallTopkPlan &pl,
                         const SmallTopk &p, hipStream_t s) {
    using L = SmallLaunch<IS_L1>;
    const uint32_t rc = h->row_chunks;
    if (rc == 1) return L::template go<1, 1>(h, qc, qo, fq, pl, p, s);
    if (rc == 2) return L::template go<2, 1>(h, qc, qo, fq, pl, p, s);
    if (rc <= 4) return L::template go<4, 1>(h, qc, qo, fq, pl, p, s);
    if (rc <= 8) return L::template go<8, 1>(h, qc, qo, fq, pl, p, s);
    switch ((rc + 15) / 16) {
        case 1: return L::template go<16, 1>(h, qc, qo, fq, pl, p, s);
        case 2: return L::template go<16, 2>(h, qc, qo, fq, pl, p, s);
        case 3: return L::template go<16, 3>(h, qc, qo, fq, pl, p, s);
        case 4: return L::template go<16, 4>(h, qc, qo, fq, pl, p, s);
        case 5: return L::template go<16, 5>(h, qc, qo, fq, pl, p, s);
        case 6: return L::template go<16, 6>(h, qc, qo, fq, pl, p, s);
        case 7: return L::template go<16, 7>(h, qc, qo, fq, pl, p, s);
        case 8: return L::template go<16, 8>(h, qc, qo, fq, pl, p, s);
        default: break;
    }
    return fail(QAMD_ERR_ARGUMENTS, "no small top-k kernel for %u chunks", rc);
}

// ---- several queries per pass (u8_scan_multi_kernel) -------------------------------------------
template <bool IS_L1, int G, int ITERS, int UNROLL, int NQ>
void launch_multi_shape(const qamd_u8 *h, const uint8_t *qcodes, uint64_t q_pitch, const float *q_offs, uint32_t nq_valid,
                        float *out, const TopkFilterSlices *slices, hipStream_t s) {
    constexpr int TILE = (64 / G) * UNROLL * NQ;  // NQ tiles per wave (u8_scan_multi_kernel TPW)
    const uint64_t waves = (h->count + TILE - 1) / TILE;
    const unsigned grid = (unsigned)((waves + kScanBlock / 64 - 1) / (kScanBlock / 64));
    const bool exact = h->row_chunks == (uint32_t)(G * ITERS);
#define QAMD_U8_MULTI(EX, FI)                                                                                      \
    hipLaunchKernelGGL((u8_scan_multi_kernel<G, ITERS, UNROLL, NQ, IS_L1, EX, FI>), dim3(grid), dim3(kScanBlock), 0, s, \
                       h->codes.as<uint4>(), h->offsets.as<float>(), qcodes, (uint32_t)q_pitch, q_offs, nq_valid,   \
                       h->meta.multiplier, (uint32_t)h->count, h->row_chunks, out, (uint64_t)h->count,             \
                       slices ? *slices : TopkFilterSlices{})
    if (slices) {
        if (exact) QAMD_U8_MULTI(true, true);
        else QAMD_U8_MULTI(false, true);
    } else {
        if (exact) QAMD_U8_MULTI(true, false);
        else QAMD_U8_MULTI(false, false);
    }
#undef QAMD_U8_MULTI
}

// Widest pass of the multi-query scan for this row size: 4 queries (2 for rows of more than 96
// pieces), 0 = none.  Measured on 10M x 768 / 12.5M x 1536, top-30 per query, whole call: 2 queries
// 1.24 / 2.93 ms and 4 queries 1.38 / 3.50 ms against 1.65 / 3.85 ms on the matrix-core path; an
// 8-wide pass (96 query VGPRs, 3 waves per SIMD) took 1.97 ms at dim 768 -- worse than the matrix
// cores -- so batches of 5 and more stay there.  L1 has no matrix form: its batches are served by this kernel alone,
// and there the 8-wide pass is the better one (8 queries per 1.97 ms against two 4-wide passes of 1.38 ms; rows of up
// to 768 bytes: 96 query registers).
uint32_t multi_width(const qamd_u8 *h) {
    const uint32_t rc = h->row_chunks, iters = (rc + 15) / 16;
    const bool l1 = h->meta.vector_parameters.distance_type == QAMD_L1;
    if (rc < 9 || iters > 8 || (h->lane_mode != 0 && !l1)) return 0;
    return l1 && iters <= 3 ? 8 : iters <= 6 ? 4 : 2;
}

// One pass for `nq_valid` (2 .. multi_width) queries; false = unsupported.
template <bool IS_L1>
bool launch_multi(const qamd_u8 *h, uint32_t nq_valid, const uint8_t *qcodes, uint64_t q_pitch, const float *q_offs,
                  float *out, const TopkFilterSlices *slices, hipStream_t s) {
    const uint32_t iters = (h->row_chunks + 15) / 16, width = multi_width(h);
    if (nq_valid < 2 || nq_valid > width) return false;
#define QAMD_U8_MULTI_IT(IT, UN)                                                                              \
    case IT:                                                                                                  \
        if constexpr (IS_L1 && IT <= 3) {                                                                     \
            if (nq_valid > 4) {                                                                               \
                launch_multi_shape<IS_L1, 16, IT, UN, 8>(h, qcodes, q_pitch, q_offs, nq_valid, out, slices, s); \
                return true;                                                                                  \
            }                                                                                                 \
        }                                                                                                     \
        if (nq_valid > 2) launch_multi_shape<IS_L1, 16, IT, UN, (IT <= 6 ? 4 : 2)>(h, qcodes, q_pitch, q_offs, nq_valid, out, slices, s); \
        else launch_multi_shape<IS_L1, 16, IT, UN, 2>(h, qcodes, q_pitch, q_offs, nq_valid, out, slices, s);  \
        return true;
    switch (iters) {
        QAMD_U8_MULTI_IT(1, 4) QAMD_U8_MULTI_IT(2, 2) QAMD_U8_MULTI_IT(3, 2) QAMD_U8_MULTI_IT(4, 2)
        QAMD_U8_MULTI_IT(5, 1) QAMD_U8_MULTI_IT(6, 1) QAMD_U8_MULTI_IT(7, 1) QAMD_U8_MULTI_IT(8, 1)
        default: break;
    }
#undef QAMD_U8_MULTI_IT
    return false;
}

bool fused_capable(const qamd_u8 *h) {
    const bool is_l1 = h->meta.vector_parameters.distance_type == QAMD_L1;
    return (is_l1 || h->lane_mode == 0) && h->row_chunks <= 128;
}

// The single-launch top-k serves this store and k (and, for a fused query, these dims): its plan.
bool u8_small_plan(const qamd_u8 *h, uint32_t k, SmallTopkPlan &plan) {
    const int g = small_group(h->row_chunks);
    if (!g || !fused_capable(h)) return false;
    const uint32_t tile = 2 * (64 / g);
    const uint32_t least = (uint32_t)std::max<uint64_t>(16 * tile, (128 * 1024) / std::max<uint64_t>(h->meta.actual_dim, 1));
    return small_topk_plan(h->count, k, tile, least, plan);
}

// qc: the query's actual_dim codes (16-byte aligned), qo: its f32 offset -- inside a qamd_u8_query
// or straight out of a query batch ([q][pitch] codes, [q] offsets).
qamd_status scan_ptrs(const qamd_u8 *h, const uint4 *qc, const float *qo, float *out_dev, hipStream_t s,
                      const TopkFilter *filt = nullptr) {
    if (h->count == 0) return QAMD_OK;
    const bool is_l1 = h->meta.vector_parameters.distance_type == QAMD_L1;
    if (!is_l1 && h->lane_mode == 1) {
        uint64_t waves = (h->count + 3) / 4;
        unsigned grid = (unsigned)((waves + kBlock / 64 - 1) / (kBlock / 64));
        hipLaunchKernelGGL((u8_scan_avx2_lanes_kernel<true>), dim3(grid), dim3(kBlock), 0, s,
                           h->codes.as<uint4>(), h->offsets.as<float>(), qc, qo, h->meta.multiplier,
                           (uint32_t)h->count, h->row_chunks, out_dev);
    } else if (is_l1) {
        if (!launch_scan<true>(h, qc, qo, out_dev, filt, s)) launch_scan_generic<true>(h, qc, qo, out_dev, s);
    } else {
        if (!launch_scan<false>(h, qc, qo, out_dev, filt, s)) launch_scan_generic<false>(h, qc, qo, out_dev, s);
    }
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

qamd_status scan_into(const qamd_u8 *h, const qamd_u8_query *q, float *out_dev, hipStream_t s,
                      const TopkFilter *filt = nullptr) {
    return scan_ptrs(h, reinterpret_cast<const uint4 *>(q->buf.as<uint8_t>() + 16), q->buf.as<float>(), out_dev, s, filt);
}

qamd_status check_query(const qamd_u8 *h, const qamd_u8_query *q) {
    if (!h || !q) return fail(QAMD_ERR_ARGUMENTS, "null handle or query");
    if (q->actual_dim != h->meta.actual_dim)
        return fail(QAMD_ERR_ARGUMENTS, "query has %llu codes, store rows have %llu",
                    (unsigned long long)q->actual_dim, (unsigned long long)h->meta.actual_dim);
    return QAMD_OK;
}

// One launch of u8_score_pairs_kernel.  lists == nullptr: the single query (qc, qo) for every id.
template <bool IS_L1>
qamd_status launch_pairs(const qamd_u8 *h, const uint4 *qc, const float *qo, const uint8_t *q_batch, uint32_t q_pitch,
                         const float *q_offs, const uint32_t *lists, uint32_t n_lists, const uint32_t *list_rows, float diff,
                         int mode, const uint32_t *ids_dev, uint64_t n_ids, float *out_dev, hipStream_t s) {
    const uint64_t ppb = pairs_per_block(n_ids, 16);  // 16 lane groups per workgroup
    const unsigned grid = (unsigned)((n_ids + ppb - 1) / ppb);
    const uint32_t iters = (h->row_chunks + 15) / 16;
#define QAMD_U8_PAIRS(IT)                                                                                                \
    hipLaunchKernelGGL((u8_score_pairs_kernel<IS_L1, IT>), dim3(grid), dim3(kBlock), 0, s, h->codes.as<uint4>(),         \
                       h->offsets.as<float>(), qc, qo, q_batch, q_pitch, q_offs, lists, n_lists, list_rows,              \
                       h->meta.multiplier, diff, mode, ids_dev, n_ids, (uint32_t)h->count, h->row_chunks, (uint32_t)ppb, \
                       out_dev)
    switch (iters) {
        case 1: QAMD_U8_PAIRS(1); break;
        case 2: QAMD_U8_PAIRS(2); break;
        case 3: QAMD_U8_PAIRS(3); break;
        case 4: QAMD_U8_PAIRS(4); break;
        case 5: QAMD_U8_PAIRS(5); break;
        case 6: QAMD_U8_PAIRS(6); break;
        case 7: QAMD_U8_PAIRS(7); break;
        case 8: QAMD_U8_PAIRS(8); break;
        default: QAMD_U8_PAIRS(0); break;
    }
#undef QAMD_U8_PAIRS
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

qamd_status score_pairs_dev(const qamd_u8 *h, const uint4 *qc, const float *qo, const uint8_t *q_batch, uint32_t q_pitch,
                            const float *q_offs, const uint32_t *lists, uint32_t n_lists, const uint32_t *list_rows,
                            float diff, int mode, const uint32_t *ids_dev, uint64_t n_ids, float *out_dev, hipStream_t s) {
    if (n_ids == 0) return QAMD_OK;
    if (h->meta.vector_parameters.distance_type == QAMD_L1)
        return launch_pairs<true>(h, qc, qo, q_batch, q_pitch, q_offs, lists, n_lists, list_rows, diff, mode, ids_dev, n_ids, out_dev, s);
    return launch_pairs<false>(h, qc, qo, q_batch, q_pitch, q_offs, lists, n_lists, list_rows, diff, mode, ids_dev, n_ids, out_dev, s);
}

qamd_status score_ids_dev(const qamd_u8 *h, const uint4 *qc, const float *qo, float diff, int mode,
                          const uint32_t *ids_dev, uint64_t n_ids, float *out_dev, hipStream_t s) {
    return score_pairs_dev(h, qc, qo, nullptr, 0, nullptr, nullptr, 0, nullptr, diff, mode, ids_dev, n_ids, out_dev, s);
}

// encoded_vectors_u8.rs:389-395: the `diff` of score_internal, actual_dim * offset * offset (negated if invert)
float internal_diff(const qamd_u8 *h) {
    float diff = (float)h->meta.actual_dim * h->meta.offset * h->meta.offset;
    return h->meta.vector_parameters.invert ? -diff : diff;
}

// out[k] = score of (query (qc, qo), ids[k]) for host or device ids / outputs: the body of score_ids
// and, with (qc, qo) = row i of the store and the internal epilogue, of score_internal_ids.
qamd_status score_ids_any(const qamd_u8 *h, const uint4 *qc, const float *qo, float diff, int mode, const uint32_t *ids,
                          uint64_t n_ids, qamd_mem ids_mem, float *out, qamd_mem out_mem, hipStream_t s) {
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
        QAMD_TRY(score_ids_dev(h, qc, qo, diff, mode, hs.dev, n_ids, reinterpret_cast<float *>(hs.dev + 1024), s));
        QAMD_HIP(hipStreamSynchronize(s));
        memcpy(out, hs.host + 1024, n_ids * 4);
        return QAMD_OK;
    }
    if (ids_mem == QAMD_MEM_HOST) {
        for (uint64_t k = 0; k < n_ids; k++)
            if (ids[k] >= h->count)  // the reference panics here (encoded_storage.rs:29)
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
    QAMD_TRY(score_ids_dev(h, qc, qo, diff, mode, ids_dev, n_ids, out_dev, s));
    if (out_mem == QAMD_MEM_HOST) QAMD_TRY(copy_out(out, QAMD_MEM_HOST, out_dev, n_ids * 4, s));
    else if (ids_mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));
    return QAMD_OK;
}

// Gather `n_out` evenly strided rows of a device-resident [count][dim] array (quantile sample).
__global__ __launch_bounds__(kBlock) void gather_strided_rows_kernel(const float *__restrict__ data, uint64_t count,
                                                                    uint32_t dim, uint64_t n_out,
                                                                    float *__restrict__ out) {
    const uint64_t total = n_out * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t k = i / dim, j = i - k * dim;
        const uint64_t r = (uint64_t)((unsigned __int128)k * count / n_out);
        out[i] = data[r * dim + j];
    }
}

bool quantile_cut(uint64_t slice, uint64_t dim, float quantile, uint64_t &cut);

// find_quantile_interval (quantile.rs:21-71).  The reference keeps, after two
// select_nth_unstable calls, the values of sorted rank (cut, len - cut) exclusive and returns
// their min and max: sorted[cut + 1] and sorted[len - cut - 1], i.e. the (cut+2)-th smallest
// and the (cut+1)-th largest value — two exact radix selects on the GPU.
// count <= 100 000: the sample is every vector (exactly the reference).  Larger stores: an
// evenly strided subset (the reference draws a random one; statistic, not bits).
qamd_status quantile_interval_device(const float *data, qamd_mem data_mem, uint64_t count, uint64_t dim,
                                     float quantile, hipStream_t s, bool &found, float &mn, float &mx) {
    found = false;
    const uint64_t slice = std::min<uint64_t>(count, kQuantileSample);
    const uint64_t len = slice * dim;
    uint64_t cut = 0;
    if (!quantile_cut(slice, dim, quantile, cut)) return QAMD_OK;
    DevBuf sample;
    const float *vals = data;
    if (data_mem == QAMD_MEM_HOST) {
        QAMD_TRY(sample.alloc(len * 4));
        if (slice == count) {
            QAMD_TRY(copy_in(sample.ptr, data, QAMD_MEM_HOST, len * 4, s));
        } else {
            std::vector<float> rows(len);  // row gather only: no arithmetic on the host
            for (uint64_t k = 0; k < slice; k++)
                memcpy(&rows[k * dim], data + (uint64_t)((unsigned __int128)k * count / slice) * dim, dim * 4);
            QAMD_TRY(copy_in(sample.ptr, rows.data(), QAMD_MEM_HOST, len * 4, s));
        }
        vals = sample.as<float>();
    } else if (slice != count) {
        QAMD_TRY(sample.alloc(len * 4));
        hipLaunchKernelGGL(gather_strided_rows_kernel, dim3(grid_for(len, kBlock * 4, 8)), dim3(kBlock), 0, s, data,
                           count, (uint32_t)dim, slice, sample.as<float>());
        QAMD_HIP(hipGetLastError());
        vals = sample.as<float>();
    }
    QAMD_TRY(select_kth_f32(vals, len, cut + 2, false, &mn, s));
    QAMD_TRY(select_kth_f32(vals, len, cut + 1, true, &mx, s));
    found = true;
    return QAMD_OK;
}


// PASS 1 accumulator (quantile.rs:5-19): global min / max over any number of device-resident
// batches, kept on the device (64 sharded slot pairs) until result() reads it back once.
struct MinMaxAcc {
    DevBuf slots, partial;
    std::vector<float> hp;
    int mm_grid = 0;
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;  // folded from the scalar-form launches

    qamd_status init(hipStream_t s) {
        QAMD_TRY(slots.alloc(kMinmaxSlots * kMinmaxSlotStride * sizeof(uint32_t)));
        std::vector<uint32_t> init(kMinmaxSlots * kMinmaxSlotStride, 0u);
        for (int b = 0; b < kMinmaxSlots; b++) {
            init[b * kMinmaxSlotStride] = 0xFFFFFFFFu;  // min slot: largest key
            init[b * kMinmaxSlotStride + 1] = 0u;       // max slot: smallest key
        }
        QAMD_TRY(copy_in(slots.ptr, init.data(), QAMD_MEM_HOST, init.size() * 4, s));
        mm_grid = device_info().cu_count * 4;
        QAMD_TRY(partial.alloc((size_t)mm_grid * 2 * sizeof(float)));
        hp.resize((size_t)mm_grid * 2);
        return QAMD_OK;
    }

    // src: device memory, nvals f32.  Synchronises only for the scalar-form (unaligned / tail) launches.
    qamd_status feed(const float *src, uint64_t nvals, hipStream_t s) {
        if (nvals == 0) return QAMD_OK;
        if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && nvals >= 4) {
            const uint64_t n4 = nvals / 4;
            const unsigned grid = (unsigned)((n4 + 1024 * (kScanBlock / 64) - 1) / (1024 * (kScanBlock / 64)));
            hipLaunchKernelGGL(minmax_stream_kernel, dim3(grid), dim3(kScanBlock), 0, s,
                               reinterpret_cast<const float4 *>(src), n4, slots.as<uint32_t>());
            if (nvals % 4) {  // the 1..3 trailing values
                hipLaunchKernelGGL(minmax_kernel, dim3(1), dim3(kBlock), 0, s, src + n4 * 4, nvals % 4,
                                   partial.as<float>());
                QAMD_TRY(copy_out(hp.data(), QAMD_MEM_HOST, partial.ptr, 8, s));
                if (hp[0] < mn) mn = hp[0];
                if (hp[1] > mx) mx = hp[1];
            }
        } else {  // unaligned input: grid-stride scalar form
            hipLaunchKernelGGL(minmax_kernel, dim3(mm_grid), dim3(kBlock), 0, s, src, nvals, partial.as<float>());
            QAMD_TRY(copy_out(hp.data(), QAMD_MEM_HOST, partial.ptr, hp.size() * 4, s));
            for (int b = 0; b < mm_grid; b++) {
                if (hp[2 * b] < mn) mn = hp[2 * b];
                if (hp[2 * b + 1] > mx) mx = hp[2 * b + 1];
            }
        }
        QAMD_HIP(hipGetLastError());
        return QAMD_OK;
    }

    qamd_status result(hipStream_t s, float &out_mn, float &out_mx) {
        std::vector<uint32_t> all_slots(kMinmaxSlots * kMinmaxSlotStride);
        QAMD_TRY(copy_out(all_slots.data(), QAMD_MEM_HOST, slots.ptr, all_slots.size() * 4, s));
        auto key_to_f32 = [](uint32_t k) {
            uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
            float f;
            memcpy(&f, &u, 4);
            return f;
        };
        for (int b = 0; b < kMinmaxSlots; b++) {
            const uint32_t kmn = all_slots[b * kMinmaxSlotStride], kmx = all_slots[b * kMinmaxSlotStride + 1];
            if (kmn != 0xFFFFFFFFu) {
                const float v = key_to_f32(kmn);
                if (v < mn) mn = v;
            }
            if (kmx != 0u) {
                const float v = key_to_f32(kmx);
                if (v > mx) mx = v;
            }
        }
        out_mn = mn;
        out_mx = mx;
        return QAMD_OK;
    }
};

// PASS 2 for `nr` device-resident rows (encoded_vectors_u8.rs:73-118): rows r0 .. r0+nr of the store.
qamd_status launch_quantize(qamd_u8 *h, const float *src, uint64_t nr, uint64_t r0, float alpha, float offset,
                            hipStream_t s) {
    if (nr == 0) return QAMD_OK;
    const qamd_vector_parameters &vp = h->meta.vector_parameters;
    const uint64_t dim = vp.dim;
    if (dim % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint64_t waves = (nr + 3) / 4;
        const unsigned grid = (unsigned)((waves + kScanBlock / 64 - 1) / (kScanBlock / 64));
        const uint32_t per_lane = (uint32_t)((h->meta.actual_dim / 4 + 15) / 16);  // float4 per lane
        // the division-free conversion needs a normal alpha well inside the exponent range (f32_to_u8_fast)
        const bool fast = alpha >= 8.673617379884035e-19f && alpha <= 1.152921504606847e+18f;  // 2^-60 .. 2^60
#define QAMD_Q16(IT)                                                                                          \
    do {                                                                                                      \
        if (fast)                                                                                             \
            hipLaunchKernelGGL((quantize16_kernel<IT, true>), dim3(grid), dim3(kScanBlock), 0, s, src, nr,    \
                               (uint32_t)dim, (uint32_t)h->meta.actual_dim, alpha, offset, vp.distance_type,  \
                               vp.invert, h->codes.as<uint32_t>(), h->offsets.as<float>(), r0);               \
        else                                                                                                  \
            hipLaunchKernelGGL((quantize16_kernel<IT, false>), dim3(grid), dim3(kScanBlock), 0, s, src, nr,   \
                               (uint32_t)dim, (uint32_t)h->meta.actual_dim, alpha, offset, vp.distance_type,  \
                               vp.invert, h->codes.as<uint32_t>(), h->offsets.as<float>(), r0);               \
    } while (0)
        if (per_lane <= 2) QAMD_Q16(2);
        else if (per_lane <= 4) QAMD_Q16(4);
        else if (per_lane <= 6) QAMD_Q16(6);
        else if (per_lane <= 8) QAMD_Q16(8);
        else if (per_lane <= 12) QAMD_Q16(12);
        else if (per_lane <= 16) QAMD_Q16(16);
        else QAMD_Q16(0);
#undef QAMD_Q16
    } else {
        int grid = grid_for(nr, kBlock / 64, 8);
        hipLaunchKernelGGL(quantize_kernel, dim3(grid), dim3(kBlock), 0, s, src, nr, (uint32_t)dim,
                           (uint32_t)h->meta.actual_dim, alpha, offset, vp.distance_type, vp.invert,
                           h->codes.as<uint32_t>(), h->offsets.as<float>(), r0);
    }
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// quantile.rs:52-61 bounds shared by the one-shot and the streaming encoder: false = the
// reference keeps the plain min/max interval.
bool quantile_cut(uint64_t slice, uint64_t dim, float quantile, uint64_t &cut) {
    const uint64_t len = slice * dim;
    if (len < 4) return false;  // :48-50
    cut = std::min<uint64_t>((len - 1) / 2, (uint64_t)((float)slice * (1.0f - quantile) / 2.0f));  // :52-55
    cut = std::max<uint64_t>(cut, 1);
    const uint64_t lo = cut + 1, hi = len - cut;
    return !(hi <= lo || hi - lo < 2);  // :63-65
}

}  // namespace

// ================================================================================== C ABI
extern "C" {

uint64_t qamd_u8_actual_dim(const qamd_vector_parameters *vp) { return actual_dim_of(vp->dim); }

uint64_t qamd_u8_quantized_vector_size(const qamd_vector_parameters *vp) {
    return actual_dim_of(vp->dim) + sizeof(float);
}

qamd_status qamd_u8_encode(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                           const float *quantile, const float *alpha_offset, qamd_stop_fn stop,
                           void *stop_user, void *stream, qamd_u8 **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (vp->distance_type < 0 || vp->distance_type > 2)
        return fail(QAMD_ERR_ARGUMENTS, "bad distance_type %d", vp->distance_type);
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    if (vp->count > 0 && vp->dim > 0 && !data) return fail(QAMD_ERR_ARGUMENTS, "data is null");
    QAMD_ON_DEVICE(current_device());
    hipStream_t s = as_stream(stream);
    std::unique_ptr<qamd_u8> h(new qamd_u8);
    h->device = current_device();
    h->count = vp->count;
    h->meta.actual_dim = actual_dim_of(vp->dim);
    h->meta.vector_parameters = *vp;
    QAMD_TRY(alloc_store(h.get()));
    if (vp->count == 0) {  // encoded_vectors_u8.rs:43-54
        h->meta.alpha = h->meta.offset = h->meta.multiplier = 0.0f;
        if (alpha_offset) {  // an empty SHARD of a store whose interval was found elsewhere keeps that store's metadata
            h->meta.alpha = alpha_offset[0];
            h->meta.offset = alpha_offset[1];
            h->meta.multiplier = host_multiplier(alpha_offset[0], vp->distance_type, vp->invert);
        }
        *out = h.release();
        return QAMD_OK;
    }
    const uint64_t dim = vp->dim, count = vp->count;

    // Host rows: both passes (min/max, quantize) need every value, so when the f32 data fits in
    // a quarter of the free HBM it is uploaded ONCE into a scratch copy and encoded from there
    // (the PCIe transfer is the whole cost of a host-side encode); larger inputs are staged
    // twice, 256 MiB at a time.  With alpha_offset given there is only one pass anyway.
    StreamBuf whole;  // from the stream-ordered pool: a fresh hipMalloc of 6 GB costs as much as its upload
    if (data_mem == QAMD_MEM_HOST && !alpha_offset) {
        size_t free_b = 0, total_b = 0;
        const uint64_t bytes = count * dim * sizeof(float);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes <= free_b / 4) {
            QAMD_TRY(whole.alloc(bytes, s));
            const uint64_t piece = 256ull << 20;
            for (uint64_t off = 0; off < bytes; off += piece) {
                if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");
                QAMD_TRY(copy_in(static_cast<char *>(whole.ptr) + off, reinterpret_cast<const char *>(data) + off,
                                 QAMD_MEM_HOST, std::min<uint64_t>(piece, bytes - off), s));
            }
            data = whole.as<float>();
            data_mem = QAMD_MEM_DEVICE;
        }
    }

    // Source rows: device-resident rows are processed in place (large batches: the stop
    // callback is polled between them), host rows are staged in bounded 256 MiB batches.
    const uint64_t batch_bytes = data_mem == QAMD_MEM_HOST ? (256ull << 20) : (8ull << 30);
    const uint64_t batch_rows = std::max<uint64_t>(1, std::min<uint64_t>(count, batch_bytes / (dim * 4 + 1)));
    DevBuf stage;
    if (data_mem == QAMD_MEM_HOST) QAMD_TRY(stage.alloc(batch_rows * dim * sizeof(float)));
    auto batch_src = [&](uint64_t r0, uint64_t nr, const float *&src) -> qamd_status {
        src = data + r0 * dim;
        if (data_mem == QAMD_MEM_HOST) {
            QAMD_TRY(copy_in(stage.ptr, src, QAMD_MEM_HOST, nr * dim * 4, s));
            src = stage.as<float>();
        }
        return QAMD_OK;
    };

    float alpha, offset;
    if (alpha_offset) {
        alpha = alpha_offset[0];
        offset = alpha_offset[1];
    } else {
        // PASS 1 (:57): global min/max, accumulated on the device across batches.
        MinMaxAcc acc;
        QAMD_TRY(acc.init(s));
        for (uint64_t r0 = 0; r0 < count; r0 += batch_rows) {
            if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");
            const uint64_t nr = std::min(batch_rows, count - r0);
            const float *src = nullptr;
            QAMD_TRY(batch_src(r0, nr, src));
            QAMD_TRY(acc.feed(src, nr * dim, s));
            if (data_mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));  // staging buffer is reused
        }
        float mn, mx;
        QAMD_TRY(acc.result(s, mn, mx));
        alpha = (mx - mn) / 127.0f;  // :228-232
        offset = mn;
        // PASS 1b (:58-71): quantile interval on <= 100 000 sampled vectors.
        if (quantile && !(count < 127 || *quantile >= 1.0f)) {  // :27-29
            bool found = false;
            float qmn = 0.0f, qmx = 0.0f;
            QAMD_TRY(quantile_interval_device(data, data_mem, count, dim, *quantile, s, found, qmn, qmx));
            if (found) {
                alpha = (qmx - qmn) / 127.0f;
                offset = qmn;
            }
        }
    }

    // PASS 2 (:73-118): quantize rows, stop_condition polled between batches.
    for (uint64_t r0 = 0; r0 < count; r0 += batch_rows) {
        if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");
        const uint64_t nr = std::min(batch_rows, count - r0);
        const float *src = nullptr;
        QAMD_TRY(batch_src(r0, nr, src));
        QAMD_TRY(launch_quantize(h.get(), src, nr, r0, alpha, offset, s));
        if (data_mem == QAMD_MEM_HOST || stop) QAMD_HIP(hipStreamSynchronize(s));
    }
    QAMD_HIP(hipStreamSynchronize(s));
    h->meta.alpha = alpha;
    h->meta.offset = offset;
    h->meta.multiplier = host_multiplier(alpha, vp->distance_type, vp->invert);
    *out = h.release();
    return QAMD_OK;
}

qamd_status qamd_u8_from_rows(const uint8_t *rows, qamd_mem rows_mem, const qamd_u8_metadata *meta,
                              void *stream, qamd_u8 **out) {
    if (!meta || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    const qamd_vector_parameters &vp = meta->vector_parameters;
    if (meta->actual_dim != actual_dim_of(vp.dim))
        return fail(QAMD_ERR_ARGUMENTS, "metadata actual_dim %llu does not match dim %llu",
                    (unsigned long long)meta->actual_dim, (unsigned long long)vp.dim);
    if (vp.count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    if (vp.count > 0 && !rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_ON_DEVICE(current_device());
    hipStream_t s = as_stream(stream);
    std::unique_ptr<qamd_u8> h(new qamd_u8);
    h->device = current_device();
    h->count = vp.count;
    h->meta = *meta;
    QAMD_TRY(alloc_store(h.get()));
    const uint64_t stride = meta->actual_dim + 4;
    const uint32_t row_dwords = (uint32_t)(stride / 4);
    const uint64_t batch_rows = std::max<uint64_t>(1, (256ull << 20) / stride);
    DevBuf stage;
    if (rows_mem == QAMD_MEM_HOST && vp.count)
        QAMD_TRY(stage.alloc(std::min<uint64_t>(batch_rows, vp.count) * stride));
    for (uint64_t r0 = 0; r0 < vp.count; r0 += batch_rows) {
        const uint64_t nr = std::min(batch_rows, vp.count - r0);
        const uint8_t *src = rows + r0 * stride;
        if (rows_mem == QAMD_MEM_HOST) {
            QAMD_TRY(copy_in(stage.ptr, src, QAMD_MEM_HOST, nr * stride, s));
            src = stage.as<uint8_t>();
        }
        int grid = grid_for(nr * row_dwords, kBlock * 4, 8);
        hipLaunchKernelGGL(split_rows_kernel, dim3(grid), dim3(kBlock), 0, s,
                           reinterpret_cast<const uint32_t *>(src), nr, row_dwords,
                           h->codes.as<uint32_t>() + r0 * (row_dwords - 1), h->offsets.as<float>() + r0);
        QAMD_HIP(hipGetLastError());
        if (rows_mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));
    }
    QAMD_HIP(hipStreamSynchronize(s));
    *out = h.release();
    return QAMD_OK;
}

// Rows [first_row, first_row + n_rows) in the reference's storage format: what a caller-owned
// EncodedStorageBuilder receives through push_vector_data (encoded_storage.rs:17-25), in bounded
// pieces -- the store never has to exist as one host buffer (19 GB per C4 shard).
qamd_status qamd_u8_export_rows_range(const qamd_u8 *h, uint64_t first_row, uint64_t n_rows, uint8_t *rows,
                                      qamd_mem rows_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (first_row > h->count || n_rows > h->count - first_row)
        return fail(QAMD_ERR_OUT_OF_RANGE, "rows [%llu, +%llu) out of range (count %llu)", (unsigned long long)first_row,
                    (unsigned long long)n_rows, (unsigned long long)h->count);
    if (n_rows == 0) return QAMD_OK;
    if (!rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    const uint64_t stride = h->meta.actual_dim + 4;
    const uint32_t row_dwords = (uint32_t)(stride / 4);
    const uint64_t batch_rows = std::max<uint64_t>(1, (256ull << 20) / stride);
    DevBuf stage;
    if (rows_mem == QAMD_MEM_HOST) QAMD_TRY(stage.alloc(std::min<uint64_t>(batch_rows, n_rows) * stride));
    for (uint64_t r0 = 0; r0 < n_rows; r0 += batch_rows) {
        const uint64_t nr = std::min(batch_rows, n_rows - r0), src_row = first_row + r0;
        uint8_t *dst = rows_mem == QAMD_MEM_HOST ? stage.as<uint8_t>() : rows + r0 * stride;
        int grid = grid_for(nr * row_dwords, kBlock * 4, 8);
        hipLaunchKernelGGL(join_rows_kernel, dim3(grid), dim3(kBlock), 0, s,
                           h->codes.as<uint32_t>() + src_row * (row_dwords - 1), h->offsets.as<float>() + src_row, nr,
                           row_dwords, reinterpret_cast<uint32_t *>(dst));
        QAMD_HIP(hipGetLastError());
        if (rows_mem == QAMD_MEM_HOST)
            QAMD_TRY(copy_out(rows + r0 * stride, QAMD_MEM_HOST, dst, nr * stride, s));
    }
    return QAMD_OK;
}

qamd_status qamd_u8_export_rows(const qamd_u8 *h, uint8_t *rows, qamd_mem rows_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return qamd_u8_export_rows_range(h, 0, h->count, rows, rows_mem, stream);
}

qamd_status qamd_u8_get_metadata(const qamd_u8 *h, qamd_u8_metadata *out) {
    if (!h || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    *out = h->meta;
    return QAMD_OK;
}

// save (:263-271): serde_json metadata in struct field order, then the raw row file.
qamd_status qamd_u8_save(const qamd_u8 *h, const char *data_path, const char *meta_path) {
    if (!h || !data_path || !meta_path) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::string js = "{\"actual_dim\":" + std::to_string(h->meta.actual_dim) +
                     ",\"alpha\":" + json_f32(h->meta.alpha) + ",\"offset\":" + json_f32(h->meta.offset) +
                     ",\"multiplier\":" + json_f32(h->meta.multiplier) +
                     ",\"vector_parameters\":" + vector_parameters_json(h->meta.vector_parameters) + "}";
    make_parent_dirs(meta_path);
    if (!write_file(meta_path, js.data(), js.size()))
        return fail(QAMD_ERR_IO, "cannot write %s", meta_path);
    const uint64_t stride = h->meta.actual_dim + 4;
    std::vector<uint8_t> rows(h->count * stride);
    QAMD_TRY(qamd_u8_export_rows(h, rows.data(), QAMD_MEM_HOST, nullptr));
    make_parent_dirs(data_path);
    if (!write_file(data_path, rows.data(), rows.size()))
        return fail(QAMD_ERR_IO, "cannot write %s", data_path);
    return QAMD_OK;
}

// load (:273-288): metadata from JSON; row size/count from the CALLER's vector_parameters;
// the file length must equal size*count (encoded_storage.rs:40-51).
qamd_status qamd_u8_load(const char *data_path, const char *meta_path, const qamd_vector_parameters *vp,
                         qamd_u8 **out) {
    if (!data_path || !meta_path || !vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    JsonValue root;
    QAMD_TRY(read_metadata(meta_path, root));
    qamd_u8_metadata meta{};
    {   // Metadata (:24-31), the fields in any order
        std::string err;
        const JsonValue *vpj = nullptr;
        if (!json_usize(root, "actual_dim", meta.actual_dim, err) || !json_f32_field(root, "alpha", meta.alpha, err) ||
            !json_f32_field(root, "offset", meta.offset, err) || !json_f32_field(root, "multiplier", meta.multiplier, err) ||
            !(vpj = json_field(root, "vector_parameters", err)) || !parse_vector_parameters(*vpj, meta.vector_parameters, err))
            return fail(QAMD_ERR_IO, "%s: %s", meta_path, err.c_str());
    }
    const uint64_t size = qamd_u8_quantized_vector_size(vp);
    std::string bytes;
    if (!read_file(data_path, bytes)) return fail(QAMD_ERR_IO, "cannot read %s", data_path);
    const uint64_t expected = size * vp->count;
    if (bytes.size() != expected)
        return fail(QAMD_ERR_IO, "Loaded storage size %zu is not equal to expected size %llu", bytes.size(),
                    (unsigned long long)expected);
    // The reference keeps the file's metadata but sizes rows from the caller's parameters.
    qamd_u8_metadata eff = meta;
    eff.vector_parameters.dim = vp->dim;
    eff.vector_parameters.count = vp->count;
    eff.actual_dim = actual_dim_of(vp->dim);
    qamd_status st = qamd_u8_from_rows(reinterpret_cast<const uint8_t *>(bytes.data()), QAMD_MEM_HOST, &eff,
                                       nullptr, out);
    if (st == QAMD_OK) (*out)->meta = meta.actual_dim == eff.actual_dim ? meta : eff;
    return st;
}

}  // extern "C"

namespace {

// Runs the encode kernel for `query` (host or device f32) into q's buffer on stream s.
qamd_status encode_now(const qamd_u8_query *q, const float *query, uint64_t qdim, qamd_mem query_mem, hipStream_t s) {
    const uint64_t ad = q->actual_dim;
    // The query is always encoded on the device (one implementation, no CPU arithmetic in the
    // product): a host query is uploaded first (qdim * 4 bytes; the buffer keeps room for it
    // behind the codes).
    const float *q_dev = query;
    bool via_scratch = false;
    if (query_mem == QAMD_MEM_HOST && qdim) {
        // through the thread's mapped host scratch when it fits: one memcpy on the host, the kernel
        // reads the values over PCIe itself -- no copy call, no synchronisation
        const float *dev_view = nullptr;
        if (float *hq = host_query_acquire(qdim, &dev_view)) {
            memcpy(hq, query, qdim * 4);
            q_dev = dev_view;
            via_scratch = true;
        } else {
            float *stage = reinterpret_cast<float *>(q->buf.as<uint8_t>() + 16 + round_up(ad, 16));
            QAMD_TRY(copy_in(stage, query, QAMD_MEM_HOST, qdim * 4, s));
            q_dev = stage;
        }
    }
    hipLaunchKernelGGL(encode_query_kernel, dim3(1), dim3(64), 0, s, q_dev, (uint32_t)qdim, (uint32_t)ad, q->alpha,
                       q->offset, q->distance, q->invert, q->buf.as<uint8_t>());
    QAMD_HIP(hipGetLastError());
    if (via_scratch) host_query_release(s);
    return q->ready.record(s);
}

// A deferred host query (qamd_u8_query::host_f32) is encoded now, on the consumer's stream.
qamd_status ensure_encoded(const qamd_u8_query *q, hipStream_t s) {
    if (!q->deferred.load(std::memory_order_acquire)) return QAMD_OK;
    std::lock_guard<std::mutex> lk(q->encode_mu);
    if (!q->deferred.load(std::memory_order_relaxed)) return QAMD_OK;
    QAMD_TRY(encode_now(q, q->host_f32.data(), q->host_f32.size(), QAMD_MEM_HOST, s));
    q->deferred.store(false, std::memory_order_release);
    return QAMD_OK;
}

}  // namespace

extern "C" {

qamd_status qamd_u8_encode_query(const qamd_u8 *h, const float *query, uint64_t qdim, qamd_mem query_mem,
                                 void *stream, qamd_u8_query **query_io) {
    if (!h || !query_io || (!query && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    const uint64_t ad = actual_dim_of(qdim);
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    qamd_u8_query *q = *query_io;
    std::unique_ptr<qamd_u8_query> fresh;
    if (!q) {
        fresh.reset(new qamd_u8_query);
        q = fresh.get();
        q->device = h->device;
    }
    if (q->actual_dim != ad || !q->buf.ptr) {
        if (q->pooled) query_buf_put(q->buf, false);
        QAMD_TRY(query_buf_get(16 + round_up(ad, 16) + ad * 4 + 16, q->buf));  // offset | codes | f32 staging
        q->pooled = true;
        q->actual_dim = ad;
    }
    q->alpha = h->meta.alpha;
    q->offset = h->meta.offset;
    q->distance = h->meta.vector_parameters.distance_type;
    q->invert = h->meta.vector_parameters.invert;
    if (query_mem == QAMD_MEM_HOST && u8_host_encode_is_lazy(qdim)) {
        // Deferred: the values are kept, nothing is launched.  On a small store the top-k kernel takes them
        // by value and quantises them itself (u8_topk_small_fused_kernel); any other consumer encodes first.
        std::lock_guard<std::mutex> lk(q->encode_mu);
        q->host_f32.assign(query, query + qdim);
        q->deferred.store(true, std::memory_order_release);
    } else {
        {
            std::lock_guard<std::mutex> lk(q->encode_mu);
            q->deferred.store(false, std::memory_order_release);
        }
        QAMD_TRY(encode_now(q, query, qdim, query_mem, s));
    }
    if (fresh) *query_io = fresh.release();
    return QAMD_OK;
}

qamd_status qamd_u8_query_read(const qamd_u8_query *q, float *offset, uint8_t *codes, uint64_t capacity,
                               uint64_t *codes_len) {
    if (!q) return fail(QAMD_ERR_ARGUMENTS, "null query");
    QAMD_ON_DEVICE(q->device);
    if (codes_len) *codes_len = q->actual_dim;
    QAMD_TRY(ensure_encoded(q, nullptr));
    QAMD_TRY(q->ready.wait(nullptr));
    if (offset) QAMD_TRY(copy_out(offset, QAMD_MEM_HOST, q->buf.ptr, 4, nullptr));
    if (codes) {
        if (capacity < q->actual_dim) return fail(QAMD_ERR_ARGUMENTS, "codes buffer too small");
        QAMD_TRY(copy_out(codes, QAMD_MEM_HOST, q->buf.as<uint8_t>() + 16, q->actual_dim, nullptr));
    }
    return QAMD_OK;
}

void qamd_u8_query_free(qamd_u8_query *q) { delete q; }

qamd_status qamd_u8_score_all(const qamd_u8 *h, const qamd_u8_query *q, float *out, qamd_mem out_mem,
                              void *stream) {
    QAMD_TRY(check_query(h, q));
    if (h->count == 0) return QAMD_OK;
    if (!out) return fail(QAMD_ERR_ARGUMENTS, "out is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(ensure_encoded(q, s));
    QAMD_TRY(q->ready.wait(s));
    if (out_mem == QAMD_MEM_DEVICE) {
        q->async_used.store(true, std::memory_order_relaxed);
        return scan_into(h, q, out, s);
    }
    float *tmp = nullptr;  // per-thread workspace: no hipMalloc / hipFree per query
    QAMD_TRY(thread_ws_acquire(WS_SCORES, h->count * sizeof(float), s, reinterpret_cast<void **>(&tmp)));
    qamd_status st = scan_into(h, q, tmp, s);
    if (st == QAMD_OK) st = copy_out(out, QAMD_MEM_HOST, tmp, h->count * sizeof(float), s);
    thread_ws_release(WS_SCORES, s, st == QAMD_OK);  // the download synchronised the stream
    return st;
}

qamd_status qamd_u8_score_ids(const qamd_u8 *h, const qamd_u8_query *q, const uint32_t *ids, uint64_t n_ids,
                              qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_query(h, q));
    if (n_ids == 0) return QAMD_OK;
    if (!ids || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    QAMD_TRY(ensure_encoded(q, s));
    QAMD_TRY(q->ready.wait(s));
    if (out_mem == QAMD_MEM_DEVICE) q->async_used.store(true, std::memory_order_relaxed);
    return score_ids_any(h, reinterpret_cast<const uint4 *>(q->buf.as<uint8_t>() + 16), q->buf.as<float>(), 0.0f, EPI_POINT,
                         ids, n_ids, ids_mem, out, out_mem, s);
}

// score_internal (:386-453) for one stored row against many: out[k] = score_internal(i, ids[k]).
qamd_status qamd_u8_score_internal_ids(const qamd_u8 *h, uint32_t i, const uint32_t *ids, uint64_t n_ids,
                                       qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (n_ids == 0) return QAMD_OK;
    if (!ids || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (i >= h->count) return fail(QAMD_ERR_OUT_OF_RANGE, "row id %u out of range (count %llu)", i, (unsigned long long)h->count);
    QAMD_ON_DEVICE(h->device);
    return score_ids_any(h, h->codes.as<uint4>() + (uint64_t)i * h->row_chunks, h->offsets.as<float>() + i, internal_diff(h),
                         EPI_INTERNAL, ids, n_ids, ids_mem, out, out_mem, as_stream(stream));
}

// Many stored rows, each against its own id list, in one launch (lists.hpp):
// out[p] = score_internal(rows[l], ids[p]) for p in [list_offsets[l], list_offsets[l + 1]).
qamd_status qamd_u8_score_internal_ids_batch(const qamd_u8 *h, const uint32_t *rows, const uint32_t *list_offsets,
                                             uint32_t n_lists, const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem,
                                             float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    if (n_lists && !rows) return fail(QAMD_ERR_ARGUMENTS, "rows is null");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    return run_lists(list_offsets, n_lists, ids, n_ids, rows, lists_mem, out, out_mem, h->count, s, [&](const ListArgs &a) {
        return score_pairs_dev(h, nullptr, nullptr, nullptr, 0, nullptr, a.offsets, a.n_lists, a.rows, internal_diff(h),
                               EPI_INTERNAL, a.ids, a.n_pairs, a.out, s);
    });
}

qamd_status qamd_u8_score_point(const qamd_u8 *h, const qamd_u8_query *q, uint32_t i, float *out) {
    return qamd_u8_score_ids(h, q, &i, 1, QAMD_MEM_HOST, out, QAMD_MEM_HOST, nullptr);
}

qamd_status qamd_u8_score_internal(const qamd_u8 *h, uint32_t i, uint32_t j, float *out) {
    if (!h || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    return qamd_u8_score_internal_ids(h, i, &j, 1, QAMD_MEM_HOST, out, QAMD_MEM_HOST, nullptr);
}

qamd_status qamd_u8_topk(const qamd_u8 *h, const qamd_u8_query *q, uint32_t k, int largest, uint32_t *out_ids,
                         float *out_scores, qamd_mem out_mem, void *stream) {
    QAMD_TRY(check_query(h, q));
    if (k == 0) return QAMD_OK;
    if (!out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null output");
    QAMD_ON_DEVICE(h->device);
    hipStream_t s = as_stream(stream);
    if (q->deferred.load(std::memory_order_acquire)) {
        // a host query that has not been encoded yet: on a small store the search is ONE launch, the
        // top-k kernel quantises the query (passed by value) in its prologue and leaves the codes in the
        // query object
        std::unique_lock<std::mutex> lk(q->encode_mu);
        SmallTopkPlan plan;
        if (q->deferred.load(std::memory_order_relaxed) && u8_small_plan(h, k, plan)) {
            FusedQuery fq{q->host_f32.data(), (uint32_t)q->host_f32.size(), q->buf.as<uint8_t>()};
            if (out_mem == QAMD_MEM_DEVICE) q->async_used.store(true, std::memory_order_relaxed);
            qamd_status st = u8_topk_ptrs(h, q->buf.as<uint8_t>() + 16, q->buf.as<float>(), k, largest, out_ids, out_scores,
                                          out_mem, s, &fq);
            if (st == QAMD_OK) {
                st = q->ready.record(s);  // the codes are in the object once this launch has run
                q->deferred.store(false, std::memory_order_release);
            }
            return st;
        }
        lk.unlock();
        QAMD_TRY(ensure_encoded(q, s));
    }
    QAMD_TRY(q->ready.wait(s));
    if (out_mem == QAMD_MEM_DEVICE) q->async_used.store(true, std::memory_order_relaxed);
    return u8_topk_ptrs(h, q->buf.as<uint8_t>() + 16, q->buf.as<float>(), k, largest, out_ids, out_scores, out_mem, s);
}

void qamd_u8_free(qamd_u8 *h) { delete h; }

uint64_t qamd_u8_scan_bytes_per_row(const qamd_u8 *h) { return h ? h->meta.actual_dim + 4 : 0; }

// Selects how sums above 2^24 are rounded (0: once, 1: avx2.c lane order).  Not part of the
// reference surface; see the header of this file.
qamd_status qamd_u8_set_lane_mode(qamd_u8 *h, int mode) {
    if (!h || mode < 0 || mode > 1) return fail(QAMD_ERR_ARGUMENTS, "bad lane mode");
    h->lane_mode = mode;
    return QAMD_OK;
}

}  // extern "C"


// ============================================================================= streaming encode
// The reference encodes from a clonable ITERATOR and walks it twice (encoded_vectors_u8.rs:34-40:
// pass 1 :57-71 find_min_max_from_iter + quantile sample, pass 2 :73-118 quantize + push_vector_data),
// never holding the f32 data.  This is that contract in bounded batches: observe() = pass 1,
// push() = pass 2 (rows appended in call order, as EncodedStorageBuilder::push_vector_data,
// encoded_storage.rs:17-25).  Same kernels and the same sample rows as qamd_u8_encode, so the store
// is byte-identical to the one-shot call.
namespace {

// Rows of a batch that belong to the evenly strided quantile sample: sample slot k holds global row
// floor(k * count / slice).
__global__ __launch_bounds__(kBlock) void gather_sample_batch_kernel(const float *__restrict__ batch, uint64_t r_base,
                                                                    uint32_t dim, uint64_t count, uint64_t slice,
                                                                    uint64_t k0, uint64_t k1, float *__restrict__ sample) {
    const uint64_t total = (k1 - k0) * dim;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t k = k0 + i / dim, j = i % dim;
        const uint64_t r = (uint64_t)((unsigned __int128)k * count / slice);
        sample[k * dim + j] = batch[(r - r_base) * dim + j];
    }
}

uint64_t first_sample_at_or_after(uint64_t row, uint64_t count, uint64_t slice) {
    // smallest k with floor(k * count / slice) >= row  <=>  k * count >= row * slice
    const unsigned __int128 need = (unsigned __int128)row * slice;
    return (uint64_t)((need + count - 1) / count);
}

constexpr uint64_t kStagePieceBytes = 256ull << 20;

}  // namespace

struct qamd_u8_encoder {
    int device = 0;
    hipStream_t stream = nullptr;
    qamd_vector_parameters vp{};
    bool has_quantile = false, has_interval = false;
    float quantile = 0.0f, alpha = 0.0f, offset = 0.0f;
    qamd_stop_fn stop = nullptr;
    void *stop_user = nullptr;
    std::unique_ptr<qamd_u8> h;
    MinMaxAcc acc;
    bool acc_ready = false;
    uint64_t observed = 0, pushed = 0;
    uint64_t slice = 0;  // quantile sample rows (0: no sample kept)
    DevBuf sample, stage;
};

namespace {

qamd_status encoder_close_pass1(qamd_u8_encoder *e) {
    if (e->has_interval) return QAMD_OK;
    const uint64_t count = e->vp.count, dim = e->vp.dim;
    if (count == 0) {  // :43-54: nothing was observed, nothing will be quantized
        e->has_interval = true;
        return QAMD_OK;
    }
    if (e->observed != count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu (observe pass ended early)",
                    (unsigned long long)e->observed, (unsigned long long)count);
    float mn, mx;
    QAMD_TRY(e->acc.result(e->stream, mn, mx));
    e->alpha = (mx - mn) / 127.0f;  // :228-232
    e->offset = mn;
    if (e->slice) {
        uint64_t cut = 0;
        if (quantile_cut(e->slice, dim, e->quantile, cut)) {
            float qmn = 0.0f, qmx = 0.0f;
            const uint64_t len = e->slice * dim;
            QAMD_TRY(select_kth_f32(e->sample.as<float>(), len, cut + 2, false, &qmn, e->stream));
            QAMD_TRY(select_kth_f32(e->sample.as<float>(), len, cut + 1, true, &qmx, e->stream));
            e->alpha = (qmx - qmn) / 127.0f;
            e->offset = qmn;
        }
        e->sample.release();
    }
    e->has_interval = true;
    return QAMD_OK;
}

}  // namespace

extern "C" {

qamd_status qamd_u8_encoder_begin(const qamd_vector_parameters *vp, const float *quantile, const float *alpha_offset,
                                  qamd_stop_fn stop, void *stop_user, void *stream, qamd_u8_encoder **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (vp->distance_type < 0 || vp->distance_type > 2)
        return fail(QAMD_ERR_ARGUMENTS, "bad distance_type %d", vp->distance_type);
    if (vp->count > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
    QAMD_ON_DEVICE(current_device());
    std::unique_ptr<qamd_u8_encoder> e(new qamd_u8_encoder);
    e->device = current_device();
    e->stream = as_stream(stream);
    e->vp = *vp;
    e->stop = stop;
    e->stop_user = stop_user;
    e->h.reset(new qamd_u8);
    e->h->device = e->device;
    e->h->count = vp->count;
    e->h->meta.actual_dim = actual_dim_of(vp->dim);
    e->h->meta.vector_parameters = *vp;
    QAMD_TRY(alloc_store(e->h.get()));
    if (alpha_offset) {
        e->alpha = alpha_offset[0];
        e->offset = alpha_offset[1];
        e->has_interval = true;
    } else if (vp->count) {
        QAMD_TRY(e->acc.init(e->stream));
        e->acc_ready = true;
        if (quantile && !(vp->count < 127 || *quantile >= 1.0f)) {  // quantile.rs:27-29
            e->has_quantile = true;
            e->quantile = *quantile;
            e->slice = std::min<uint64_t>(vp->count, kQuantileSample);
            QAMD_TRY(e->sample.alloc(std::max<uint64_t>(e->slice * vp->dim, 4) * sizeof(float)));
        }
    }
    *out = e.release();
    return QAMD_OK;
}

qamd_status qamd_u8_encoder_observe(qamd_u8_encoder *e, const float *batch, uint64_t n_rows, qamd_mem batch_mem) {
    if (!e || (!batch && n_rows && e->vp.dim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (e->stop && e->stop(e->stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");
    if (e->has_interval) {  // alpha_offset given (or pass 1 closed): nothing to learn
        e->observed += n_rows;
        return QAMD_OK;
    }
    if (e->pushed) return fail(QAMD_ERR_ARGUMENTS, "observe after push");
    if (e->observed + n_rows > e->vp.count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)(e->observed + n_rows), (unsigned long long)e->vp.count);
    QAMD_ON_DEVICE(e->device);
    const uint64_t dim = e->vp.dim, count = e->vp.count;
    if (dim == 0) {
        e->observed += n_rows;
        return QAMD_OK;
    }
    const uint64_t piece_rows = std::max<uint64_t>(1, kStagePieceBytes / (dim * 4));
    for (uint64_t r = 0; r < n_rows; r += piece_rows) {
        const uint64_t nr = std::min(piece_rows, n_rows - r);
        const void *src = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(batch + r * dim, batch_mem, nr * dim * 4, e->stage, e->stream, &src, &staged));
        QAMD_TRY(e->acc.feed(static_cast<const float *>(src), nr * dim, e->stream));
        if (e->slice) {
            const uint64_t base = e->observed + r;
            const uint64_t k0 = first_sample_at_or_after(base, count, e->slice);
            const uint64_t k1 = std::min<uint64_t>(e->slice, first_sample_at_or_after(base + nr, count, e->slice));
            if (k1 > k0) {
                hipLaunchKernelGGL(gather_sample_batch_kernel, dim3(grid_for((k1 - k0) * dim, kBlock * 4, 8)), dim3(kBlock),
                                   0, e->stream, static_cast<const float *>(src), base, (uint32_t)dim, count, e->slice, k0,
                                   k1, e->sample.as<float>());
                QAMD_HIP(hipGetLastError());
            }
        }
        if (staged) QAMD_HIP(hipStreamSynchronize(e->stream));  // the staging buffer is reused
    }
    e->observed += n_rows;
    return QAMD_OK;
}

qamd_status qamd_u8_encoder_push(qamd_u8_encoder *e, const float *batch, uint64_t n_rows, qamd_mem batch_mem) {
    if (!e || (!batch && n_rows && e->vp.dim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (e->stop && e->stop(e->stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");  // :74-76
    if (e->pushed + n_rows > e->vp.count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)(e->pushed + n_rows), (unsigned long long)e->vp.count);
    QAMD_ON_DEVICE(e->device);
    QAMD_TRY(encoder_close_pass1(e));
    const uint64_t dim = e->vp.dim;
    const uint64_t piece_rows = std::max<uint64_t>(1, kStagePieceBytes / (dim * 4 + 1));
    for (uint64_t r = 0; r < n_rows; r += piece_rows) {
        const uint64_t nr = std::min(piece_rows, n_rows - r);
        const void *src = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(batch + r * dim, batch_mem, nr * dim * 4, e->stage, e->stream, &src, &staged));
        QAMD_TRY(launch_quantize(e->h.get(), static_cast<const float *>(src), nr, e->pushed + r, e->alpha, e->offset,
                                 e->stream));
        if (staged) QAMD_HIP(hipStreamSynchronize(e->stream));
    }
    e->pushed += n_rows;
    return QAMD_OK;
}

qamd_status qamd_u8_encoder_finish(qamd_u8_encoder *e, qamd_u8 **out) {
    if (!e || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_u8_encoder> own(e);  // consumed whatever happens
    if (e->pushed != e->vp.count)
        return fail(QAMD_ERR_ARGUMENTS, "Vector count %llu does not match vector parameters count %llu",
                    (unsigned long long)e->pushed, (unsigned long long)e->vp.count);
    QAMD_ON_DEVICE(e->device);
    QAMD_HIP(hipStreamSynchronize(e->stream));
    if (e->vp.count == 0) {  // :43-54
        e->h->meta.alpha = e->h->meta.offset = e->h->meta.multiplier = 0.0f;
    } else {
        e->h->meta.alpha = e->alpha;
        e->h->meta.offset = e->offset;
        e->h->meta.multiplier = host_multiplier(e->alpha, e->vp.distance_type, e->vp.invert);
    }
    *out = e->h.release();
    return QAMD_OK;
}

void qamd_u8_encoder_abort(qamd_u8_encoder *e) {
    if (!e) return;
    DeviceGuard g(e->device);
    (void)hipStreamSynchronize(e->stream);  // kernels may still be writing into the store being dropped
    delete e;
}

}  // extern "C"

namespace qamd {

// Pass-1 pieces for the sharded encoder (sharded.hip): min/max of a row range read from wherever it
// lives (host, this device, another device), and the quantile interval of a whole data set.
qamd_status u8_minmax_range(const float *data, qamd_mem mem, uint64_t n_rows, uint64_t dim, hipStream_t s, float *mn,
                            float *mx) {
    MinMaxAcc acc;
    QAMD_TRY(acc.init(s));
    DevBuf stage;
    const uint64_t piece_rows = std::max<uint64_t>(1, kStagePieceBytes / std::max<uint64_t>(dim * 4, 1));
    for (uint64_t r = 0; r < n_rows && dim; r += piece_rows) {
        const uint64_t nr = std::min(piece_rows, n_rows - r);
        const void *src = nullptr;
        bool staged = false;
        QAMD_TRY(local_view(data + r * dim, mem, nr * dim * 4, stage, s, &src, &staged));
        QAMD_TRY(acc.feed(static_cast<const float *>(src), nr * dim, s));
        if (staged) QAMD_HIP(hipStreamSynchronize(s));
    }
    return acc.result(s, *mn, *mx);
}

qamd_status u8_quantile_interval(const float *data, qamd_mem mem, uint64_t count, uint64_t dim, float quantile,
                                 hipStream_t s, bool *found, float *mn, float *mx) {
    *found = false;
    if (count < 127 || quantile >= 1.0f) return QAMD_OK;  // quantile.rs:27-29
    return quantile_interval_device(data, mem, count, dim, quantile, s, *found, *mn, *mx);
}

}  // namespace qamd

extern "C" {

qamd_status qamd_u8_find_min_max(const float *data, qamd_mem data_mem, uint64_t n_rows, uint64_t dim, void *stream,
                                 float *min, float *max) {
    if (!min || !max || (!data && n_rows && dim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(current_device());
    return qamd::u8_minmax_range(data, data_mem, n_rows, dim, as_stream(stream), min, max);
}

qamd_status qamd_u8_find_quantile_interval(const float *data, qamd_mem data_mem, uint64_t count, uint64_t dim,
                                           float quantile, void *stream, int *found, float *min, float *max) {
    if (!found || !min || !max || (!data && count && dim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    QAMD_ON_DEVICE(current_device());
    bool f = false;
    *min = *max = 0.0f;
    QAMD_TRY(qamd::u8_quantile_interval(data, data_mem, count, dim, quantile, as_stream(stream), &f, min, max));
    *found = f ? 1 : 0;
    return QAMD_OK;
}

}  // extern "C"

namespace qamd {

qamd_status u8_encode_queries_device(const qamd_u8 *h, const float *queries_dev, uint64_t n_queries, uint64_t qdim,
                                     uint8_t *codes_dev, uint64_t code_pitch, float *offsets_dev, hipStream_t stream) {
    if (n_queries == 0) return QAMD_OK;
    const qamd_vector_parameters &vp = h->meta.vector_parameters;
    hipLaunchKernelGGL(encode_queries_kernel, dim3((unsigned)n_queries), dim3(64), 0, stream, queries_dev,
                       (uint32_t)qdim, (uint32_t)actual_dim_of(qdim), h->meta.alpha, h->meta.offset,
                       vp.distance_type, vp.invert, codes_dev, (uint32_t)code_pitch, offsets_dev);
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// The top-k of one query given as raw device pointers (its codes and its offset): the body of
// qamd_u8_topk, also the per-query route of the batch API, which hands in rows of a query batch
// without copying them into a query object.  Runs on the current device.
qamd_status u8_topk_ptrs(const qamd_u8 *h, const uint8_t *codes_dev, const float *qo, uint32_t k, int largest,
                         uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t s, const FusedQuery *fq) {
    const uint4 *qc = reinterpret_cast<const uint4 *>(codes_dev);
    FusedScan scan;
    scan.scan_scores = [&](float *scores, hipStream_t st) { return scan_ptrs(h, qc, qo, scores, st); };
    scan.scan_filter = [&](const TopkFilter &f, hipStream_t st) { return scan_ptrs(h, qc, qo, nullptr, st, &f); };
    scan.score_ids = [&](const uint32_t *ids, uint64_t n_ids, float *out, hipStream_t st) {
        return score_ids_dev(h, qc, qo, 0.0f, EPI_POINT, ids, n_ids, out, st);
    };
    {   // small stores: one launch, no status read-back (device outputs only enqueue)
        SmallTopkPlan plan;
        if (u8_small_plan(h, k, plan)) {
            const bool is_l1 = h->meta.vector_parameters.distance_type == QAMD_L1;
            return small_topk(plan, k, largest, out_ids, out_scores, out_mem, s,
                              [&](const SmallTopk &p, hipStream_t st) {
                                  return is_l1 ? launch_small<true>(h, qc, qo, fq, plan, p, st)
                                               : launch_small<false>(h, qc, qo, fq, plan, p, st);
                              });
        }
    }
    if (fq) return fail(QAMD_ERR_ARGUMENTS, "a deferred query needs the single-launch path");
    if (!fused_capable(h)) {  // rare layouts: classic path only
        float *scores = nullptr;
        QAMD_TRY(thread_ws_acquire(WS_SCORES, std::max<uint64_t>(h->count, 1) * 4, s, reinterpret_cast<void **>(&scores)));
        qamd_status st = scan_ptrs(h, qc, qo, scores, s);
        if (st == QAMD_OK) st = topk_finish(scores, h->count, k, largest, out_ids, out_scores, out_mem, s);
        thread_ws_release(WS_SCORES, s);
        return st;
    }
    return fused_topk(h->count, k, largest, out_ids, out_scores, out_mem, s, scan);
}

// The batch API for the metrics with no matrix form (L1: sum |q - v| is not a contraction): the
// per-query fused pipelines of all queries enqueued back to back (fused_topk_batch), one status
// read-back per 32 queries.  codes_dev: [Q][pitch], offsets_dev: [Q].
qamd_status u8_topk_batch_scans(const qamd_u8 *h, const uint8_t *codes_dev, uint64_t pitch, const float *offsets_dev,
                                uint32_t n_queries, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                                qamd_mem out_mem, hipStream_t stream) {
    auto qc = [&](uint32_t q) { return reinterpret_cast<const uint4 *>(codes_dev + (uint64_t)q * pitch); };
    BatchScan scan;
    scan.filter_capable = fused_capable(h);
    scan.scan_scores = [&](uint32_t q, float *scores, hipStream_t st) { return scan_ptrs(h, qc(q), offsets_dev + q, scores, st); };
    scan.scan_filter = [&](uint32_t q, const TopkFilter &f, hipStream_t st) {
        return scan_ptrs(h, qc(q), offsets_dev + q, nullptr, st, &f);
    };
    scan.score_ids = [&](uint32_t q, const uint32_t *ids, uint64_t n_ids, float *out, hipStream_t st) {
        return score_ids_dev(h, qc(q), offsets_dev + q, 0.0f, EPI_POINT, ids, n_ids, out, st);
    };
    const bool is_l1 = h->meta.vector_parameters.distance_type == QAMD_L1;
    const uint32_t width = multi_width(h);
    if (width)  // one filtering pass over the rows for up to `width` queries
        scan.scan_filter_multi = [&, width, is_l1](uint32_t q, uint32_t left, const TopkFilterSlices &sl, hipStream_t st,
                                                   qamd_status &status) -> uint32_t {
            const uint32_t nq = std::min(left, width);
            if (nq < 2) return 0;
            const uint8_t *qcodes = codes_dev + (uint64_t)q * pitch;
            const bool ok = is_l1 ? launch_multi<true>(h, nq, qcodes, pitch, offsets_dev + q, nullptr, &sl, st)
                                  : launch_multi<false>(h, nq, qcodes, pitch, offsets_dev + q, nullptr, &sl, st);
            if (ok && hipGetLastError() != hipSuccess) status = fail(QAMD_ERR_DEVICE, "u8 multi-query filter launch failed");
            return ok ? nq : 0;
        };
    return fused_topk_batch(h->count, n_queries, k, largest, out_ids, out_scores, out_mem, stream, scan);
}

bool u8_host_encode_is_lazy(uint64_t qdim) { return qdim > 0 && actual_dim_of(qdim) <= kFusedQueryDims; }

// score_ids_batch: query l of a batch ([q][pitch] codes, [q] offsets) against list l (lists.hpp).
qamd_status u8_score_lists(const qamd_u8 *h, const uint8_t *codes_dev, uint64_t pitch, const float *offsets_dev,
                           const ListArgs &a, hipStream_t stream) {
    return score_pairs_dev(h, nullptr, nullptr, codes_dev, (uint32_t)pitch, offsets_dev, a.offsets, a.n_lists, nullptr, 0.0f,
                           EPI_POINT, a.ids, a.n_pairs, a.out, stream);
}

// How many queries the vector-ALU multi-query scan takes per pass for this store (0: none).
uint32_t u8_multi_width(const qamd_u8 *h) { return multi_width(h); }

// out[j * count + row] for queries [0, n_queries) of a batch through the multi-query scan (groups of
// `width`, then smaller groups, then single scans).
qamd_status u8_score_batch_scans(const qamd_u8 *h, const uint8_t *codes_dev, uint64_t pitch, const float *offsets_dev,
                                 uint32_t n_queries, float *out_dev, hipStream_t stream) {
    const bool is_l1 = h->meta.vector_parameters.distance_type == QAMD_L1;
    const uint32_t width = multi_width(h);
    uint32_t q = 0;
    while (q < n_queries) {
        const uint32_t nq = std::min(width, n_queries - q);
        const uint8_t *qcodes = codes_dev + (uint64_t)q * pitch;
        float *o = out_dev + (uint64_t)q * h->count;
        const bool ok = nq >= 2 && (is_l1 ? launch_multi<true>(h, nq, qcodes, pitch, offsets_dev + q, o, nullptr, stream)
                                          : launch_multi<false>(h, nq, qcodes, pitch, offsets_dev + q, o, nullptr, stream));
        if (ok) {
            q += nq;
        } else {
            QAMD_TRY(scan_ptrs(h, reinterpret_cast<const uint4 *>(qcodes), offsets_dev + q, o, stream));
            q += 1;
        }
    }
    QAMD_HIP(hipGetLastError());
    return QAMD_OK;
}

// Exact single-query top-k for one member of a query batch (its per-query fallback): no copy, no
// allocation; on small stores (single-launch path) with device outputs it only enqueues.
qamd_status u8_topk_single(const qamd_u8 *h, const uint8_t *codes_dev, const float *offset_dev, uint32_t k,
                           int largest, uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t stream) {
    return u8_topk_ptrs(h, codes_dev, offset_dev, k, largest, out_ids, out_scores, out_mem, stream, nullptr);
}

// score_all for one member of a query batch (the L1 route of the batch API: L1 has no MFMA form).
qamd_status u8_score_single(const qamd_u8 *h, const uint8_t *codes_dev, const float *offset_dev, float *out_dev,
                            hipStream_t stream) {
    return scan_ptrs(h, reinterpret_cast<const uint4 *>(codes_dev), offset_dev, out_dev, stream);
}

}  // namespace qamd

#ifdef QAMD_DEV
// Developer-only accessors for the tuning harness (tune.hip): libquantization_amd_dev.so only.
extern "C" __attribute__((visibility("default"))) void qamd_dev_u8_ptrs(const qamd_u8 *h, const void **codes,
                                                                        const void **offsets) {
    *codes = h->codes.ptr;
    *offsets = h->offsets.ptr;
}
extern "C" __attribute__((visibility("default"))) const void *qamd_dev_u8_query_ptr(const qamd_u8_query *q) {
    return q->buf.ptr;
}
#endif
