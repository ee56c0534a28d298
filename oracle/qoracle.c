/*
 * qoracle.c — CPU restatement of qdrant/quantization's encode-and-score path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and the CPU-baseline legs of the two measurement
 * harnesses (bench.py's cpu_baseline; the CPU column of tools/ann_protocol.py,
 * the reference's own ann_benchmark protocol) may build, load or call it - as
 * the checker or the timed CPU side, never as the thing shipped.  The product
 * (quantization_amd/) never links or imports it and has no CPU fallback.
 *
 * Every function cites the reference file:line (under /root/reference/) whose
 * arithmetic it restates.  Plain C, sequential f32 in the reference's exact
 * operation order; compile with -ffp-contract=off and WITHOUT -ffast-math
 * (oracle/Makefile does), because Rust never contracts a*b+c into an FMA.
 *
 * Pinning: the pair-level kernels below (qo_dot_*, qo_l1_*, qo_xor_popcnt) are
 * differential-tested against the reference's own C kernels compiled from
 * /root/reference/quantization/cpp/{avx2,sse}.c into oracle/_ref/ (see
 * oracle/Makefile target `ref`) and against the committed golden vectors in
 * tests/golden/ that were produced by that build.  The Rust half (encode,
 * encode_query, epilogue) cannot be compiled here (no rustc); it is pinned by
 * the reference tests' known-answer / tolerance properties
 * (quantization/tests/test_binary.rs:14-71, test_simple.rs:15-49,
 * test_pq.rs:16-50) which tests/test_oracle_reference_spec.py re-runs on it.
 * k-means (qo_kmeans / qo_find_centroids) restates kmeans.rs:7-167 and
 * encoded_vectors_pq.rs:278-342 GIVEN the sampled row indices; the two random
 * draws of the reference (the Permutor sample, :300-302, and the thread_rng
 * re-seed of an empty cluster, kmeans.rs:111-118) are inputs / a stated fixed
 * rule here, so centroid values are pinned only conditionally on them.
 * qo_topk_heap restates the caller's 30-entry heap
 * (demos/src/ann_benchmark_data.rs:20-33,151-167) over Rust std's BinaryHeap.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QO_EXPORT __attribute__((visibility("default")))

enum { QO_DOT = 0, QO_L1 = 1, QO_L2 = 2 };

#define QO_ALIGNMENT 16 /* encoded_vectors_u8.rs:12 */

typedef struct {
    uint64_t actual_dim;
    float alpha;
    float offset;
    float multiplier;
    uint64_t dim;
    uint64_t count;
    int32_t distance_type;
    int32_t invert;
} qo_u8_meta; /* encoded_vectors_u8.rs:24-31 + encoded_vectors.rs:13-19 */

typedef float (*qo_pair_f32_fn)(const uint8_t *, const uint8_t *, uint32_t);

/* ------------------------------------------------------------------------ */
/* quantile.rs:5-19 find_min_max_from_iter                                   */
QO_EXPORT void qo_find_min_max(const float *data, uint64_t n_values, float *out_min,
                               float *out_max) {
    float mn = 3.40282347e+38f;  /* f32::MAX */
    float mx = -3.40282347e+38f; /* f32::MIN */
    for (uint64_t i = 0; i < n_values; i++) {
        float v = data[i];
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    *out_min = mn;
    *out_max = mx;
}

/* encoded_vectors_u8.rs:228-232 alpha_offset_from_min_max */
QO_EXPORT void qo_u8_alpha_offset(float mn, float mx, float *alpha, float *offset) {
    *alpha = (mx - mn) / 127.0f;
    *offset = mn;
}

/* encoded_vectors_u8.rs:234-237 f32_to_u8:
 *   ((i - offset) / alpha).clamp(0.0, 127.0) as u8
 * f32::clamp keeps NaN as NaN; `as u8` truncates toward zero, saturates, NaN -> 0. */
QO_EXPORT uint8_t qo_f32_to_u8(float v, float alpha, float offset) {
    float x = (v - offset) / alpha;
    if (x < 0.0f) x = 0.0f;
    if (x > 127.0f) x = 127.0f;
    if (x != x) return 0;
    return (uint8_t)x; /* 0 <= x <= 127: C truncation == Rust `as u8` */
}

/* encoded_vectors_u8.rs:257-259 get_actual_dim */
QO_EXPORT uint64_t qo_u8_actual_dim(uint64_t dim) {
    return dim + (QO_ALIGNMENT - dim % QO_ALIGNMENT) % QO_ALIGNMENT;
}

/* encoded_vectors_u8.rs:252-255 get_quantized_vector_size */
QO_EXPORT uint64_t qo_u8_quantized_vector_size(uint64_t dim) {
    return qo_u8_actual_dim(dim) + sizeof(float);
}

/* quantile.rs:21-71 find_quantile_interval, for the deterministic case
 * count <= QUANTILE_SAMPLE_SIZE (the Permutor then selects every vector, :31-34).
 * Returns 1 and writes (min,max) of the kept order statistics, 0 for None.
 * Two select_nth_unstable calls keep the values of sorted rank
 * (cut_index, len - cut_index) exclusive on both ends (:58-61). */
static int qo_cmp_f32(const void *a, const void *b) {
    float x = *(const float *)a, y = *(const float *)b;
    return (x < y) ? -1 : (x > y) ? 1 : 0;
}
QO_EXPORT int qo_find_quantile_interval(const float *data, uint64_t dim, uint64_t count,
                                        float quantile, float *out_min, float *out_max) {
    if (count < 127 || quantile >= 1.0f) return 0; /* :27-29 */
    uint64_t slice_size = count < 100000 ? count : 100000; /* :31 */
    if (slice_size != count) return -1; /* random sample: outside the deterministic oracle */
    uint64_t len = slice_size * dim;
    if (len < 4) return 0; /* :48-50 */
    uint64_t a = (len - 1) / 2;
    uint64_t b = (uint64_t)((float)slice_size * (1.0f - quantile) / 2.0f); /* :52-55 */
    uint64_t cut = a < b ? a : b;
    if (cut < 1) cut = 1; /* :56 */
    float *s = (float *)malloc(len * sizeof(float));
    memcpy(s, data, len * sizeof(float));
    qsort(s, len, sizeof(float), qo_cmp_f32);
    /* first select: left part = sorted[0 .. len-cut); second select on it at `cut`:
     * right part = sorted[cut+1 .. len-cut) */
    uint64_t lo = cut + 1, hi = len - cut; /* [lo, hi) */
    if (hi <= lo || hi - lo < 2) { /* :63-65 */
        free(s);
        return 0;
    }
    qo_find_min_max(s + lo, hi - lo, out_min, out_max);
    free(s);
    return 1;
}

/* encoded_vectors_u8.rs:119-128 multiplier */
static float qo_u8_multiplier(float alpha, int distance, int invert) {
    float m;
    if (distance == QO_DOT)
        m = alpha * alpha;
    else if (distance == QO_L1)
        m = alpha;
    else
        m = -2.0f * alpha * alpha;
    return invert ? -m : m;
}

/* encoded_vectors_u8.rs:73-118: one row = [vector_offset f32 ne][dim codes][pad codes] */
QO_EXPORT void qo_u8_encode_row(const float *vec, uint64_t dim, float alpha, float offset,
                                int distance, int invert, uint8_t *row /* actual_dim+4 */) {
    uint64_t actual_dim = qo_u8_actual_dim(dim);
    uint8_t *codes = row + 4;
    for (uint64_t j = 0; j < dim; j++) codes[j] = qo_f32_to_u8(vec[j], alpha, offset);
    if (dim % QO_ALIGNMENT != 0) { /* :84-93 */
        float placeholder = (distance == QO_DOT) ? 0.0f : offset;
        uint8_t pc = qo_f32_to_u8(placeholder, alpha, offset);
        for (uint64_t j = dim; j < actual_dim; j++) codes[j] = pc;
    }
    /* :94-109; the iterator also covers the four zero placeholder bytes, adding 0.0 */
    float vo;
    if (distance == QO_DOT) {
        float s = 0.0f;
        for (uint64_t j = 0; j < actual_dim; j++) s += (float)codes[j];
        vo = (float)actual_dim * offset * offset + s * alpha * offset;
    } else if (distance == QO_L1) {
        vo = 0.0f;
    } else {
        float s = 0.0f;
        for (uint64_t j = 0; j < actual_dim; j++) s += (float)codes[j] * (float)codes[j];
        vo = (float)actual_dim * offset * offset + s * alpha * alpha;
    }
    if (invert) vo = -vo; /* :110-114 */
    memcpy(row, &vo, 4);  /* :115-116 native-endian */
}

/* encoded_vectors_u8.rs:34-140 EncodedVectorsU8::encode (quantile: <0 means None).
 * rows must hold count*(actual_dim+4) bytes.  Returns 0, or 4 (Stopped) never here. */
QO_EXPORT int qo_u8_encode(const float *data, uint64_t count, uint64_t dim, int distance,
                           int invert, float quantile, uint8_t *rows, qo_u8_meta *meta) {
    uint64_t actual_dim = qo_u8_actual_dim(dim);
    meta->actual_dim = actual_dim;
    meta->dim = dim;
    meta->count = count;
    meta->distance_type = distance;
    meta->invert = invert;
    if (count == 0) { /* :43-54 */
        meta->alpha = 0.0f;
        meta->offset = 0.0f;
        meta->multiplier = 0.0f;
        return 0;
    }
    float mn, mx, alpha, offset;
    qo_find_min_max(data, count * dim, &mn, &mx); /* :57 */
    qo_u8_alpha_offset(mn, mx, &alpha, &offset);
    if (quantile >= 0.0f) { /* :58-71 */
        float qmn, qmx;
        int r = qo_find_quantile_interval(data, dim, count, quantile, &qmn, &qmx);
        if (r < 0) return -1;
        if (r == 1) qo_u8_alpha_offset(qmn, qmx, &alpha, &offset);
    }
    uint64_t stride = actual_dim + 4;
    for (uint64_t i = 0; i < count; i++)
        qo_u8_encode_row(data + i * dim, dim, alpha, offset, distance, invert, rows + i * stride);
    meta->alpha = alpha;
    meta->offset = offset;
    meta->multiplier = qo_u8_multiplier(alpha, distance, invert);
    return 0;
}

/* Same as qo_u8_encode with (alpha, offset) given — the conditional-parity form
 * used where the reference's interval comes from a random sample (count>100k). */
QO_EXPORT void qo_u8_encode_with(const float *data, uint64_t count, uint64_t dim, int distance,
                                 int invert, float alpha, float offset, uint8_t *rows,
                                 qo_u8_meta *meta) {
    uint64_t actual_dim = qo_u8_actual_dim(dim);
    uint64_t stride = actual_dim + 4;
    for (uint64_t i = 0; i < count; i++)
        qo_u8_encode_row(data + i * dim, dim, alpha, offset, distance, invert, rows + i * stride);
    meta->actual_dim = actual_dim;
    meta->dim = dim;
    meta->count = count;
    meta->distance_type = distance;
    meta->invert = invert;
    meta->alpha = alpha;
    meta->offset = offset;
    meta->multiplier = qo_u8_multiplier(alpha, distance, invert);
}

/* encoded_vectors_u8.rs:290-329 encode_query -> EncodedQueryU8{offset, encoded_query} */
QO_EXPORT float qo_u8_encode_query(const qo_u8_meta *m, const float *query, uint64_t qdim,
                                   uint8_t *codes /* actual_dim(qdim) */) {
    uint64_t n = qdim;
    for (uint64_t j = 0; j < qdim; j++) codes[j] = qo_f32_to_u8(query[j], m->alpha, m->offset);
    if (qdim % QO_ALIGNMENT != 0) { /* :296-306 */
        float placeholder = (m->distance_type == QO_DOT) ? 0.0f : m->offset;
        uint8_t pc = qo_f32_to_u8(placeholder, m->alpha, m->offset);
        uint64_t pad = QO_ALIGNMENT - qdim % QO_ALIGNMENT;
        for (uint64_t j = 0; j < pad; j++) codes[n++] = pc;
    }
    float off;
    if (m->distance_type == QO_DOT) { /* :308-312 */
        float s = 0.0f;
        for (uint64_t j = 0; j < n; j++) s += (float)codes[j];
        off = s * m->alpha * m->offset;
    } else if (m->distance_type == QO_L1) {
        off = 0.0f;
    } else { /* :314-318 */
        float s = 0.0f;
        for (uint64_t j = 0; j < n; j++) s += (float)codes[j] * (float)codes[j];
        off = s * m->alpha * m->alpha;
    }
    return m->invert ? -off : off; /* :320-324 */
}

/* ------------------------------------------------------------------------ */
/* Pair kernels.                                                             */

/* encoded_vectors_u8.rs:456-464 impl_score_dot (scalar, i32) */
QO_EXPORT int32_t qo_dot_i32(const uint8_t *q, const uint8_t *v, uint32_t dim) {
    int32_t s = 0;
    for (uint32_t i = 0; i < dim; i++) s += (int32_t)q[i] * (int32_t)v[i];
    return s;
}

/* encoded_vectors_u8.rs:466-474 impl_score_l1 (scalar, i32) */
QO_EXPORT int32_t qo_l1_i32(const uint8_t *q, const uint8_t *v, uint32_t dim) {
    int32_t s = 0;
    for (uint32_t i = 0; i < dim; i++) {
        int32_t d = (int32_t)q[i] - (int32_t)v[i];
        s += d < 0 ? -d : d;
    }
    return s;
}

QO_EXPORT float qo_dot_simple(const uint8_t *q, const uint8_t *v, uint32_t dim) {
    return (float)qo_dot_i32(q, v, dim); /* `score as f32`, encoded_vectors_u8.rs:158 */
}
QO_EXPORT float qo_l1_simple(const uint8_t *q, const uint8_t *v, uint32_t dim) {
    return (float)qo_l1_i32(q, v, dim);
}

/* cpp/avx2.c:25-63 impl_score_dot_avx, restated lane by lane in scalar C.
 * maddubs pairs bytes (2p,2p+1); cvtepi16_epi32 of the low/high 128-bit halves
 * sends pair p (0..15) of each 32-byte block to i32 lane p%8 (:41-45).  The 16-byte
 * tail (:49-58) sends byte b to lane b/2.  Lanes are converted to f32 one by one
 * (:59) and summed ((l0+l4)+(l2+l6))+((l1+l5)+(l3+l7)) (HSUM256_PS, :7-14).
 * maddubs treats the second operand as signed and saturates to i16: with codes
 * <= 127 neither matters; both are modelled anyway for adversarial inputs. */
static int32_t qo_sat16(int32_t x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }
QO_EXPORT float qo_dot_avx2_order(const uint8_t *q, const uint8_t *v, uint32_t dim) {
    int32_t lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t blocks = dim / 32;
    for (uint32_t b = 0; b < blocks; b++) {
        const uint8_t *vb = v + 32 * b, *qb = q + 32 * b;
        for (int p = 0; p < 16; p++) {
            /* _mm256_maddubs_epi16(v, q): v unsigned, q signed */
            int32_t t = (int32_t)vb[2 * p] * (int32_t)(int8_t)qb[2 * p] +
                        (int32_t)vb[2 * p + 1] * (int32_t)(int8_t)qb[2 * p + 1];
            lane[p % 8] = (int32_t)((uint32_t)lane[p % 8] + (uint32_t)qo_sat16(t));
        }
    }
    if (dim % 32 != 0) {
        const uint8_t *vb = v + 32 * blocks, *qb = q + 32 * blocks;
        for (int e = 0; e < 16; e++) {
            uint32_t prod = ((uint32_t)vb[e] * (uint32_t)qb[e]) & 0xFFFFu; /* mullo_epi16 */
            lane[e / 2] = (int32_t)((uint32_t)lane[e / 2] + prod);
        }
    }
    float f[8];
    for (int k = 0; k < 8; k++) f[k] = (float)lane[k];
    float a0 = f[0] + f[4], a1 = f[1] + f[5], a2 = f[2] + f[6], a3 = f[3] + f[7];
    float b0 = a0 + a2, b1 = a1 + a3;
    return b0 + b1;
}

/* cpp/sse.c:23-47 impl_score_dot_sse: 4 i32 lanes; pair p (0..7) of each 16-byte
 * block goes to lane p%4 (:36-40); f32 hsum (l0+l2)+(l1+l3) (HSUM128_PS, :7-13). */
QO_EXPORT float qo_dot_sse_order(const uint8_t *q, const uint8_t *v, uint32_t dim) {
    int32_t lane[4] = {0, 0, 0, 0};
    for (uint32_t b = 0; b < dim / 16; b++) {
        const uint8_t *vb = v + 16 * b, *qb = q + 16 * b;
        for (int p = 0; p < 8; p++) {
            int32_t t = (int32_t)vb[2 * p] * (int32_t)(int8_t)qb[2 * p] +
                        (int32_t)vb[2 * p + 1] * (int32_t)(int8_t)qb[2 * p + 1];
            lane[p % 4] = (int32_t)((uint32_t)lane[p % 4] + (uint32_t)qo_sat16(t));
        }
    }
    float f0 = (float)lane[0], f1 = (float)lane[1], f2 = (float)lane[2], f3 = (float)lane[3];
    return (f0 + f2) + (f1 + f3);
}

/* cpp/avx2.c:65-122 impl_score_l1_avx: u16 lane sums then widened; the result is
 * the exact integer for dim <= ~8256; restated as the exact integer. */
QO_EXPORT float qo_l1_avx2_order(const uint8_t *q, const uint8_t *v, uint32_t dim) {
    /* 16 u16 lanes (wrap mod 2^16 per lane, :87-88), tail added as u16 too (:110-111) */
    uint16_t lane[16];
    memset(lane, 0, sizeof lane);
    uint32_t m = dim - dim % 32;
    for (uint32_t i = 0; i < m; i += 32) {
        /* unpacklo/hi_epi8 within each 128-bit half: byte e of half h -> u16 lane
         * (e%8) + 8*h, from the lo (e<8) or hi (e>=8) unpack; both are added. */
        for (int h = 0; h < 2; h++)
            for (int e = 0; e < 16; e++) {
                int a = q[i + 16 * h + e], b = v[i + 16 * h + e];
                uint16_t d = (uint16_t)(a > b ? a - b : b - a);
                lane[(e % 8) + 8 * h] = (uint16_t)(lane[(e % 8) + 8 * h] + d);
            }
    }
    if (m < dim) {
        /* :96-112 the 16 diffs widen u8->u16->u32 into 8 i32 lanes each (lo: e 0..7,
         * hi: e 8..15), added with add_epi16: i32 lane k = u16 lanes 2k (value), 2k+1 (0) */
        for (int e = 0; e < 16; e++) {
            int a = q[m + e], b = v[m + e];
            uint16_t d = (uint16_t)(a > b ? a - b : b - a);
            int k = e % 8;
            lane[2 * k] = (uint16_t)(lane[2 * k] + d);
        }
    }
    uint32_t total = 0;
    for (int k = 0; k < 16; k++) total += lane[k];
    return (float)(int32_t)total;
}

/* cpp/sse.c:49-106 impl_xor_popcnt_sse_uint{128,64,32}: sum of popcounts over bytes */
QO_EXPORT uint32_t qo_xor_popcnt(const uint8_t *q, const uint8_t *v, uint32_t n_bytes) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < n_bytes; i++) r += (uint32_t)__builtin_popcount(q[i] ^ v[i]);
    return r;
}

/* ------------------------------------------------------------------------ */
/* encoded_vectors_u8.rs:331-384 score_point; order 0 = simple(:142-159, i32 as f32),
 * 1 = AVX2 (:336-349), 2 = SSE (:352-365).  Epilogue :347  (m*s + qo) + vo, no FMA. */
static float qo_u8_pair(const qo_u8_meta *m, const uint8_t *q, const uint8_t *v, int order,
                        qo_pair_f32_fn ref_dot, qo_pair_f32_fn ref_l1) {
    uint32_t d = (uint32_t)m->actual_dim;
    if (m->distance_type == QO_L1) {
        if (ref_l1) return ref_l1(q, v, d);
        return order == 1 ? qo_l1_avx2_order(q, v, d) : qo_l1_simple(q, v, d);
    }
    if (ref_dot) return ref_dot(q, v, d);
    if (order == 1) return qo_dot_avx2_order(q, v, d);
    if (order == 2) return qo_dot_sse_order(q, v, d);
    return qo_dot_simple(q, v, d);
}

QO_EXPORT float qo_u8_score_point(const qo_u8_meta *m, const uint8_t *rows, const uint8_t *qcodes,
                                  float qoffset, uint64_t i, int order) {
    uint64_t stride = m->actual_dim + 4;
    const uint8_t *row = rows + stride * i; /* :239-250 get_vec_ptr */
    float vo;
    memcpy(&vo, row, 4);
    float s = qo_u8_pair(m, qcodes, row + 4, order, NULL, NULL);
    return m->multiplier * s + qoffset + vo;
}

/* The caller loop the GPU path replaces: demos/src/ann_benchmark.rs:247-252
 *   for i in 0..n { scores[i] = score_point(&q, i) }
 * ref_dot / ref_l1 may point at the compiled reference kernels (oracle/_ref). */
QO_EXPORT void qo_u8_score_all(const qo_u8_meta *m, const uint8_t *rows, const uint8_t *qcodes,
                               float qoffset, uint64_t begin, uint64_t end, int order,
                               qo_pair_f32_fn ref_dot, qo_pair_f32_fn ref_l1, float *out) {
    uint64_t stride = m->actual_dim + 4;
    for (uint64_t i = begin; i < end; i++) {
        const uint8_t *row = rows + stride * i;
        float vo;
        memcpy(&vo, row, 4);
        float s = qo_u8_pair(m, qcodes, row + 4, order, ref_dot, ref_l1);
        out[i - begin] = m->multiplier * s + qoffset + vo;
    }
}

/* encoded_vectors_u8.rs:386-453 score_internal */
QO_EXPORT float qo_u8_score_internal(const qo_u8_meta *m, const uint8_t *rows, uint64_t i,
                                     uint64_t j, int order) {
    uint64_t stride = m->actual_dim + 4;
    const uint8_t *ri = rows + stride * i, *rj = rows + stride * j;
    float qo_, vo;
    memcpy(&qo_, ri, 4);
    memcpy(&vo, rj, 4);
    float diff = (float)m->actual_dim * m->offset * m->offset; /* :389 */
    if (m->invert) diff = -diff;
    float offset = qo_ + vo - diff; /* :395 */
    float s = qo_u8_pair(m, ri + 4, rj + 4, order, NULL, NULL);
    return m->multiplier * s + offset; /* :409 */
}

/* ------------------------------------------------------------------------ */
/* Binary quantization.                                                      */

/* encoded_vectors_binary.rs:99-116 (u8 store) / :152-159 (u128 store), in BYTES
 * (= get_quantized_vector_size_from_params, :210-213).  store: 0 = u8, 1 = u128. */
QO_EXPORT uint64_t qo_bin_row_bytes(uint64_t dim, int store) {
    if (store == 1) {
        uint64_t r = dim / 128;
        if (dim % 128 != 0) r += 1;
        return r * 16;
    }
    uint64_t bytes_count = dim > 128 ? 16 : dim > 64 ? 8 : dim > 32 ? 4 : 1;
    uint64_t bits = 8 * bytes_count;
    uint64_t r = dim / bits;
    if (dim % bits != 0) r += 1;
    return r * bytes_count;
}

/* encoded_vectors_binary.rs:193-208 encode_vector: bit i set iff v[i] > 0.0, element
 * i/bits, bit i%bits; little-endian => byte i/8, bit i%8 for both store types. */
QO_EXPORT void qo_bin_encode_vector(const float *vec, uint64_t dim, int store, uint8_t *row) {
    uint64_t nb = qo_bin_row_bytes(dim, store);
    memset(row, 0, nb);
    for (uint64_t i = 0; i < dim; i++)
        if (vec[i] > 0.0f) row[i / 8] |= (uint8_t)(1u << (i % 8));
}

QO_EXPORT void qo_bin_encode(const float *data, uint64_t count, uint64_t dim, int store,
                             uint8_t *rows) {
    uint64_t nb = qo_bin_row_bytes(dim, store);
    for (uint64_t i = 0; i < count; i++) qo_bin_encode_vector(data + i * dim, dim, store, rows + i * nb);
}

/* encoded_vectors_binary.rs:219-253 calculate_metric */
QO_EXPORT float qo_bin_metric(uint32_t xor_popcnt, uint64_t dim, int distance, int invert) {
    float xor_product = (float)xor_popcnt;
    float d = (float)dim;
    float zeros_count = d - xor_product;
    if (distance == QO_DOT) return invert ? xor_product - zeros_count : zeros_count - xor_product;
    return invert ? zeros_count - xor_product : xor_product - zeros_count;
}

/* encoded_vectors_binary.rs:293-300 score_point */
QO_EXPORT float qo_bin_score_point(const uint8_t *rows, const uint8_t *q, uint64_t dim, int store,
                                   int distance, int invert, uint64_t i) {
    uint64_t nb = qo_bin_row_bytes(dim, store);
    return qo_bin_metric(qo_xor_popcnt(q, rows + nb * i, (uint32_t)nb), dim, distance, invert);
}

typedef uint32_t (*qo_pair_u32_fn)(const uint8_t *, const uint8_t *, uint32_t);

/* scan loop; ref_popcnt128 may be the compiled reference impl_xor_popcnt_sse_uint128
 * (used when the row is a whole number of 16-byte blocks, as
 * encoded_vectors_binary.rs:51-56 / :126-131 do). */
QO_EXPORT void qo_bin_score_all(const uint8_t *rows, const uint8_t *q, uint64_t dim, int store,
                                int distance, int invert, uint64_t begin, uint64_t end,
                                qo_pair_u32_fn ref_popcnt128, float *out) {
    uint64_t nb = qo_bin_row_bytes(dim, store);
    for (uint64_t i = begin; i < end; i++) {
        uint32_t x;
        if (ref_popcnt128 && nb % 16 == 0 && (store == 1 || nb > 16))
            x = ref_popcnt128(q, rows + nb * i, (uint32_t)(nb / 16));
        else
            x = qo_xor_popcnt(q, rows + nb * i, (uint32_t)nb);
        out[i - begin] = qo_bin_metric(x, dim, distance, invert);
    }
}

/* encoded_vectors_binary.rs:302-314 score_internal */
QO_EXPORT float qo_bin_score_internal(const uint8_t *rows, uint64_t dim, int store, int distance,
                                      int invert, uint64_t i, uint64_t j) {
    uint64_t nb = qo_bin_row_bytes(dim, store);
    return qo_bin_metric(qo_xor_popcnt(rows + nb * i, rows + nb * j, (uint32_t)nb), dim, distance,
                         invert);
}

/* ------------------------------------------------------------------------ */
/* Product quantization (given centroids).                                   */

/* encoded_vectors_pq.rs:109-114 get_quantized_vector_size == number of chunks */
QO_EXPORT uint64_t qo_pq_chunks(uint64_t dim, uint64_t chunk_size) {
    return (dim + chunk_size - 1) / chunk_size;
}

/* encoded_vectors.rs:37-45 DistanceType::distance, sequential f32 */
static float qo_distance(int distance, const float *a, const float *b, uint64_t n) {
    float s = 0.0f;
    if (distance == QO_DOT)
        for (uint64_t i = 0; i < n; i++) s += a[i] * b[i];
    else if (distance == QO_L1)
        for (uint64_t i = 0; i < n; i++) s += fabsf(a[i] - b[i]);
    else
        for (uint64_t i = 0; i < n; i++) s += (a[i] - b[i]) * (a[i] - b[i]);
    return s;
}

/* encoded_vectors_pq.rs:290-297: count <= 256 -> centroids are the vectors, rest zero.
 * centroids: [256][dim] centroid-major, full-dim rows (:39-44, :336-338). */
QO_EXPORT void qo_pq_centroids_small(const float *data, uint64_t count, uint64_t dim,
                                     float *centroids) {
    memset(centroids, 0, 256 * dim * sizeof(float));
    memcpy(centroids, data, count * dim * sizeof(float));
}

/* encoded_vectors_pq.rs:237-265 encode_vector: per chunk argmin of sequential
 * sum (a-b).powi(2), strict '<' (lowest index wins), f32::MAX start. */
QO_EXPORT void qo_pq_encode_vector(const float *vec, uint64_t dim, uint64_t chunk_size,
                                   const float *centroids, uint8_t *codes) {
    uint64_t m = qo_pq_chunks(dim, chunk_size);
    for (uint64_t c = 0; c < m; c++) {
        uint64_t lo = c * chunk_size;
        uint64_t hi = lo + chunk_size < dim ? lo + chunk_size : dim; /* :116-121 */
        float min_d = 3.40282347e+38f;
        uint64_t min_i = 0;
        for (uint64_t k = 0; k < 256; k++) {
            const float *cen = centroids + k * dim;
            float d = 0.0f;
            for (uint64_t j = lo; j < hi; j++) {
                float t = vec[j] - cen[j];
                d += t * t;
            }
            if (d < min_d) {
                min_d = d;
                min_i = k;
            }
        }
        codes[c] = (uint8_t)min_i;
    }
}

QO_EXPORT void qo_pq_encode(const float *data, uint64_t count, uint64_t dim, uint64_t chunk_size,
                            const float *centroids, uint8_t *rows) {
    uint64_t m = qo_pq_chunks(dim, chunk_size);
    for (uint64_t i = 0; i < count; i++)
        qo_pq_encode_vector(data + i * dim, dim, chunk_size, centroids, rows + i * m);
}

/* encoded_vectors_pq.rs:525-547 encode_query: lut[chunk*256 + c] = +-distance */
QO_EXPORT void qo_pq_encode_query(const float *query, uint64_t dim, uint64_t chunk_size,
                                  const float *centroids, int distance, int invert, float *lut) {
    uint64_t m = qo_pq_chunks(dim, chunk_size);
    for (uint64_t c = 0; c < m; c++) {
        uint64_t lo = c * chunk_size;
        uint64_t hi = lo + chunk_size < dim ? lo + chunk_size : dim;
        for (uint64_t k = 0; k < 256; k++) {
            float d = qo_distance(distance, query + lo, centroids + k * dim + lo, hi - lo);
            lut[c * 256 + k] = invert ? -d : d;
        }
    }
}

/* encoded_vectors_pq.rs:476-494 score_point_simple: sequential sum */
QO_EXPORT float qo_pq_score_simple(const uint8_t *codes, uint64_t m, const float *lut) {
    float s = 0.0f;
    for (uint64_t c = 0; c < m; c++) s += lut[c * 256 + codes[c]];
    return s;
}

/* encoded_vectors_pq.rs:405-440 score_point_sse: lane k sums chunks 4t+k in order;
 * (l0+l2)+(l1+l3) (:430-432); then the len%4 tail sequentially (:434-438). */
QO_EXPORT float qo_pq_score_sse_order(const uint8_t *codes, uint64_t m, const float *lut) {
    float l[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    uint64_t q4 = m / 4;
    for (uint64_t t = 0; t < q4; t++)
        for (int k = 0; k < 4; k++) l[k] += lut[(4 * t + k) * 256 + codes[4 * t + k]];
    float s = (l[0] + l[2]) + (l[1] + l[3]);
    for (uint64_t c = 4 * q4; c < m; c++) s += lut[c * 256 + codes[c]];
    return s;
}

QO_EXPORT void qo_pq_score_all(const uint8_t *rows, uint64_t m, const float *lut, uint64_t begin,
                               uint64_t end, int order, float *out) {
    for (uint64_t i = begin; i < end; i++)
        out[i - begin] = order == 2 ? qo_pq_score_sse_order(rows + i * m, m, lut)
                                    : qo_pq_score_simple(rows + i * m, m, lut);
}

/* encoded_vectors_pq.rs:566-593 score_internal */
QO_EXPORT float qo_pq_score_internal(const uint8_t *rows, uint64_t dim, uint64_t chunk_size,
                                     const float *centroids, int distance, int invert, uint64_t i,
                                     uint64_t j) {
    uint64_t m = qo_pq_chunks(dim, chunk_size);
    const uint8_t *ci = rows + i * m, *cj = rows + j * m;
    float s = 0.0f;
    for (uint64_t c = 0; c < m; c++) {
        uint64_t lo = c * chunk_size;
        uint64_t hi = lo + chunk_size < dim ? lo + chunk_size : dim;
        s += qo_distance(distance, centroids + (uint64_t)ci[c] * dim + lo,
                         centroids + (uint64_t)cj[c] * dim + lo, hi - lo);
    }
    return invert ? -s : s;
}

/* ------------------------------------------------------------------------ */
/* k-means: kmeans.rs:7-167.  data: [n][dim] (one chunk's sub-vectors of the sample).
 *   - centroids start as the first `centroids_count` rows (:25)
 *   - per iteration: update_indexes (:139-166: argmin of the sequential f32 sum of (a-b).powi(2),
 *     strict '<', f32::MAX start), then update_centroids (:49-137): `max_threads` workers take
 *     contiguous row ranges of n / max_threads rows, the last one the remainder (:77-82), each adds
 *     its rows into f64 accumulators in row order (:84-93), the partials are merged in worker order
 *     into a zeroed accumulator (:97-108), mean in f64 (:119-121), cast to f32 and the shift
 *     sum_f32 |old - new| over [centroid][j] in order (:125-135); stop when < accuracy (:136).
 *   - an EMPTY cluster takes a thread_rng row in the reference (:111-118) — not reproducible.
 *     Here: row (hash(chunk_index, centroid, iteration) % n) with the hash below (the product uses
 *     the same rule); *empties counts how often that happened, so a test can require 0.
 * trace (optional): [max_iterations][n] u32, the assignments of every iteration that ran. */
static uint32_t qo_reseed_hash(uint32_t chunk, uint32_t centroid, uint32_t iter) {
    uint32_t h = (chunk * 2654435761u) ^ (centroid * 40503u) ^ (iter * 2246822519u);
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    return h;
}

static void qo_update_indexes(const float *data, uint64_t n, uint64_t dim, const float *centroids,
                              uint64_t centroids_count, uint32_t *indexes) {
    for (uint64_t i = 0; i < n; i++) {
        const float *v = data + i * dim;
        float min_d = 3.40282347e+38f;
        uint32_t min_i = 0;
        for (uint64_t k = 0; k < centroids_count; k++) {
            const float *c = centroids + k * dim;
            float d = 0.0f;
            for (uint64_t j = 0; j < dim; j++) {
                float t = v[j] - c[j];
                d += t * t;
            }
            if (d < min_d) {
                min_d = d;
                min_i = (uint32_t)k;
            }
        }
        indexes[i] = min_i;
    }
}

QO_EXPORT int qo_kmeans(const float *data, uint64_t n, uint64_t dim, uint64_t centroids_count,
                        uint32_t max_iterations, uint32_t max_threads, float accuracy, uint32_t chunk_index,
                        float *centroids /* [centroids_count][dim] */, uint32_t *iterations, uint32_t *empties,
                        uint32_t *trace) {
    const uint64_t nc = centroids_count * dim;
    if (n < centroids_count || max_threads == 0) return -1;
    memcpy(centroids, data, nc * sizeof(float)); /* :25 */
    uint32_t *indexes = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    double *acc = (double *)malloc(nc * sizeof(double));
    double *part = (double *)malloc(nc * sizeof(double));
    uint64_t *cnt = (uint64_t *)malloc(centroids_count * sizeof(uint64_t));
    uint32_t it = 0, em = 0;
    for (; it < max_iterations; it++) {
        qo_update_indexes(data, n, dim, centroids, centroids_count, indexes);
        if (trace) memcpy(trace + (uint64_t)it * n, indexes, n * sizeof(uint32_t));
        /* update_centroids */
        for (uint64_t i = 0; i < nc; i++) acc[i] = 0.0;
        memset(cnt, 0, centroids_count * sizeof(uint64_t));
        const uint64_t per = n / max_threads; /* :77 */
        for (uint32_t w = 0; w < max_threads; w++) {
            const uint64_t lo = per * w, hi = (w + 1 == max_threads) ? n : per * (w + 1);
            for (uint64_t i = 0; i < nc; i++) part[i] = 0.0;
            for (uint64_t r = lo; r < hi; r++) {
                const uint32_t ci = indexes[r];
                cnt[ci] += 1;
                for (uint64_t j = 0; j < dim; j++) part[ci * dim + j] += (double)data[r * dim + j];
            }
            for (uint64_t i = 0; i < nc; i++) acc[i] += part[i]; /* :101-107 */
        }
        for (uint64_t k = 0; k < centroids_count; k++) {
            if (cnt[k] == 0) {
                const uint64_t row = qo_reseed_hash(chunk_index, (uint32_t)k, it) % n;
                for (uint64_t j = 0; j < dim; j++) acc[k * dim + j] = (double)data[row * dim + j];
                em++;
            } else {
                const double c = (double)cnt[k];
                for (uint64_t j = 0; j < dim; j++) acc[k * dim + j] /= c;
            }
        }
        float diff = 0.0f;
        for (uint64_t i = 0; i < nc; i++) {
            const float c_acc = (float)acc[i];
            diff += fabsf(centroids[i] - c_acc);
            centroids[i] = c_acc;
        }
        if (diff < accuracy) {
            it++;
            break;
        }
    }
    /* the reference runs update_indexes once more (:45); its result is dropped */
    if (iterations) *iterations = it;
    if (empties) *empties = em;
    free(indexes);
    free(acc);
    free(part);
    free(cnt);
    return 0;
}

/* encoded_vectors_pq.rs:278-342 find_centroids GIVEN the sampled row indices (the reference draws
 * them from a random Permutor, :300-302, and sorts them, :307).  count <= 256: the vectors
 * themselves (:290-297).  Otherwise, chunk by chunk: the sample's sub-vectors (:311-323) ->
 * kmeans(256 centroids, <= 100 iterations, 1e-5) (:325-333) -> written into centroid rows
 * [256][dim] at the chunk's columns (:336-338). */
QO_EXPORT int qo_find_centroids(const float *data, uint64_t count, uint64_t dim, uint64_t chunk_size,
                                const uint64_t *sample_rows, uint64_t sample_size, uint32_t max_threads,
                                float *centroids /* [256][dim] */, uint32_t *iterations /* [m] or NULL */,
                                uint32_t *empties /* total, or NULL */) {
    if (count <= 256) {
        qo_pq_centroids_small(data, count, dim, centroids);
        return 0;
    }
    const uint64_t m = qo_pq_chunks(dim, chunk_size);
    float *subset = (float *)malloc(sample_size * chunk_size * sizeof(float));
    float *cen = (float *)malloc(256 * chunk_size * sizeof(float));
    uint32_t em_total = 0;
    int rc = 0;
    for (uint64_t c = 0; c < m && rc == 0; c++) {
        const uint64_t lo = c * chunk_size, hi = lo + chunk_size < dim ? lo + chunk_size : dim, len = hi - lo;
        for (uint64_t s = 0; s < sample_size; s++)
            memcpy(subset + s * len, data + sample_rows[s] * dim + lo, len * sizeof(float));
        uint32_t it = 0, em = 0;
        rc = qo_kmeans(subset, sample_size, len, 256, 100, max_threads, 1e-5f, (uint32_t)c, cen, &it, &em, NULL);
        for (uint64_t k = 0; k < 256; k++) memcpy(centroids + k * dim + lo, cen + k * len, len * sizeof(float));
        if (iterations) iterations[c] = it;
        em_total += em;
    }
    if (empties) *empties = em_total;
    free(subset);
    free(cen);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* The caller's selection: demos/src/ann_benchmark_data.rs:151-167 keeps `k` (30) entries in a
 * BinaryHeap<Score> ordered by score only (Score::cmp, :29-33: partial_cmp().unwrap() -> a NaN
 * would panic): while the heap is full a new score replaces the top (the largest kept) iff
 * top.score > score (strict: an equal score never displaces), and into_sorted_vec() returns
 * them ascending.  So the kept multiset is the k smallest scores; among rows tied at the boundary
 * the EARLIER rows win, up to the heap's internal order.
 * BinaryHeap itself is Rust std (alloc::collections::binary_heap; the reference pins no toolchain):
 * push = sift_up with `hole <= parent -> stop`; PeekMut drop = sift_down_range(0, len) choosing
 * the greater child (right on ties) and stopping at `hole >= child`; into_sorted_vec = repeated
 * swap(0, end) + sift_down_range(0, end).  Restated below; the order it leaves among EQUAL
 * scores is an implementation detail no caller may rely on. */
typedef struct {
    uint32_t index;
    float score;
} qo_score;

static void qo_heap_sift_up(qo_score *d, size_t start, size_t pos) {
    qo_score hole = d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (hole.score <= d[parent].score) break;
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = hole;
}

static void qo_heap_sift_down_range(qo_score *d, size_t pos, size_t end) {
    qo_score hole = d[pos];
    size_t child = 2 * pos + 1;
    while (child <= (end >= 2 ? end - 2 : 0) && end >= 2) {
        child += (d[child].score <= d[child + 1].score) ? 1 : 0;
        if (hole.score >= d[child].score) {
            d[pos] = hole;
            return;
        }
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (end >= 1 && child == end - 1 && hole.score < d[child].score) {
        d[pos] = d[child];
        pos = child;
    }
    d[pos] = hole;
}

/* scores[i] = postprocess(score_point(q, i)); returns the number kept (min(k, n)). */
QO_EXPORT uint64_t qo_topk_heap(const float *scores, uint64_t n, uint64_t k, uint32_t *out_ids,
                                float *out_scores) {
    if (k == 0) return 0;
    qo_score *heap = (qo_score *)malloc(k * sizeof(qo_score));
    size_t len = 0;
    for (uint64_t i = 0; i < n; i++) {
        qo_score s = {(uint32_t)i, scores[i]};
        if (len == k) {
            if (heap[0].score > s.score) { /* :157-160 */
                heap[0] = s;
                qo_heap_sift_down_range(heap, 0, len);
            }
        } else {
            heap[len] = s; /* :162 */
            qo_heap_sift_up(heap, 0, len);
            len++;
        }
    }
    size_t end = len; /* into_sorted_vec (:165) */
    while (end > 1) {
        end--;
        qo_score t = heap[0];
        heap[0] = heap[end];
        heap[end] = t;
        qo_heap_sift_down_range(heap, 0, end);
    }
    for (size_t i = 0; i < len; i++) {
        out_ids[i] = heap[i].index;
        out_scores[i] = heap[i].score;
    }
    free(heap);
    return len;
}

/* ------------------------------------------------------------------------ */
/* Unquantised f32 metrics used by the reference's tests
 * (quantization/tests/metrics.rs:1-11). */
QO_EXPORT float qo_metric_f32(int distance, const float *a, const float *b, uint64_t n) {
    return qo_distance(distance, a, b, n);
}
