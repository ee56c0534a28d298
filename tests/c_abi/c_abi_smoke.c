/*
 * c_abi_smoke.c -- the drop-in boundary exercised from plain C99 (no Python, no ctypes, no torch):
 * what a Rust `extern "C"` binding of include/quantization_amd.h would do.
 *
 *   encode (one-shot AND streaming) -> get_metadata -> export_rows (whole and in ranges) -> encode_query -> score_all
 *   -> topk -> sharded (2 logical shards) score_all / topk
 *
 * Inputs come from a fixed LCG (every value k/65536, exactly representable), so the pytest wrapper
 * (tests/test_c_abi.py) regenerates the same inputs, runs the ORACLE over them and compares the
 * bit patterns this program prints.  Build: gcc -std=c99 -pedantic -Wall -Iinclude c_abi_smoke.c
 * -Lquantization_amd -lquantization_amd.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "quantization_amd.h"

#define COUNT 3000u
#define DIM 72u /* not a multiple of 16: exercises the padding to 80 */
#define K 30u

static uint32_t lcg_state = 12345u;
static float lcg_next(void) {
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return (float)((lcg_state >> 8) & 0xFFFFu) / 65536.0f;
}

#define CHECK(expr)                                                                        \
    do {                                                                                   \
        qamd_status st_ = (expr);                                                          \
        if (st_ != QAMD_OK) {                                                              \
            fprintf(stderr, "FAILED %s: status %d: %s\n", #expr, (int)st_, qamd_last_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

static uint32_t bits_of(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

static int stop_never(void *user) {
    (void)user;
    return 0;
}

int main(int argc, char **argv) {
    int dry = argc > 1 && strcmp(argv[1], "--link-only") == 0;
    float *data = (float *)malloc(sizeof(float) * COUNT * DIM);
    float query[DIM];
    float *scores = (float *)malloc(sizeof(float) * COUNT);
    float *scores2 = (float *)malloc(sizeof(float) * COUNT);
    uint32_t ids[K], ids2[K];
    float top[K], top2[K];
    qamd_vector_parameters vp;
    qamd_u8_metadata meta, meta2;
    qamd_u8 *h = NULL, *hs = NULL;
    qamd_u8_query *q = NULL;
    qamd_u8_encoder *enc = NULL;
    qamd_u8_sharded *sh = NULL;
    qamd_u8_sharded_query *sq = NULL;
    uint8_t *rows, *rows2;
    uint64_t stride, i, off;
    int devices[2] = {0, 0};

    printf("version %s devices %d\n", qamd_version(), qamd_device_count());
    if (dry) return 0; /* proves the program links against every symbol it names */
    if (qamd_device_count() < 1) {
        fprintf(stderr, "no GPU: quantization_amd has no CPU fallback\n");
        return 2;
    }
    for (i = 0; i < (uint64_t)COUNT * DIM; i++) data[i] = lcg_next();
    for (i = 0; i < DIM; i++) query[i] = lcg_next();

    vp.dim = DIM;
    vp.count = COUNT;
    vp.distance_type = QAMD_DOT;
    vp.invert = 0;
    stride = qamd_u8_quantized_vector_size(&vp);
    rows = (uint8_t *)malloc(stride * COUNT);
    rows2 = (uint8_t *)malloc(stride * COUNT);

    /* one-shot encode */
    CHECK(qamd_u8_encode(data, QAMD_MEM_HOST, &vp, NULL, NULL, stop_never, NULL, NULL, &h));
    CHECK(qamd_u8_get_metadata(h, &meta));
    CHECK(qamd_u8_export_rows(h, rows, QAMD_MEM_HOST, NULL));

    /* streaming encode: two walks over the data in ragged batches */
    CHECK(qamd_u8_encoder_begin(&vp, NULL, NULL, stop_never, NULL, NULL, &enc));
    for (off = 0; off < COUNT; off += 701) {
        uint64_t n = COUNT - off < 701 ? COUNT - off : 701;
        CHECK(qamd_u8_encoder_observe(enc, data + off * DIM, n, QAMD_MEM_HOST));
    }
    for (off = 0; off < COUNT; off += 997) {
        uint64_t n = COUNT - off < 997 ? COUNT - off : 997;
        CHECK(qamd_u8_encoder_push(enc, data + off * DIM, n, QAMD_MEM_HOST));
    }
    CHECK(qamd_u8_encoder_finish(enc, &hs));
    CHECK(qamd_u8_get_metadata(hs, &meta2));
    /* the caller-owned-storage half of encode (storage_builder.push_vector_data, encoded_storage.rs:17-25):
     * the rows leave the handle in bounded ranges, here 512 at a time, and must be the whole export */
    memset(rows2, 0xEE, stride * COUNT);
    for (off = 0; off < COUNT; off += 512) {
        uint64_t n = COUNT - off < 512 ? COUNT - off : 512;
        CHECK(qamd_u8_export_rows_range(hs, off, n, rows2 + off * stride, QAMD_MEM_HOST, NULL));
    }
    if (qamd_u8_export_rows_range(hs, COUNT - 1, 2, rows2, QAMD_MEM_HOST, NULL) != QAMD_ERR_OUT_OF_RANGE) {
        fprintf(stderr, "a range past the end of the store must be refused\n");
        return 1;
    }
    if (memcmp(rows, rows2, stride * COUNT) != 0 || bits_of(meta.alpha) != bits_of(meta2.alpha) ||
        bits_of(meta.offset) != bits_of(meta2.offset) || bits_of(meta.multiplier) != bits_of(meta2.multiplier)) {
        fprintf(stderr, "streaming encode differs from the one-shot encode\n");
        return 1;
    }

    /* the rank-per-GPU route of encode: two holders of half the rows each find their own min / max
     * (quantile.rs:5-19), fold them (order-free), and encode their rows with the agreed interval
     * (alpha_offset_from_min_max, encoded_vectors_u8.rs:228-232): the two stores together are the one-shot store */
    {
        const uint64_t half = COUNT / 2 + 3; /* ragged on purpose */
        float mn0, mx0, mn1, mx1, ao[2];
        qamd_u8 *h0 = NULL, *h1 = NULL;
        qamd_vector_parameters vp0 = vp, vp1 = vp;
        int peer_state = -1;
        const char *why = NULL;
        CHECK(qamd_u8_find_min_max(data, QAMD_MEM_HOST, half, DIM, NULL, &mn0, &mx0));
        CHECK(qamd_u8_find_min_max(data + half * DIM, QAMD_MEM_HOST, COUNT - half, DIM, NULL, &mn1, &mx1));
        ao[1] = mn0 < mn1 ? mn0 : mn1;
        ao[0] = ((mx0 > mx1 ? mx0 : mx1) - ao[1]) / 127.0f;
        vp0.count = half;
        vp1.count = COUNT - half;
        CHECK(qamd_u8_encode(data, QAMD_MEM_HOST, &vp0, NULL, ao, stop_never, NULL, NULL, &h0));
        CHECK(qamd_u8_encode(data + half * DIM, QAMD_MEM_HOST, &vp1, NULL, ao, stop_never, NULL, NULL, &h1));
        CHECK(qamd_u8_export_rows(h0, rows2, QAMD_MEM_HOST, NULL));
        CHECK(qamd_u8_export_rows(h1, rows2 + half * stride, QAMD_MEM_HOST, NULL));
        CHECK(qamd_u8_get_metadata(h1, &meta2));
        if (memcmp(rows, rows2, stride * COUNT) != 0 || bits_of(meta.alpha) != bits_of(meta2.alpha) ||
            bits_of(meta.offset) != bits_of(meta2.offset) || bits_of(meta.multiplier) != bits_of(meta2.multiplier)) {
            fprintf(stderr, "two holders that agreed on the interval do not reproduce the one-shot encode\n");
            return 1;
        }
        qamd_u8_free(h0);
        qamd_u8_free(h1);
        /* how a logical shard reaches devices[0] is on record (peer access is never silently ignored) */
        CHECK(qamd_u8_sharded_from_rows(rows, QAMD_MEM_HOST, &meta, devices, 2, NULL, &sh));
        CHECK(qamd_u8_sharded_peer_access(sh, 1, &peer_state, &why));
        if (peer_state != QAMD_PEER_SAME_DEVICE || !why) {
            fprintf(stderr, "a logical shard of device 0 must report QAMD_PEER_SAME_DEVICE\n");
            return 1;
        }
        qamd_u8_sharded_free(sh);
        sh = NULL;
    }

    /* query, scan, selection */
    CHECK(qamd_u8_encode_query(h, query, DIM, QAMD_MEM_HOST, NULL, &q));
    CHECK(qamd_u8_score_all(h, q, scores, QAMD_MEM_HOST, NULL));
    CHECK(qamd_u8_topk(h, q, K, 1, ids, top, QAMD_MEM_HOST, NULL));

    /* the same store behind a sharded handle (two logical shards on device 0) */
    CHECK(qamd_u8_sharded_from_rows(rows, QAMD_MEM_HOST, &meta, devices, 2, NULL, &sh));
    CHECK(qamd_u8_sharded_encode_query(sh, query, DIM, QAMD_MEM_HOST, NULL, &sq));
    CHECK(qamd_u8_sharded_score_all(sh, sq, scores2, QAMD_MEM_HOST, NULL));
    CHECK(qamd_u8_sharded_topk(sh, sq, K, 1, ids2, top2, QAMD_MEM_HOST, NULL));
    if (memcmp(scores, scores2, sizeof(float) * COUNT) != 0 || memcmp(ids, ids2, sizeof ids) != 0 ||
        memcmp(top, top2, sizeof top) != 0) {
        fprintf(stderr, "sharded result differs from the single-handle result\n");
        return 1;
    }

    printf("meta %08x %08x %08x %llu\n", bits_of(meta.alpha), bits_of(meta.offset), bits_of(meta.multiplier),
           (unsigned long long)meta.actual_dim);
    {
        uint32_t sum = 0;
        for (i = 0; i < stride * COUNT; i++) sum = sum * 31u + rows[i];
        printf("rows %08x\n", sum);
        sum = 0;
        for (i = 0; i < COUNT; i++) sum = sum * 31u + bits_of(scores[i]);
        printf("scores %08x\n", sum);
    }
    for (i = 0; i < K; i++) printf("top %u %08x\n", ids[i], bits_of(top[i]));

    qamd_u8_sharded_query_free(sq);
    qamd_u8_sharded_free(sh);
    qamd_u8_query_free(q);
    qamd_u8_free(hs);
    qamd_u8_free(h);
    qamd_thread_release();
    free(rows2);
    free(rows);
    free(scores2);
    free(scores);
    free(data);
    printf("OK\n");
    return 0;
}
