/*
 * sharded_threads.c -- K search threads on ONE row-sharded handle, from plain C (pthreads): the
 * reference's score_* methods take `&self` and are called from many search threads at once
 * (quantization/src/encoded_vectors.rs:21-35), so the sharded handle must let them overlap.
 *
 *   build a store of ROWS x DIM scalar-u8 rows (reference row format, LCG bytes <= 127), adopt it as
 *   SHARDS logical shards on device 0 and as one plain handle;
 *   serial:      THREADS x CALLS searches (encode_query + topk(30), host in / host out) from one thread;
 *   concurrent:  the same searches from THREADS threads at once;
 *   every search must return exactly what the plain handle returns for that query.
 *
 * Prints "serial_us <t> concurrent_us <t> mismatches <n>"; tests/test_c_abi.py asserts on them.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "quantization_amd.h"

#define ROWS 24000u
#define DIM 64u
#define SHARDS 4u
#define THREADS 6u
#define CALLS 300u
#define K 30u

static uint32_t lcg_state = 777u;
static uint32_t lcg_next(void) {
    lcg_state = lcg_state * 1664525u + 1013904223u;
    return lcg_state >> 8;
}

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}

static qamd_u8_sharded *g_store;
static float g_queries[THREADS][DIM];
static uint32_t g_want_ids[THREADS][K];
static float g_want_scores[THREADS][K];
static int g_mismatches[THREADS];
static int g_failures[THREADS];

static void *search_thread(void *arg) {
    const unsigned t = (unsigned)(size_t)arg;
    qamd_u8_sharded_query *q = NULL;
    uint32_t ids[K];
    float scores[K];
    unsigned c;
    for (c = 0; c < CALLS; c++) {
        if (qamd_u8_sharded_encode_query(g_store, g_queries[t], DIM, QAMD_MEM_HOST, NULL, &q) != QAMD_OK ||
            qamd_u8_sharded_topk(g_store, q, K, 1, ids, scores, QAMD_MEM_HOST, NULL) != QAMD_OK) {
            g_failures[t]++;
            break;
        }
        if (memcmp(ids, g_want_ids[t], sizeof ids) != 0 || memcmp(scores, g_want_scores[t], sizeof scores) != 0)
            g_mismatches[t]++;
    }
    qamd_u8_sharded_query_free(q);
    qamd_thread_release();
    return NULL;
}

int main(void) {
    const size_t stride = DIM + 4;
    uint8_t *rows = (uint8_t *)malloc((size_t)ROWS * stride);
    qamd_u8_metadata meta;
    qamd_u8 *plain = NULL;
    qamd_u8_query *pq = NULL;
    int devices[SHARDS];
    pthread_t th[THREADS];
    unsigned i, j, t;
    double t0, serial_us, concurrent_us;
    int mismatches = 0, failures = 0;

    if (qamd_device_count() < 1) {
        printf("no device\n");
        return 0;
    }
    for (i = 0; i < ROWS; i++) {
        float off = (float)(lcg_next() & 0xFFFFu) / 4096.0f;
        memcpy(rows + i * stride, &off, 4);
        for (j = 0; j < DIM; j++) rows[i * stride + 4 + j] = (uint8_t)(lcg_next() & 127u);
    }
    memset(&meta, 0, sizeof meta);
    meta.actual_dim = DIM;
    meta.alpha = 1.0f / 127.0f;
    meta.offset = 0.0f;
    meta.multiplier = meta.alpha * meta.alpha;
    meta.vector_parameters.dim = DIM;
    meta.vector_parameters.count = ROWS;
    meta.vector_parameters.distance_type = QAMD_DOT;
    meta.vector_parameters.invert = 0;
    for (i = 0; i < SHARDS; i++) devices[i] = 0;
    if (qamd_u8_from_rows(rows, QAMD_MEM_HOST, &meta, NULL, &plain) != QAMD_OK ||
        qamd_u8_sharded_from_rows(rows, QAMD_MEM_HOST, &meta, devices, SHARDS, NULL, &g_store) != QAMD_OK) {
        fprintf(stderr, "set-up failed: %s\n", qamd_last_error());
        return 1;
    }
    for (t = 0; t < THREADS; t++) {
        for (j = 0; j < DIM; j++) g_queries[t][j] = (float)(lcg_next() & 0xFFFFu) / 65536.0f;
        if (qamd_u8_encode_query(plain, g_queries[t], DIM, QAMD_MEM_HOST, NULL, &pq) != QAMD_OK ||
            qamd_u8_topk(plain, pq, K, 1, g_want_ids[t], g_want_scores[t], QAMD_MEM_HOST, NULL) != QAMD_OK) {
            fprintf(stderr, "reference top-k failed: %s\n", qamd_last_error());
            return 1;
        }
    }
    /* warm-up: lanes, slots and per-thread workspaces come into being */
    for (t = 0; t < THREADS; t++) search_thread((void *)(size_t)t);
    memset(g_mismatches, 0, sizeof g_mismatches);

    t0 = now_us();
    for (t = 0; t < THREADS; t++) search_thread((void *)(size_t)t);
    serial_us = now_us() - t0;

    t0 = now_us();
    for (t = 0; t < THREADS; t++) pthread_create(&th[t], NULL, search_thread, (void *)(size_t)t);
    for (t = 0; t < THREADS; t++) pthread_join(th[t], NULL);
    concurrent_us = now_us() - t0;

    for (t = 0; t < THREADS; t++) {
        mismatches += g_mismatches[t];
        failures += g_failures[t];
    }
    printf("serial_us %.0f concurrent_us %.0f mismatches %d failures %d per_search_serial_us %.1f\n", serial_us,
           concurrent_us, mismatches, failures, serial_us / (double)(THREADS * CALLS));
    qamd_u8_query_free(pq);
    qamd_u8_sharded_free(g_store);
    qamd_u8_free(plain);
    free(rows);
    return (mismatches || failures) ? 2 : 0;
}
