/*
 * host_path.c -- developer timing, plain C against the C ABI: what a Rust caller pays per search on
 * a small store (host f32 query in, host ids + scores out), split by call.
 *   gcc -O2 -std=c99 -Iinclude tools/c/host_path.c -Lquantization_amd -lquantization_amd -o /tmp/host_path
 *   LD_LIBRARY_PATH=quantization_amd /tmp/host_path [rows] [dim]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "quantization_amd.h"

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
#define CHECK(e)                                                                       \
    do {                                                                               \
        qamd_status s_ = (e);                                                          \
        if (s_ != QAMD_OK) {                                                           \
            fprintf(stderr, "FAILED %s: %d %s\n", #e, (int)s_, qamd_last_error());     \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv) {
    uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 100000, dim = argc > 2 ? strtoull(argv[2], NULL, 10) : 768;
    float *data = (float *)malloc(sizeof(float) * n * dim), *query = (float *)malloc(sizeof(float) * dim);
    uint32_t seed = 1u, ids[30];
    float sc[30];
    qamd_vector_parameters vp;
    qamd_u8 *h = NULL;
    qamd_u8_query *q = NULL;
    uint64_t i;
    int it, reps = 3000;
    double t0, t_enc = 0, t_topk = 0, t_both;
    for (i = 0; i < n * dim; i++) {
        seed = seed * 1664525u + 1013904223u;
        data[i] = (float)((seed >> 8) & 0xFFFFu) / 65536.0f;
    }
    for (i = 0; i < dim; i++) query[i] = data[i * 7 % (n * dim)];
    vp.dim = dim;
    vp.count = n;
    vp.distance_type = QAMD_DOT;
    vp.invert = 0;
    CHECK(qamd_u8_encode(data, QAMD_MEM_HOST, &vp, NULL, NULL, NULL, NULL, NULL, &h));
    for (it = 0; it < 200; it++) {
        CHECK(qamd_u8_encode_query(h, query, dim, QAMD_MEM_HOST, NULL, &q));
        CHECK(qamd_u8_topk(h, q, 30, 1, ids, sc, QAMD_MEM_HOST, NULL));
    }
    for (it = 0; it < reps; it++) {
        t0 = now_us();
        CHECK(qamd_u8_encode_query(h, query, dim, QAMD_MEM_HOST, NULL, &q));
        t_enc += now_us() - t0;
        t0 = now_us();
        CHECK(qamd_u8_topk(h, q, 30, 1, ids, sc, QAMD_MEM_HOST, NULL));
        t_topk += now_us() - t0;
    }
    t0 = now_us();
    for (it = 0; it < reps; it++) {
        CHECK(qamd_u8_encode_query(h, query, dim, QAMD_MEM_HOST, NULL, &q));
        CHECK(qamd_u8_topk(h, q, 30, 1, ids, sc, QAMD_MEM_HOST, NULL));
    }
    t_both = (now_us() - t0) / reps;
    printf("rows %llu dim %llu: encode_query(host) returns after %.1f us, topk(30, host out) %.1f us; back to back %.1f us per search; best id %u\n",
           (unsigned long long)n, (unsigned long long)dim, t_enc / reps, t_topk / reps, t_both, ids[0]);
    qamd_u8_query_free(q);
    qamd_u8_free(h);
    free(data);
    free(query);
    return 0;
}
