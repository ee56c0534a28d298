// Handle layouts of the scalar-u8 quantizer, shared by u8.hip (single-query path) and
// u8_batch.hip (multi-query MFMA path).  Not part of the C ABI.
#pragma once
#include <atomic>
#include <mutex>

#include "common.hpp"
#include "lists.hpp"

struct qamd_u8 {
    int device = 0;
    qamd_u8_metadata meta{};
    uint64_t count = 0;
    uint64_t padded_rows = 0;  // round_up(count, 1024) + 1024 zero rows: tiles never need a load guard
    uint32_t row_chunks = 0;   // actual_dim / 16
    int lane_mode = 0;         // 0: integer sum rounded once; 1: avx2.c lane order
    qamd::DevBuf codes;        // [padded_rows][actual_dim]
    qamd::DevBuf offsets;      // [padded_rows] f32
    // The batched top-k's pivot sample (u8_batch.hip): rows hash(j) of the store, j < sample_rows, then 512
    // zero rows; gathered once on first use (count / 64 rows at most: 1.6 % of the store), immutable after.
    mutable std::mutex sample_mu;
    mutable qamd::DevBuf sample_codes, sample_offsets;
    mutable uint32_t sample_rows = 0;
};

struct qamd_u8_query {
    int device = 0;
    uint64_t actual_dim = 0;
    qamd::DevBuf buf;  // [0..4) offset f32, [16..16+actual_dim) codes
    mutable qamd::ReadyEvent ready;  // the last encode_query (stream order for consumers on other streams)
    // A HOST query of up to kFusedQueryDims values is not encoded by encode_query itself: its f32 values
    // wait here until the first consumer.  The single-launch top-k of a small store takes them BY VALUE in
    // its kernel arguments and quantises them in its prologue (one launch per search instead of two, no
    // PCIe read by the GPU); any other consumer runs the encode kernel first (u8.hip ensure_encoded).
    mutable std::vector<float> host_f32;
    float alpha = 0.0f, offset = 0.0f;  // of the store the query was (or will be) encoded for
    int distance = 0, invert = 0;
    mutable std::atomic<bool> deferred{false};
    mutable std::mutex encode_mu;  // two threads may consume one deferred query
    bool pooled = false;     // buf came from / goes back to the query buffer cache
    mutable std::atomic<bool> async_used{false};  // a consumer call only ENQUEUED (device outputs)
    ~qamd_u8_query() {
        if (pooled) qamd::query_buf_put(buf, !async_used.load(std::memory_order_relaxed) && ready.complete());
    }
};

// Pass-1 pieces used by the sharded encoder; defined in u8.hip.
namespace qamd {
qamd_status u8_minmax_range(const float *data, qamd_mem mem, uint64_t n_rows, uint64_t dim, hipStream_t s, float *mn,
                            float *mx);
qamd_status u8_quantile_interval(const float *data, qamd_mem mem, uint64_t count, uint64_t dim, float quantile,
                                 hipStream_t s, bool *found, float *mn, float *mx);
}  // namespace qamd

// encode_query for one query per wave (device-resident f32 queries); defined in u8.hip.
namespace qamd {
qamd_status u8_encode_queries_device(const qamd_u8 *h, const float *queries_dev, uint64_t n_queries, uint64_t qdim,
                                     uint8_t *codes_dev /* [n][code_pitch], actual_dim written */, uint64_t code_pitch,
                                     float *offsets_dev /* [n] */,
                                     hipStream_t stream);
bool u8_host_encode_is_lazy(uint64_t qdim);  // encode_query(host query of qdim values) launches nothing
// A host query the single-launch top-k quantises itself (qamd_u8_query::host_f32) and where its codes go.
struct FusedQuery {
    const float *values = nullptr;
    uint32_t qdim = 0;
    uint8_t *qbuf = nullptr;
};
qamd_status u8_topk_ptrs(const qamd_u8 *h, const uint8_t *codes_dev, const float *offset_dev, uint32_t k, int largest,
                         uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t stream,
                         const FusedQuery *fq = nullptr);
qamd_status u8_topk_batch_scans(const qamd_u8 *h, const uint8_t *codes_dev, uint64_t pitch, const float *offsets_dev,
                                uint32_t n_queries, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                                qamd_mem out_mem, hipStream_t stream);
uint32_t u8_multi_width(const qamd_u8 *h);
qamd_status u8_score_batch_scans(const qamd_u8 *h, const uint8_t *codes_dev, uint64_t pitch, const float *offsets_dev,
                                 uint32_t n_queries, float *out_dev, hipStream_t stream);
qamd_status u8_score_single(const qamd_u8 *h, const uint8_t *codes_dev, const float *offset_dev, float *out_dev,
                            hipStream_t stream);
qamd_status u8_score_lists(const qamd_u8 *h, const uint8_t *codes_dev, uint64_t pitch, const float *offsets_dev,
                           const ListArgs &a, hipStream_t stream);
qamd_status u8_topk_single(const qamd_u8 *h, const uint8_t *codes_dev, const float *offset_dev, uint32_t k,
                           int largest, uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t stream);
}  // namespace qamd
