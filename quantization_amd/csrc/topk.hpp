// Exact device top-k over an f32 score array (see topk.hip).
#pragma once
#include <functional>

#include "common.hpp"

namespace qamd {

size_t topk_workspace_bytes(uint32_t k);

// Selects the k best rows of scores_dev[0..n): largest scores when `largest`, else the
// smallest; ties go to the lower index; output sorted best-first.  All pointers are device
// memory; only enqueues work on `stream`.
qamd_status topk_f32(const float *scores_dev, uint64_t n, uint32_t k, bool largest,
                     uint32_t *out_ids_dev, float *out_scores_dev, void *workspace_dev,
                     hipStream_t stream);

// k-th smallest / largest value (k 1-based) of a device f32 array; synchronises `stream`.
qamd_status select_kth_f32(const float *vals_dev, uint64_t n, uint64_t k, bool largest, float *out_host,
                           hipStream_t stream);

// Shared tail of the three *_topk entry points: scores already computed into scores_dev.
qamd_status topk_finish(const float *scores_dev, uint64_t n, uint32_t k, int largest,
                        uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t stream);

// Fused scan + selection: the scan never materialises the score array.
//   1. score S pseudo-randomly chosen rows (random-access kernel; S = 16384 .. 131072, growing with n),
//   2. take the r-th best sample as pivot so that ~max(2048, 3k) rows are expected to pass,
//   3. run the scan in FILTER mode: rows at least as good as the pivot are appended to a
//      candidate buffer (8192 slots),
//   4. sort the candidates in one workgroup and emit the k best (ties to the lower id).
// Exact whenever k <= #candidates <= 8192; otherwise (heavy ties, e.g. binary scores, or an
// unlucky pivot) the classic path runs: full score array + exact radix select.  The status
// read-back makes this call synchronise `stream`.
constexpr uint32_t kTopkSample = 16384;
struct TopkFilter;
struct TopkFilterSlices;
struct FusedScan {
    std::function<qamd_status(const uint32_t *ids_dev, uint64_t n_ids, float *out_dev, hipStream_t)> score_ids;
    std::function<qamd_status(const TopkFilter &, hipStream_t)> scan_filter;
    std::function<qamd_status(float *scores_dev, hipStream_t)> scan_scores;
};
qamd_status fused_topk(uint64_t n, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                       qamd_mem out_mem, hipStream_t stream, const FusedScan &scan);

// Batched form of fused_topk for quantizers whose scan serves one query at a time (PQ: the LDS holds
// one LUT; binary): the Q per-query pipelines (sample -> pivot -> filtering scan -> sort) are all
// ENQUEUED back to back on per-query workspace slices, statuses are read back once per chunk of 32
// queries, and only the queries whose candidate list over/underflowed are redone by the exact
// classic path.  Results [Q][k]; ordering contract as fused_topk.
struct BatchScan {
    std::function<qamd_status(uint32_t q, const uint32_t *ids_dev, uint64_t n_ids, float *out_dev, hipStream_t)> score_ids;
    std::function<qamd_status(uint32_t q, const TopkFilter &, hipStream_t)> scan_filter;
    std::function<qamd_status(uint32_t q, float *scores_dev, hipStream_t)> scan_scores;
    // optional: the quantizer's single-launch top-k (stores <= 2M rows, k <= 64) with DEVICE outputs --
    // when set it serves every query (enqueue-only, no status word); returns false when not applicable
    std::function<bool(uint32_t q, uint32_t *ids_dev, float *scores_dev, hipStream_t, qamd_status &st)> topk_small;
    bool filter_capable = true;  // false: no FILTER-mode scan for this store -> classic path for every query
    // optional: ONE filtering scan for `nq` consecutive queries (their slices start at `slices.base`);
    // returns how many queries it took (8, 4, 2) or 0 when it has no kernel for this store / count
    std::function<uint32_t(uint32_t q0, uint32_t nq, const TopkFilterSlices &slices, hipStream_t, qamd_status &st)>
        scan_filter_multi;
};
qamd_status fused_topk_batch(uint64_t n, uint32_t n_queries, uint32_t k, int largest, uint32_t *out_ids,
                             float *out_scores, qamd_mem out_mem, hipStream_t stream, const BatchScan &scan);

// Single-launch top-k for small stores (count <= 2M rows, k <= 64): one kernel scans the rows,
// every wave keeps its own best 64 in registers (wave-level bitonic merges, topk_device.hpp), the
// workgroup folds its 16 waves and the last workgroup to finish folds the workgroups' lists and
// writes the result -- no score array, no sample pass, no status read-back.  With device outputs
// the call only ENQUEUES.
struct SmallTopk;
struct SmallTopkPlan {
    uint32_t workgroups = 0, rows_per_wg = 0;
};
// rows_per_tile: rows one wave takes per pass; min_rows_per_wg: below this share a further workgroup is not worth its
// start-up and its list in the final fold (callers: a few passes of the 16 waves, >= 128 KiB of row bytes).
bool small_topk_plan(uint64_t n, uint32_t k, uint32_t rows_per_tile, uint32_t min_rows_per_wg, SmallTopkPlan &plan);
qamd_status small_topk(const SmallTopkPlan &plan, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                       qamd_mem out_mem, hipStream_t stream,
                       const std::function<qamd_status(const SmallTopk &, hipStream_t)> &launch);

}  // namespace qamd
