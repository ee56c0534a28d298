// Exact device top-k over an f32 score array (see topk.hip).
#pragma once
#include "common.hpp"

namespace qamd {

size_t topk_workspace_bytes(uint32_t k);

// Selects the k best rows of scores_dev[0..n): largest scores when `largest`, else the
// smallest; ties go to the lower index; output sorted best-first.  All pointers are device
// memory; only enqueues work on `stream`.
qamd_status topk_f32(const float *scores_dev, uint64_t n, uint32_t k, bool largest,
                     uint32_t *out_ids_dev, float *out_scores_dev, void *workspace_dev,
                     hipStream_t stream);

// Shared tail of the three *_topk entry points: scores already computed into scores_dev.
qamd_status topk_finish(const float *scores_dev, uint64_t n, uint32_t k, int largest,
                        uint32_t *out_ids, float *out_scores, qamd_mem out_mem, hipStream_t stream);

}  // namespace qamd
