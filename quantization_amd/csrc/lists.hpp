// Batched random-access scoring: many (query | stored row, id list) pairs in ONE launch.
//
// The reference scores one pair per call -- score_point(query, i) / score_internal(i, j)
// (quantization/src/encoded_vectors.rs:21-35) -- and its production caller, an HNSW index, makes
// those calls in bursts: a search hop scores the ~M neighbours of the current node for every
// in-flight query, graph construction scores a new row against candidate lists
// (encoded_vectors_u8.rs:386-453, encoded_vectors_pq.rs:566-593, encoded_vectors_binary.rs:302-314).
// One launch + one synchronisation per pair (15-19 us) makes that path unusable on a GPU, so the
// burst is the unit here: list l is ids[list_offsets[l] .. list_offsets[l + 1]) and is scored against
// query l of a query batch (score_ids_batch) or against stored row rows[l]
// (score_internal_ids_batch); out[p] is the score of pair p, in list order.  Every score is the
// single-pair call's, bit for bit.
//
// This header is the quantizer-independent part: marshalling of host / device lists and outputs
// (small host bursts ride in the calling thread's mapped scratch: no allocation, no copy call), and
// the device-side pair -> list lookup.
#pragma once

#include <algorithm>

#include "common.hpp"

namespace qamd {

// What a list kernel is launched with: everything in device-addressable memory.
struct ListArgs {
    const uint32_t *offsets = nullptr;  // [n_lists + 1], offsets[0] == 0
    const uint32_t *ids = nullptr;      // [n_pairs]
    const uint32_t *rows = nullptr;     // [n_lists] stored rows acting as queries (internal form), else nullptr
    float *out = nullptr;               // [n_pairs]
    uint32_t n_lists = 0;
    uint64_t n_pairs = 0;
};

#ifdef __HIPCC__
// List that pair p belongs to: the l with offsets[l] <= p < offsets[l + 1] (empty lists are skipped).
__device__ __forceinline__ uint32_t list_of_pair(const uint32_t *__restrict__ offsets, uint32_t n_lists, uint32_t p) {
    uint32_t lo = 0, hi = n_lists;  // invariant: offsets[lo] <= p < offsets[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= p) lo = mid;
        else hi = mid;
    }
    return lo;
}

// The kernels give every workgroup `pairs_per_block` consecutive pairs: ONE binary search per workgroup
// (its first pair), after which each lane group walks forward as its pair index grows -- a search per pair
// is ~log2(n_lists) dependent loads in front of every row fetch.
__device__ __forceinline__ uint32_t first_list_of_block(const uint32_t *__restrict__ offsets, uint32_t n_lists, uint64_t p0,
                                                        uint32_t *lds_word) {
    if (threadIdx.x == 0) *lds_word = list_of_pair(offsets, n_lists, (uint32_t)p0);
    __syncthreads();
    return *lds_word;
}
__device__ __forceinline__ uint32_t advance_list(const uint32_t *__restrict__ offsets, uint32_t n_lists, uint32_t l, uint64_t p) {
    while (l + 1 < n_lists && offsets[l + 1] <= (uint32_t)p) l++;
    return l;
}
#endif

// Pairs per workgroup for a burst of n pairs: one to four passes of the workgroup's `groups` lane groups.  Small bursts get
// one pass (every pair in flight at once: they are latency-bound); large ones at most four -- workgroups are dispatched in
// pair order, so the resident ones cover a compact window of the lists (a PQ burst re-reads each list's 96 KiB LUT from L2
// only while that window is a few dozen lists wide).
inline uint32_t pairs_per_block(uint64_t n, uint32_t groups, uint32_t max_passes = 4) {
    const uint64_t one_pass_blocks = (n + groups - 1) / groups, resident = (uint64_t)device_info().cu_count * 8;
    const uint64_t passes = std::min<uint64_t>(max_passes, std::max<uint64_t>(1, one_pass_blocks / resident));
    return (uint32_t)(groups * passes);
}

// Brings (list_offsets, ids, rows) and the output of one burst to the device, runs `launch(args)` on
// `s`, and delivers the scores.
//   lists_mem: where list_offsets, ids and rows live (one kind for all three); n_ids = list_offsets[n_lists].
//   Host lists are validated here (offsets monotone from 0, ids and rows < count: the reference panics
//   on the slice index, encoded_storage.rs:29 -> QAMD_ERR_OUT_OF_RANGE); device lists cannot be, the
//   kernels write NaN for an id or row that is out of range.
//   With device lists AND device output the call only enqueues.  Host output synchronises `s`.
template <class Launch>
qamd_status run_lists(const uint32_t *list_offsets, uint32_t n_lists, const uint32_t *ids, uint64_t n_ids,
                      const uint32_t *rows, qamd_mem lists_mem, float *out, qamd_mem out_mem, uint64_t count,
                      hipStream_t s, Launch &&launch) {
    if (n_lists == 0 || n_ids == 0) return QAMD_OK;
    if (!list_offsets || !ids || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (n_ids > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "at most 2^32 - 1 ids per burst");
    ListArgs a;
    a.n_lists = n_lists;
    a.n_pairs = n_ids;
    if (lists_mem == QAMD_MEM_HOST) {
        if (list_offsets[0] != 0) return fail(QAMD_ERR_ARGUMENTS, "list_offsets[0] must be 0");
        for (uint32_t l = 0; l < n_lists; l++)
            if (list_offsets[l + 1] < list_offsets[l]) return fail(QAMD_ERR_ARGUMENTS, "list_offsets must not decrease");
        if (list_offsets[n_lists] != n_ids)
            return fail(QAMD_ERR_ARGUMENTS, "list_offsets[n_lists] = %u, but n_ids = %llu", list_offsets[n_lists],
                        (unsigned long long)n_ids);
        for (uint64_t p = 0; p < a.n_pairs; p++)
            if (ids[p] >= count)
                return fail(QAMD_ERR_OUT_OF_RANGE, "row id %u out of range (count %llu)", ids[p], (unsigned long long)count);
        for (uint32_t l = 0; rows && l < n_lists; l++)
            if (rows[l] >= count)
                return fail(QAMD_ERR_OUT_OF_RANGE, "row id %u out of range (count %llu)", rows[l], (unsigned long long)count);
        const uint64_t words = (uint64_t)(n_lists + 1) + a.n_pairs + (rows ? n_lists : 0);
        const HostScratch hs = (out_mem == QAMD_MEM_HOST && words <= 1024 && a.n_pairs <= 1024) ? host_scratch() : HostScratch{};
        if (hs.host) {  // a small burst: lists in, scores out through the thread's mapped scratch
            uint32_t *w = hs.host;
            memcpy(w, list_offsets, (size_t)(n_lists + 1) * 4);
            memcpy(w + n_lists + 1, ids, (size_t)a.n_pairs * 4);
            if (rows) memcpy(w + n_lists + 1 + a.n_pairs, rows, (size_t)n_lists * 4);
            a.offsets = hs.dev;
            a.ids = hs.dev + n_lists + 1;
            a.rows = rows ? hs.dev + n_lists + 1 + a.n_pairs : nullptr;
            a.out = reinterpret_cast<float *>(hs.dev + 1024);
            QAMD_TRY(launch(a));
            QAMD_HIP(hipStreamSynchronize(s));
            memcpy(out, hs.host + 1024, (size_t)a.n_pairs * 4);
            return QAMD_OK;
        }
        StreamBuf in, res;
        QAMD_TRY(in.alloc(words * 4, s));
        uint32_t *d = in.as<uint32_t>();
        QAMD_HIP(hipMemcpyAsync(d, list_offsets, (size_t)(n_lists + 1) * 4, hipMemcpyHostToDevice, s));
        QAMD_HIP(hipMemcpyAsync(d + n_lists + 1, ids, (size_t)a.n_pairs * 4, hipMemcpyHostToDevice, s));
        if (rows) QAMD_HIP(hipMemcpyAsync(d + n_lists + 1 + a.n_pairs, rows, (size_t)n_lists * 4, hipMemcpyHostToDevice, s));
        a.offsets = d;
        a.ids = d + n_lists + 1;
        a.rows = rows ? d + n_lists + 1 + a.n_pairs : nullptr;
        a.out = out;
        if (out_mem == QAMD_MEM_HOST) {
            QAMD_TRY(res.alloc(a.n_pairs * 4, s));
            a.out = res.as<float>();
        }
        QAMD_TRY(launch(a));
        if (out_mem == QAMD_MEM_HOST) return copy_out(out, QAMD_MEM_HOST, a.out, a.n_pairs * 4, s);
        QAMD_HIP(hipStreamSynchronize(s));  // the pageable host lists must stay valid until they have been read
        return QAMD_OK;
    }
    // device lists (list_offsets[n_lists] == n_ids is the caller's promise): enqueue only
    if (out_mem == QAMD_MEM_HOST) return fail(QAMD_ERR_ARGUMENTS, "device lists need a device output buffer");
    a.offsets = list_offsets;
    a.ids = ids;
    a.rows = rows;
    a.out = out;
    return launch(a);
}

}  // namespace qamd
