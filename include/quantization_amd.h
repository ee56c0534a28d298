/*
 * quantization_amd.h — C ABI of the MI355X-native encode-and-score path.
 *
 * This is the drop-in boundary for qdrant/quantization's hot path.  The reference's own
 * FFI is per (query, vector) PAIR — 7 x86 symbols declared at
 *   quantization/src/encoded_vectors_u8.rs:476-483    impl_score_{dot,l1}_{avx,sse}
 *   quantization/src/encoded_vectors_binary.rs:317-324 impl_xor_popcnt_sse_uint{128,64,32}
 * — far too fine for a GPU (one launch per pair).  The boundary therefore moves up one
 * level to the Rust methods that call them: one opaque handle per encoded store, with the
 * methods of `trait EncodedVectors` (quantization/src/encoded_vectors.rs:21-35) plus the
 * batched form of the caller loop `for i in 0..n { score_point(q, i) }`
 * (demos/src/ann_benchmark.rs:247-252) as `*_score_all`.  Each entry point cites the
 * reference interface it replaces.  INTEGRATION.md shows the Rust `extern "C"` block and
 * the `impl EncodedVectors for ...` a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns a qamd_status.
 *   - `qamd_mem` says whether a caller buffer is host or device (HBM) memory.  Device
 *     buffers must belong to the handle's device.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls that take
 *     a stream only ENQUEUE when every buffer is device memory; host buffers make the call
 *     synchronous.
 *   - *_topk (all three quantizers) and *_topk_batch always synchronise `stream` before they
 *     return, device outputs included: the fused selection reads one status word back to decide
 *     whether the exact fallback has to run.  They cannot be captured into a hipGraph; score_all can.
 *   - every call runs on the handle's device and leaves the calling thread's current HIP device
 *     as it found it.
 *   - a call that only enqueues keeps using the calling thread's cached workspace until its work has
 *     run: the next call of that thread waits for it on the device (an event), whatever its stream.
 *     A hipGraph that captured such a call must not be replayed concurrently with other calls of the
 *     capturing thread.
 *   - handles own device memory; row bytes handed in stay caller-owned.
 *   - all score_* / topk calls are thread-safe on a shared handle (no interior mutation),
 *     matching `&self` in the reference.
 *   - no CPU fallback: every function fails with QAMD_ERR_DEVICE when no GPU is usable.
 */
#ifndef QUANTIZATION_AMD_H
#define QUANTIZATION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QAMD_API __attribute__((visibility("default")))

/* quantization/src/lib.rs:18-24 EncodingError{IOError,EncodingError,ArgumentsError,Stopped},
 * plus std::io::Error for save/load (-> QAMD_ERR_IO), the slice-index panic of
 * encoded_storage.rs:29 (-> QAMD_ERR_OUT_OF_RANGE) and HIP failures. */
typedef enum {
    QAMD_OK = 0,
    QAMD_ERR_IO = 1,
    QAMD_ERR_ENCODING = 2,
    QAMD_ERR_ARGUMENTS = 3,
    QAMD_ERR_STOPPED = 4,
    QAMD_ERR_OUT_OF_RANGE = 5,
    QAMD_ERR_DEVICE = 6
} qamd_status;

/* quantization/src/encoded_vectors.rs:6-11 */
typedef enum { QAMD_DOT = 0, QAMD_L1 = 1, QAMD_L2 = 2 } qamd_distance;

typedef enum { QAMD_MEM_HOST = 0, QAMD_MEM_DEVICE = 1 } qamd_mem;

/* quantization/src/encoded_vectors.rs:13-19 VectorParameters */
typedef struct {
    uint64_t dim;
    uint64_t count;
    int32_t distance_type; /* qamd_distance */
    int32_t invert;        /* bool */
} qamd_vector_parameters;

/* stop_condition: impl Fn() -> bool (encoded_vectors_u8.rs:39, _binary.rs:169, _pq.rs:62).
 * Polled between device batches; non-zero => the encode returns QAMD_ERR_STOPPED. */
typedef int (*qamd_stop_fn)(void *user);

/* Last error text of the calling thread (the String inside EncodingError / io::Error). */
QAMD_API const char *qamd_last_error(void);
QAMD_API const char *qamd_version(void);
/* Number of visible HIP devices (0 => nothing in this library can run). */
QAMD_API int qamd_device_count(void);
/* Device used by handles created afterwards on this thread (default 0). */
QAMD_API qamd_status qamd_set_device(int device);
QAMD_API int qamd_get_device(void);
/* Frees what the CALLING thread has cached on the GPUs (per-device workspaces of score_all /
 * topk with host outputs, the pinned result scratch).  They are also freed when the thread exits;
 * call this from a long-lived thread that is done querying. */
QAMD_API void qamd_thread_release(void);

/* ===================================================================================
 * Scalar u8 quantizer — quantization/src/encoded_vectors_u8.rs
 * =================================================================================== */
typedef struct qamd_u8 qamd_u8;             /* EncodedVectorsU8<TStorage>  (:14-17) */
typedef struct qamd_u8_query qamd_u8_query; /* EncodedQueryU8              (:19-22) */

/* Metadata (:24-31), serde field order. */
typedef struct {
    uint64_t actual_dim;
    float alpha;
    float offset;
    float multiplier;
    qamd_vector_parameters vector_parameters;
} qamd_u8_metadata;

/* get_quantized_vector_size (:252-255) and get_actual_dim (:257-259). */
QAMD_API uint64_t qamd_u8_quantized_vector_size(const qamd_vector_parameters *vp);
QAMD_API uint64_t qamd_u8_actual_dim(const qamd_vector_parameters *vp);

/* EncodedVectorsU8::encode (:34-140).  data: count*dim f32, row-major (the reference's
 * `orig_data` iterator flattened).  quantile: NULL = None.  For count > 100 000 the
 * reference draws a random sample (quantile.rs:31-34); here the sample is 100 000 evenly
 * strided rows — same statistic, not the same bits.
 * `alpha_offset` non-NULL overrides the interval search with {alpha, offset}
 * (conditional parity for that sampled case). */
QAMD_API qamd_status qamd_u8_encode(const float *data, qamd_mem data_mem,
                                    const qamd_vector_parameters *vp, const float *quantile,
                                    const float *alpha_offset, qamd_stop_fn stop, void *stop_user,
                                    void *stream, qamd_u8 **out);

/* Streaming form of encode: the reference takes a clonable ITERATOR and walks it twice
 * (encoded_vectors_u8.rs:34-40: pass 1 :57-71 min/max + quantile sample, pass 2 :73-118 quantize,
 * rows appended one by one through EncodedStorageBuilder::push_vector_data,
 * encoded_storage.rs:17-25), never holding the f32 data.  Same contract in bounded batches:
 *   begin(vp{count = rows that will come}, quantile | alpha_offset, stop)
 *   observe(batch) ...   every vector once, any batch sizes          (skip when alpha_offset given)
 *   push(batch) ...      every vector again, same order: rows are appended
 *   finish(&handle)      consumes the encoder; abort() drops it.
 * stop_condition is polled once per observe/push call (-> QAMD_ERR_STOPPED).  The result is
 * byte-identical to qamd_u8_encode on the concatenated data. */
typedef struct qamd_u8_encoder qamd_u8_encoder;
QAMD_API qamd_status qamd_u8_encoder_begin(const qamd_vector_parameters *vp, const float *quantile,
                                           const float *alpha_offset, qamd_stop_fn stop, void *stop_user,
                                           void *stream, qamd_u8_encoder **out);
QAMD_API qamd_status qamd_u8_encoder_observe(qamd_u8_encoder *e, const float *batch, uint64_t n_rows,
                                             qamd_mem batch_mem);
QAMD_API qamd_status qamd_u8_encoder_push(qamd_u8_encoder *e, const float *batch, uint64_t n_rows,
                                          qamd_mem batch_mem);
QAMD_API qamd_status qamd_u8_encoder_finish(qamd_u8_encoder *e, qamd_u8 **out);
QAMD_API void qamd_u8_encoder_abort(qamd_u8_encoder *e);

/* The two global statistics of encode's pass 1 ON THEIR OWN, for a host that holds the rows in several places (one
 * process per GPU, quantization_amd/sharded.py; the single-process qamd_u8_sharded_encode does the same inside).  The
 * reference finds ONE (alpha, offset) over all data before any row is quantised (encoded_vectors_u8.rs:57-71):
 *   - qamd_u8_find_min_max = find_min_max_from_iter (quantile.rs:5-19) over the caller's n_rows x dim values (NaN never
 *     wins a compare; no rows: (f32::MAX, f32::MIN), the reference's start values).  Every holder runs it on its rows,
 *     the holders fold the results with min / max - order-free, hence the bits of the single-handle encode - and hand
 *     alpha_offset_from_min_max (encoded_vectors_u8.rs:228-232: alpha = (max - min) / 127.0f, offset = min) to
 *     qamd_u8_encode / qamd_u8_encoder_begin as `alpha_offset`.
 *   - qamd_u8_find_quantile_interval = find_quantile_interval (quantile.rs:21-71) over `count` vectors; *found = 0 is the
 *     reference's None (keep the min / max interval).  Its sample is every vector for count <= 100 000 and the rows
 *     floor(k * count / 100 000), k < 100 000, beyond (the reference draws a random one there): a distributed host
 *     gathers exactly those rows, in k order, to one holder and calls this with count = the number gathered. */
QAMD_API qamd_status qamd_u8_find_min_max(const float *data, qamd_mem data_mem, uint64_t n_rows, uint64_t dim,
                                          void *stream, float *min, float *max);
QAMD_API qamd_status qamd_u8_find_quantile_interval(const float *data, qamd_mem data_mem, uint64_t count, uint64_t dim,
                                                    float quantile, void *stream, int *found, float *min, float *max);

/* Adopt rows already in the reference's storage format — what
 * EncodedStorage::get_vector_data serves (encoded_storage.rs:27-31): count rows of
 * [vector_offset f32 ne][actual_dim codes].  Used by load and by callers holding a store
 * encoded by the reference.  The rows are copied/re-laid-out into library-owned HBM. */
QAMD_API qamd_status qamd_u8_from_rows(const uint8_t *rows, qamd_mem rows_mem,
                                       const qamd_u8_metadata *meta, void *stream, qamd_u8 **out);

/* The inverse: write the reference-format row bytes (count * quantized_vector_size). */
QAMD_API qamd_status qamd_u8_export_rows(const qamd_u8 *h, uint8_t *rows, qamd_mem rows_mem,
                                         void *stream);
/* Rows [first_row, first_row + n_rows) only: the caller-owned-storage half of encode.  The reference's
 * encode pushes every row into the CALLER's store (storage_builder.push_vector_data,
 * encoded_vectors_u8.rs:34-40,117; encoded_storage.rs:17-25); a binding feeds its EncodedStorageBuilder
 * from bounded ranges (e.g. 64k rows) after encode / encoder_finish, so the store never has to exist
 * as one host buffer.  QAMD_ERR_OUT_OF_RANGE when the range leaves [0, count]. */
QAMD_API qamd_status qamd_u8_export_rows_range(const qamd_u8 *h, uint64_t first_row, uint64_t n_rows,
                                               uint8_t *rows, qamd_mem rows_mem, void *stream);
QAMD_API qamd_status qamd_u8_get_metadata(const qamd_u8 *h, qamd_u8_metadata *out);

/* EncodedVectors::save / load (:263-288): raw row file + serde_json metadata file. */
QAMD_API qamd_status qamd_u8_save(const qamd_u8 *h, const char *data_path, const char *meta_path);
QAMD_API qamd_status qamd_u8_load(const char *data_path, const char *meta_path,
                                  const qamd_vector_parameters *vp, qamd_u8 **out);

/* EncodedVectors::encode_query (:290-329).  *query is created when NULL and re-used
 * otherwise (no allocation on the hot loop).  qdim is `query.len()`. */
QAMD_API qamd_status qamd_u8_encode_query(const qamd_u8 *h, const float *query, uint64_t qdim,
                                          qamd_mem query_mem, void *stream, qamd_u8_query **query_io);
/* Inspect an encoded query: offset and the actual_dim code bytes (host buffers). */
QAMD_API qamd_status qamd_u8_query_read(const qamd_u8_query *q, float *offset, uint8_t *codes,
                                        uint64_t codes_capacity, uint64_t *codes_len);
QAMD_API void qamd_u8_query_free(qamd_u8_query *q);

/* EncodedVectors::score_point (:331-384) and score_internal (:386-453). */
QAMD_API qamd_status qamd_u8_score_point(const qamd_u8 *h, const qamd_u8_query *q, uint32_t i,
                                         float *out);
QAMD_API qamd_status qamd_u8_score_internal(const qamd_u8 *h, uint32_t i, uint32_t j, float *out);

/* score_internal (:386-453) for ONE stored row against many: out[k] = score_internal(i, ids[k]) -- what
 * graph construction asks (a new node against its candidate list), one launch instead of one per pair.
 * Host or device ids / outputs; with device ids and outputs the call only enqueues. */
QAMD_API qamd_status qamd_u8_score_internal_ids(const qamd_u8 *h, uint32_t i, const uint32_t *ids, uint64_t n_ids,
                                                qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream);
/* ... and for MANY stored rows, each against its own id list, in one launch: list l is
 * ids[list_offsets[l] .. list_offsets[l + 1]) (list_offsets: n_lists + 1 entries from 0, n_ids =
 * list_offsets[n_lists]); out[p] = score_internal(rows[l], ids[p]) for every p of list l.
 * `lists_mem` says where rows, list_offsets and ids live (one kind for the three).  Host lists are
 * validated (QAMD_ERR_OUT_OF_RANGE as score_internal) and bursts of up to ~1000 ids travel through the
 * calling thread's mapped scratch (one launch + one synchronisation, no allocation, no copy call);
 * device lists need a device output and only enqueue: an id or row out of range scores NaN. */
QAMD_API qamd_status qamd_u8_score_internal_ids_batch(const qamd_u8 *h, const uint32_t *rows,
                                                      const uint32_t *list_offsets, uint32_t n_lists,
                                                      const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem,
                                                      float *out, qamd_mem out_mem, void *stream);

/* NEW (batched caller loop, demos/src/ann_benchmark.rs:247-252):
 * out[i] = score_point(q, i) for i in [0, count). */
QAMD_API qamd_status qamd_u8_score_all(const qamd_u8 *h, const qamd_u8_query *q, float *out,
                                       qamd_mem out_mem, void *stream);
/* Random-access form (demos/benches/encode.rs "score random access"): out[k] = score_point(q, ids[k]). */
QAMD_API qamd_status qamd_u8_score_ids(const qamd_u8 *h, const qamd_u8_query *q,
                                       const uint32_t *ids, uint64_t n_ids, qamd_mem ids_mem,
                                       float *out, qamd_mem out_mem, void *stream);
/* Fused scan + selection (demos/src/ann_benchmark_data.rs:151-167 keeps the best 30 in a
 * heap).  largest != 0 keeps the k largest scores, else the k smallest; ties break to the
 * lower index; results are sorted best-first.  k <= 1024. */
QAMD_API qamd_status qamd_u8_topk(const qamd_u8 *h, const qamd_u8_query *q, uint32_t k,
                                  int largest, uint32_t *out_ids, float *out_scores,
                                  qamd_mem out_mem, void *stream);
QAMD_API void qamd_u8_free(qamd_u8 *h);

/* Multi-query form: many queries against the store at once (the caller's OUTER loop over
 * queries, demos/src/ann_benchmark.rs:245-260, with its per-query 30-entry heap,
 * demos/src/ann_benchmark_data.rs:151-167).  On the GPU this is a dense u8 x u8 -> i32
 * contraction on the matrix cores; every score is bit-identical to qamd_u8_score_all for the
 * same query.  Dot and L2 run the reference's dot kernel (encoded_vectors_u8.rs:339-341) as that
 * contraction; L1 (sum |q - v|, no matrix form) is served by the single-query kernel per query. */
typedef struct qamd_u8_query_batch qamd_u8_query_batch; /* n x EncodedQueryU8 */
/* queries: n_queries x qdim f32, row-major.  *batch_io is created when NULL, else re-used. */
QAMD_API qamd_status qamd_u8_encode_query_batch(const qamd_u8 *h, const float *queries,
                                                uint64_t n_queries, uint64_t qdim, qamd_mem queries_mem,
                                                void *stream, qamd_u8_query_batch **batch_io);
QAMD_API void qamd_u8_query_batch_free(qamd_u8_query_batch *b);
/* out[q * count + i] = score_point(query q, i): n_queries * count f32 (mind the size). */
QAMD_API qamd_status qamd_u8_score_batch(const qamd_u8 *h, const qamd_u8_query_batch *b, float *out,
                                         qamd_mem out_mem, void *stream);
/* score_point (:331-384) for many (query, id list) pairs in ONE launch -- one hop of every in-flight
 * HNSW search: list l (ids[list_offsets[l] .. list_offsets[l + 1])) is scored against query l of the
 * batch; out[p] = score_point(query l, ids[p]).  n_lists <= n_queries.  Buffers as
 * qamd_u8_score_internal_ids_batch. */
QAMD_API qamd_status qamd_u8_score_ids_batch(const qamd_u8 *h, const qamd_u8_query_batch *b,
                                             const uint32_t *list_offsets, uint32_t n_lists, const uint32_t *ids,
                                             uint64_t n_ids, qamd_mem lists_mem, float *out, qamd_mem out_mem,
                                             void *stream);
/* out_ids / out_scores: n_queries x k, per query as qamd_u8_topk.  Synchronises the stream. */
QAMD_API qamd_status qamd_u8_topk_batch(const qamd_u8 *h, const qamd_u8_query_batch *b, uint32_t k,
                                        int largest, uint32_t *out_ids, float *out_scores,
                                        qamd_mem out_mem, void *stream);

/* Extension (no reference counterpart): how a pair sum >= 2^24 (only possible for
 * actual_dim > 1040) becomes f32.  0 (default): exact integer, rounded once — what the
 * reference's scalar path does (encoded_vectors_u8.rs:158).  1: the 8-lane f32 summation
 * order of impl_score_dot_avx (quantization/cpp/avx2.c:41-62), bit for bit. */
QAMD_API qamd_status qamd_u8_set_lane_mode(qamd_u8 *h, int mode);

/* ===================================================================================
 * Binary quantizer — quantization/src/encoded_vectors_binary.rs
 * =================================================================================== */
typedef struct qamd_bin qamd_bin;             /* EncodedVectorsBin<TBitsStoreType,TStorage> (:11-15) */
typedef struct qamd_bin_query qamd_bin_query; /* EncodedBinVector (:17-19) */

typedef enum { QAMD_BITS_U8 = 0, QAMD_BITS_U128 = 1 } qamd_bits_store; /* BitsStoreType impls :44,:119 */

/* get_quantized_vector_size_from_params (:210-213), bytes per row. */
QAMD_API uint64_t qamd_bin_quantized_vector_size(const qamd_vector_parameters *vp,
                                                 qamd_bits_store store);
/* EncodedVectorsBin::encode (:165-191). */
QAMD_API qamd_status qamd_bin_encode(const float *data, qamd_mem data_mem,
                                     const qamd_vector_parameters *vp, qamd_bits_store store,
                                     qamd_stop_fn stop, void *stop_user, void *stream,
                                     qamd_bin **out);
/* Streaming form (EncodedVectorsBin::encode walks its iterator once, :165-191): begin, push
 * batches in row order, finish.  See qamd_u8_encoder_*. */
typedef struct qamd_bin_encoder qamd_bin_encoder;
QAMD_API qamd_status qamd_bin_encoder_begin(const qamd_vector_parameters *vp, qamd_bits_store store,
                                            qamd_stop_fn stop, void *stop_user, void *stream,
                                            qamd_bin_encoder **out);
QAMD_API qamd_status qamd_bin_encoder_push(qamd_bin_encoder *e, const float *batch, uint64_t n_rows,
                                           qamd_mem batch_mem);
QAMD_API qamd_status qamd_bin_encoder_finish(qamd_bin_encoder *e, qamd_bin **out);
QAMD_API void qamd_bin_encoder_abort(qamd_bin_encoder *e);
QAMD_API qamd_status qamd_bin_from_rows(const uint8_t *rows, qamd_mem rows_mem,
                                        const qamd_vector_parameters *vp, qamd_bits_store store,
                                        void *stream, qamd_bin **out);
QAMD_API qamd_status qamd_bin_export_rows(const qamd_bin *h, uint8_t *rows, qamd_mem rows_mem,
                                          void *stream);
/* Ranged form (storage_builder.push_vector_data, encoded_vectors_binary.rs:165-191 via
 * encoded_storage.rs:17-25); see qamd_u8_export_rows_range. */
QAMD_API qamd_status qamd_bin_export_rows_range(const qamd_bin *h, uint64_t first_row, uint64_t n_rows,
                                                uint8_t *rows, qamd_mem rows_mem, void *stream);
QAMD_API qamd_status qamd_bin_save(const qamd_bin *h, const char *data_path, const char *meta_path);
QAMD_API qamd_status qamd_bin_load(const char *data_path, const char *meta_path,
                                   const qamd_vector_parameters *vp, qamd_bits_store store,
                                   qamd_bin **out);
/* encode_query (:288-291). */
QAMD_API qamd_status qamd_bin_encode_query(const qamd_bin *h, const float *query, uint64_t qdim,
                                           qamd_mem query_mem, void *stream,
                                           qamd_bin_query **query_io);
QAMD_API qamd_status qamd_bin_query_read(const qamd_bin_query *q, uint8_t *bits,
                                         uint64_t capacity, uint64_t *len);
QAMD_API void qamd_bin_query_free(qamd_bin_query *q);
/* score_point (:293-300), score_internal (:302-314). */
QAMD_API qamd_status qamd_bin_score_point(const qamd_bin *h, const qamd_bin_query *q, uint32_t i,
                                          float *out);
QAMD_API qamd_status qamd_bin_score_internal(const qamd_bin *h, uint32_t i, uint32_t j, float *out);
/* Bursts of pairs in one launch (score_internal :302-314 is the metric of score_point on two stored
 * rows); arguments as qamd_u8_score_internal_ids / _ids_batch. */
QAMD_API qamd_status qamd_bin_score_internal_ids(const qamd_bin *h, uint32_t i, const uint32_t *ids, uint64_t n_ids,
                                                 qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_bin_score_internal_ids_batch(const qamd_bin *h, const uint32_t *rows,
                                                       const uint32_t *list_offsets, uint32_t n_lists,
                                                       const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem,
                                                       float *out, qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_bin_score_all(const qamd_bin *h, const qamd_bin_query *q, float *out,
                                        qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_bin_score_ids(const qamd_bin *h, const qamd_bin_query *q,
                                        const uint32_t *ids, uint64_t n_ids, qamd_mem ids_mem,
                                        float *out, qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_bin_topk(const qamd_bin *h, const qamd_bin_query *q, uint32_t k,
                                   int largest, uint32_t *out_ids, float *out_scores,
                                   qamd_mem out_mem, void *stream);
QAMD_API void qamd_bin_free(qamd_bin *h);

/* Many queries at once (the caller's outer loop, demos/src/ann_benchmark.rs:245-260; BASELINE config
 * 3/4 shape).  score_batch reads every row once for up to 8 queries; topk_batch runs the per-query
 * fused selections back to back with one status read-back per 32 queries.  Every score and list is
 * bit-identical to the single-query calls.  topk_batch synchronises the stream. */
typedef struct qamd_bin_query_batch qamd_bin_query_batch; /* n x EncodedBinVector */
QAMD_API qamd_status qamd_bin_encode_query_batch(const qamd_bin *h, const float *queries, uint64_t n_queries,
                                                 uint64_t qdim, qamd_mem queries_mem, void *stream,
                                                 qamd_bin_query_batch **batch_io);
QAMD_API void qamd_bin_query_batch_free(qamd_bin_query_batch *b);
/* out[q * count + i] = score_point(query q, i). */
QAMD_API qamd_status qamd_bin_score_batch(const qamd_bin *h, const qamd_bin_query_batch *b, float *out,
                                          qamd_mem out_mem, void *stream);
/* score_point for many (query, id list) pairs in one launch; as qamd_u8_score_ids_batch. */
QAMD_API qamd_status qamd_bin_score_ids_batch(const qamd_bin *h, const qamd_bin_query_batch *b,
                                              const uint32_t *list_offsets, uint32_t n_lists, const uint32_t *ids,
                                              uint64_t n_ids, qamd_mem lists_mem, float *out, qamd_mem out_mem,
                                              void *stream);
QAMD_API qamd_status qamd_bin_topk_batch(const qamd_bin *h, const qamd_bin_query_batch *b, uint32_t k,
                                         int largest, uint32_t *out_ids, float *out_scores,
                                         qamd_mem out_mem, void *stream);

/* ===================================================================================
 * Product quantizer — quantization/src/encoded_vectors_pq.rs
 * =================================================================================== */
typedef struct qamd_pq qamd_pq;             /* EncodedVectorsPQ<TStorage> (:27-30) */
typedef struct qamd_pq_query qamd_pq_query; /* EncodedQueryPQ{lut}        (:35-37) */

#define QAMD_PQ_CENTROIDS 256 /* CENTROIDS_COUNT (:25) */

/* get_quantized_vector_size (:109-114) == number of chunks. */
QAMD_API uint64_t qamd_pq_quantized_vector_size(const qamd_vector_parameters *vp,
                                                uint64_t chunk_size);
/* EncodedVectorsPQ::encode (:56-107).  centroids: NULL => find_centroids (:278-342)
 * runs (count <= 256: the vectors themselves, exactly as :290-297; otherwise k-means on a
 * 10 000-row sample, kmeans.rs).  The reference picks the sample rows at random (:300-302) and
 * re-seeds an empty cluster from thread_rng (kmeans.rs:111-118); here the sample is the evenly
 * strided rows floor(k * count / S) and the re-seed a fixed hash.  Everything else -- assignment,
 * f64 sums split over `max_kmeans_threads` contiguous row ranges and merged in worker order
 * (kmeans.rs:77-107), f32 shift sum, stopping rule -- is the reference's arithmetic in the
 * reference's order: given the same sample rows and no empty cluster the centroids are
 * bit-identical.  Non-NULL centroids: 256 x dim f32, centroid-major (Metadata.centroids,
 * :39-44), host memory; encode_storage (:136-226) then runs with them. */
QAMD_API qamd_status qamd_pq_encode(const float *data, qamd_mem data_mem,
                                    const qamd_vector_parameters *vp, uint64_t chunk_size,
                                    const float *centroids, uint32_t max_kmeans_threads,
                                    qamd_stop_fn stop, void *stop_user, void *stream,
                                    qamd_pq **out);
/* How the last k-means went: iterations of the slowest chunk (0: centroids were given or
 * count <= 256) and the number of empty-cluster re-seeds (0 => centroid parity holds, see above). */
QAMD_API qamd_status qamd_pq_kmeans_info(const qamd_pq *h, uint32_t *iterations, uint32_t *empty_clusters);
/* Which kernel the whole-store scan (score_all / topk; the caller loop of demos/src/ann_benchmark.rs:245-252 over
 * score_point, encoded_vectors_pq.rs:549-561) takes for THIS store, as a static string - "pq_scan_skew_kernel" (rows of
 * 16 .. 128 chunks with m % 4 == 0), "pq_scan_skew_kernel<SLICED>" (longer rows with m % 32 == 0: one launch per LUT slice of
 * the store's planar scan image), "pq_scan_fast_kernel" (other row lengths), "pq_scan_kernel" (fewer than 4096 rows) - and
 * in *n_launches how many launches one scan is.  For measurement harnesses: bench.py names the kernel it times and picks
 * its roofline bound from this instead of re-deriving the library's dispatch. */
QAMD_API const char *qamd_pq_scan_kernel(const qamd_pq *h, uint32_t *n_launches);
/* Streaming form (the reference walks its iterator twice: find_centroids :278-342, then
 * encode_storage :136-226): begin, observe every vector once (skipped when centroids are given),
 * push every vector again in the same order, finish.  See qamd_u8_encoder_*. */
typedef struct qamd_pq_encoder qamd_pq_encoder;
QAMD_API qamd_status qamd_pq_encoder_begin(const qamd_vector_parameters *vp, uint64_t chunk_size,
                                           const float *centroids, uint32_t max_kmeans_threads,
                                           qamd_stop_fn stop, void *stop_user, void *stream,
                                           qamd_pq_encoder **out);
QAMD_API qamd_status qamd_pq_encoder_observe(qamd_pq_encoder *e, const float *batch, uint64_t n_rows,
                                             qamd_mem batch_mem);
QAMD_API qamd_status qamd_pq_encoder_push(qamd_pq_encoder *e, const float *batch, uint64_t n_rows,
                                          qamd_mem batch_mem);
QAMD_API qamd_status qamd_pq_encoder_finish(qamd_pq_encoder *e, qamd_pq **out);
QAMD_API void qamd_pq_encoder_abort(qamd_pq_encoder *e);

/* find_centroids (encoded_vectors_pq.rs:278-342) ON ITS OWN: the 256 x dim centroids qamd_pq_encode would train for
 * `data` (vp->count rows), written to host memory `centroids`.  For a host that holds the rows in several places: the
 * k-means sample is the rows floor(k * count / S), k < S = min(10 000, count) (the reference draws a random Permutor
 * sample, :300-307); the holders gather exactly those rows, in k order, to one of them, which calls this with
 * vp->count = S and broadcasts the result; every holder then encodes its rows with `centroids` given.  count <= 256:
 * the vectors themselves, zero-filled (:290-297).  iterations / empty_clusters as qamd_pq_kmeans_info (may be NULL). */
QAMD_API qamd_status qamd_pq_find_centroids(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                                            uint64_t chunk_size, uint32_t max_kmeans_threads, qamd_stop_fn stop,
                                            void *stop_user, void *stream, float *centroids, uint32_t *iterations,
                                            uint32_t *empty_clusters);
QAMD_API qamd_status qamd_pq_from_rows(const uint8_t *rows, qamd_mem rows_mem,
                                       const qamd_vector_parameters *vp, uint64_t chunk_size,
                                       const float *centroids, void *stream, qamd_pq **out);
QAMD_API qamd_status qamd_pq_export_rows(const qamd_pq *h, uint8_t *rows, qamd_mem rows_mem,
                                         void *stream);
/* Ranged form (storage_builder.push_vector_data, encoded_vectors_pq.rs:136-226 via
 * encoded_storage.rs:17-25); see qamd_u8_export_rows_range. */
QAMD_API qamd_status qamd_pq_export_rows_range(const qamd_pq *h, uint64_t first_row, uint64_t n_rows,
                                               uint8_t *rows, qamd_mem rows_mem, void *stream);
/* Metadata.centroids (256 x dim f32, host). */
QAMD_API qamd_status qamd_pq_get_centroids(const qamd_pq *h, float *centroids);
QAMD_API qamd_status qamd_pq_save(const qamd_pq *h, const char *data_path, const char *meta_path);
QAMD_API qamd_status qamd_pq_load(const char *data_path, const char *meta_path,
                                  const qamd_vector_parameters *vp, qamd_pq **out);
/* encode_query (:525-547): builds the chunk-major LUT (what qamd_pq_query_read returns: EncodedQueryPQ.lut).  For stores
 * whose whole-store scan runs without LDS bank conflicts (m % 32 == 0) the object also keeps a [code][chunk] copy. */
QAMD_API qamd_status qamd_pq_encode_query(const qamd_pq *h, const float *query, uint64_t qdim,
                                          qamd_mem query_mem, void *stream,
                                          qamd_pq_query **query_io);
QAMD_API qamd_status qamd_pq_query_read(const qamd_pq_query *q, float *lut, uint64_t capacity,
                                        uint64_t *len);
QAMD_API void qamd_pq_query_free(qamd_pq_query *q);
/* score_point (:549-561 -> score_point_sse :405-440 summation order), score_internal (:566-593). */
QAMD_API qamd_status qamd_pq_score_point(const qamd_pq *h, const qamd_pq_query *q, uint32_t i,
                                         float *out);
QAMD_API qamd_status qamd_pq_score_internal(const qamd_pq *h, uint32_t i, uint32_t j, float *out);
/* Bursts of pairs in one launch (score_internal :566-593: both rows decoded to centroid sub-vectors, the
 * chunk metrics summed in chunk order); arguments as qamd_u8_score_internal_ids / _ids_batch. */
QAMD_API qamd_status qamd_pq_score_internal_ids(const qamd_pq *h, uint32_t i, const uint32_t *ids, uint64_t n_ids,
                                                qamd_mem ids_mem, float *out, qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_pq_score_internal_ids_batch(const qamd_pq *h, const uint32_t *rows,
                                                      const uint32_t *list_offsets, uint32_t n_lists,
                                                      const uint32_t *ids, uint64_t n_ids, qamd_mem lists_mem,
                                                      float *out, qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_pq_score_all(const qamd_pq *h, const qamd_pq_query *q, float *out,
                                       qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_pq_score_ids(const qamd_pq *h, const qamd_pq_query *q,
                                       const uint32_t *ids, uint64_t n_ids, qamd_mem ids_mem,
                                       float *out, qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_pq_topk(const qamd_pq *h, const qamd_pq_query *q, uint32_t k,
                                  int largest, uint32_t *out_ids, float *out_scores,
                                  qamd_mem out_mem, void *stream);
QAMD_API void qamd_pq_free(qamd_pq *h);

/* Many queries at once (BASELINE config 4's PQ leg): all LUTs are built by one launch, the per-query
 * LDS-LUT scans are enqueued back to back (the scan is LDS-gather-bound, one LUT fills the LDS, so
 * queries cannot share a pass); topk_batch reads statuses back once per 32 queries and synchronises
 * the stream.  Bit-identical to the single-query calls. */
typedef struct qamd_pq_query_batch qamd_pq_query_batch; /* n x EncodedQueryPQ */
QAMD_API qamd_status qamd_pq_encode_query_batch(const qamd_pq *h, const float *queries, uint64_t n_queries,
                                                uint64_t qdim, qamd_mem queries_mem, void *stream,
                                                qamd_pq_query_batch **batch_io);
QAMD_API void qamd_pq_query_batch_free(qamd_pq_query_batch *b);
QAMD_API qamd_status qamd_pq_score_batch(const qamd_pq *h, const qamd_pq_query_batch *b, float *out,
                                         qamd_mem out_mem, void *stream);
/* score_point for many (query, id list) pairs in one launch (LUT l for list l); as qamd_u8_score_ids_batch. */
QAMD_API qamd_status qamd_pq_score_ids_batch(const qamd_pq *h, const qamd_pq_query_batch *b,
                                             const uint32_t *list_offsets, uint32_t n_lists, const uint32_t *ids,
                                             uint64_t n_ids, qamd_mem lists_mem, float *out, qamd_mem out_mem,
                                             void *stream);
QAMD_API qamd_status qamd_pq_topk_batch(const qamd_pq *h, const qamd_pq_query_batch *b, uint32_t k,
                                        int largest, uint32_t *out_ids, float *out_scores,
                                        qamd_mem out_mem, void *stream);

/* ===================================================================================
 * Row-sharded stores: ONE process, several GPUs of a node behind one handle.
 *
 * No reference counterpart (the crate is single-device); the caller it serves is the one
 * process of demos/src/ann_benchmark.rs:245-260 (score every row, keep the best 30:
 * ann_benchmark_data.rs:151-167).  Shard g of G owns rows [g*N/G, (g+1)*N/G) as an ordinary
 * handle on devices[g]; global row id = shard base + local id; metadata is replicated.
 * `devices` may repeat a device (logical shards on one GPU).  Every call is synchronous: it
 * posts one job per shard (served by worker threads bound to the shard's device, each with its own
 * stream), waits for ITS jobs, and does ONE exchange -- score_all: each shard's scores land in their
 * slice of `out` (host: one D2H per GPU; device: peer copy of 4 B/row over xGMI to the device owning
 * `out`); topk: G*k (id, score) pairs are peer-copied to devices[0] and merged by one kernel there
 * with the single-handle ordering (best first, ties to the lower global id).  Results are
 * bit-identical to the same call on a single handle holding all rows.  Host buffers or device
 * buffers; a device output of topk must live on devices[0].
 * Threads: encode_query / score_all / topk / topk_batch may be called from any number of threads on
 * one sharded handle at once, like the `&self` methods they stand for (encoded_vectors.rs:21-35):
 * there is no per-handle lock, the callers' jobs interleave on the shards' queues and each call
 * leases its own exchange buffers.
 * `stream`: the hipStream_t that produced the call's device INPUT buffers / still uses the buffers
 * its device OUTPUTS overwrite (NULL = the null stream of the device owning the buffer).  It is
 * synchronised on entry when a buffer of the call is device memory (the workers run on their own
 * streams, which nothing else orders against the caller's); ignored for host buffers.  At return
 * every output is complete.
 * Deployment knob: the environment variable QAMD_SHARD_LANES (1..8, default 3) is the number of worker threads
 * ("lanes") per shard, read once per process: while one lane waits for its kernel the others enqueue.  It is the
 * only environment variable the product library reads.
 * Peer access: the exchanges between a shard's device and devices[0] are device-to-device copies.  At construction
 * the handle tries hipDeviceEnablePeerAccess in both directions for every such pair and RECORDS the outcome
 * (qamd_*_sharded_peer_access); where it is unavailable or fails the same hipMemcpyAsync calls still work, staged
 * through host memory by the runtime - slower, never wrong.
 * =================================================================================== */
/* How shard g's device reaches devices[0] (qamd_*_sharded_peer_access): */
typedef enum {
    QAMD_PEER_SAME_DEVICE = 0, /* the shard lives on devices[0] itself (or is a logical shard of it) */
    QAMD_PEER_ENABLED = 1,     /* direct peer copies (xGMI) in both directions */
    QAMD_PEER_UNAVAILABLE = 2, /* hipDeviceCanAccessPeer said no: copies are staged by the runtime */
    QAMD_PEER_FAILED = 3       /* hipDeviceEnablePeerAccess failed (*reason = the HIP error): staged copies */
} qamd_peer_state;
typedef struct qamd_u8_sharded qamd_u8_sharded;
typedef struct qamd_u8_sharded_query qamd_u8_sharded_query;
typedef struct qamd_u8_sharded_query_batch qamd_u8_sharded_query_batch;
/* encode (encoded_vectors_u8.rs:34-140): global min/max (or quantile interval) first, then
 * every shard quantizes its own rows.  `data` is host memory or device memory of any one GPU. */
QAMD_API qamd_status qamd_u8_sharded_encode(const float *data, qamd_mem data_mem,
                                            const qamd_vector_parameters *vp, const float *quantile,
                                            const float *alpha_offset, qamd_stop_fn stop, void *stop_user,
                                            const int *devices, uint32_t n_shards, void *stream,
                                             qamd_u8_sharded **out);
QAMD_API qamd_status qamd_u8_sharded_from_rows(const uint8_t *rows, qamd_mem rows_mem,
                                               const qamd_u8_metadata *meta, const int *devices,
                                               uint32_t n_shards, void *stream,
                                             qamd_u8_sharded **out);
QAMD_API uint32_t qamd_u8_sharded_shard_count(const qamd_u8_sharded *h);
/* *state = a qamd_peer_state for shard g; *reason (may be NULL) = a static or handle-owned string saying why. */
QAMD_API qamd_status qamd_u8_sharded_peer_access(const qamd_u8_sharded *h, uint32_t g, int *state, const char **reason);
/* Borrow shard g (owned by the sharded handle): its single-device handle, first global row, device. */
QAMD_API qamd_status qamd_u8_sharded_shard(const qamd_u8_sharded *h, uint32_t g, const qamd_u8 **shard,
                                           uint64_t *row_begin, int *device);
QAMD_API qamd_status qamd_u8_sharded_get_metadata(const qamd_u8_sharded *h, qamd_u8_metadata *out);
QAMD_API qamd_status qamd_u8_sharded_encode_query(qamd_u8_sharded *h, const float *query, uint64_t qdim,
                                                  qamd_mem query_mem, void *stream,
                                             qamd_u8_sharded_query **query_io);
QAMD_API void qamd_u8_sharded_query_free(qamd_u8_sharded_query *q);
QAMD_API qamd_status qamd_u8_sharded_score_all(qamd_u8_sharded *h, const qamd_u8_sharded_query *q, float *out,
                                               qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_u8_sharded_topk(qamd_u8_sharded *h, const qamd_u8_sharded_query *q, uint32_t k,
                                          int largest, uint32_t *out_ids, float *out_scores,
                                          qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_u8_sharded_encode_query_batch(qamd_u8_sharded *h, const float *queries,
                                                        uint64_t n_queries, uint64_t qdim,
                                                        qamd_mem queries_mem, void *stream,
                                             qamd_u8_sharded_query_batch **batch_io);
QAMD_API void qamd_u8_sharded_query_batch_free(qamd_u8_sharded_query_batch *b);
/* n_queries x k; shards x k <= 8192. */
QAMD_API qamd_status qamd_u8_sharded_topk_batch(qamd_u8_sharded *h, const qamd_u8_sharded_query_batch *b,
                                                uint32_t k, int largest, uint32_t *out_ids,
                                                float *out_scores, qamd_mem out_mem, void *stream);
QAMD_API void qamd_u8_sharded_free(qamd_u8_sharded *h);

typedef struct qamd_bin_sharded qamd_bin_sharded;
typedef struct qamd_bin_sharded_query qamd_bin_sharded_query;
QAMD_API qamd_status qamd_bin_sharded_encode(const float *data, qamd_mem data_mem,
                                             const qamd_vector_parameters *vp, qamd_bits_store store,
                                             qamd_stop_fn stop, void *stop_user, const int *devices,
                                             uint32_t n_shards, void *stream,
                                             qamd_bin_sharded **out);
QAMD_API qamd_status qamd_bin_sharded_from_rows(const uint8_t *rows, qamd_mem rows_mem,
                                                const qamd_vector_parameters *vp, qamd_bits_store store,
                                                const int *devices, uint32_t n_shards, void *stream,
                                             qamd_bin_sharded **out);
QAMD_API uint32_t qamd_bin_sharded_shard_count(const qamd_bin_sharded *h);
/* *state = a qamd_peer_state for shard g; *reason (may be NULL) = a static or handle-owned string saying why. */
QAMD_API qamd_status qamd_bin_sharded_peer_access(const qamd_bin_sharded *h, uint32_t g, int *state, const char **reason);
QAMD_API qamd_status qamd_bin_sharded_shard(const qamd_bin_sharded *h, uint32_t g, const qamd_bin **shard,
                                            uint64_t *row_begin, int *device);
QAMD_API qamd_status qamd_bin_sharded_encode_query(qamd_bin_sharded *h, const float *query, uint64_t qdim,
                                                   qamd_mem query_mem, void *stream,
                                             qamd_bin_sharded_query **query_io);
QAMD_API void qamd_bin_sharded_query_free(qamd_bin_sharded_query *q);
QAMD_API qamd_status qamd_bin_sharded_score_all(qamd_bin_sharded *h, const qamd_bin_sharded_query *q,
                                                float *out, qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_bin_sharded_topk(qamd_bin_sharded *h, const qamd_bin_sharded_query *q, uint32_t k,
                                           int largest, uint32_t *out_ids, float *out_scores,
                                           qamd_mem out_mem, void *stream);
typedef struct qamd_bin_sharded_query_batch qamd_bin_sharded_query_batch;
QAMD_API qamd_status qamd_bin_sharded_encode_query_batch(qamd_bin_sharded *h, const float *queries,
                                                         uint64_t n_queries, uint64_t qdim,
                                                         qamd_mem queries_mem, void *stream,
                                             qamd_bin_sharded_query_batch **batch_io);
QAMD_API void qamd_bin_sharded_query_batch_free(qamd_bin_sharded_query_batch *b);
QAMD_API qamd_status qamd_bin_sharded_topk_batch(qamd_bin_sharded *h, const qamd_bin_sharded_query_batch *b,
                                                 uint32_t k, int largest, uint32_t *out_ids,
                                                 float *out_scores, qamd_mem out_mem, void *stream);
QAMD_API void qamd_bin_sharded_free(qamd_bin_sharded *h);

typedef struct qamd_pq_sharded qamd_pq_sharded;
typedef struct qamd_pq_sharded_query qamd_pq_sharded_query;
/* centroids NULL: find_centroids runs once (on the device holding `data`, else devices[0]). */
QAMD_API qamd_status qamd_pq_sharded_encode(const float *data, qamd_mem data_mem,
                                            const qamd_vector_parameters *vp, uint64_t chunk_size,
                                            const float *centroids, uint32_t max_kmeans_threads,
                                            qamd_stop_fn stop, void *stop_user, const int *devices,
                                            uint32_t n_shards, void *stream,
                                             qamd_pq_sharded **out);
QAMD_API qamd_status qamd_pq_sharded_from_rows(const uint8_t *rows, qamd_mem rows_mem,
                                               const qamd_vector_parameters *vp, uint64_t chunk_size,
                                               const float *centroids, const int *devices,
                                               uint32_t n_shards, void *stream,
                                             qamd_pq_sharded **out);
QAMD_API uint32_t qamd_pq_sharded_shard_count(const qamd_pq_sharded *h);
/* *state = a qamd_peer_state for shard g; *reason (may be NULL) = a static or handle-owned string saying why. */
QAMD_API qamd_status qamd_pq_sharded_peer_access(const qamd_pq_sharded *h, uint32_t g, int *state, const char **reason);
QAMD_API qamd_status qamd_pq_sharded_shard(const qamd_pq_sharded *h, uint32_t g, const qamd_pq **shard,
                                           uint64_t *row_begin, int *device);
QAMD_API qamd_status qamd_pq_sharded_get_centroids(const qamd_pq_sharded *h, float *centroids);
QAMD_API qamd_status qamd_pq_sharded_encode_query(qamd_pq_sharded *h, const float *query, uint64_t qdim,
                                                  qamd_mem query_mem, void *stream,
                                             qamd_pq_sharded_query **query_io);
QAMD_API void qamd_pq_sharded_query_free(qamd_pq_sharded_query *q);
QAMD_API qamd_status qamd_pq_sharded_score_all(qamd_pq_sharded *h, const qamd_pq_sharded_query *q, float *out,
                                               qamd_mem out_mem, void *stream);
QAMD_API qamd_status qamd_pq_sharded_topk(qamd_pq_sharded *h, const qamd_pq_sharded_query *q, uint32_t k,
                                          int largest, uint32_t *out_ids, float *out_scores,
                                          qamd_mem out_mem, void *stream);
typedef struct qamd_pq_sharded_query_batch qamd_pq_sharded_query_batch;
QAMD_API qamd_status qamd_pq_sharded_encode_query_batch(qamd_pq_sharded *h, const float *queries,
                                                        uint64_t n_queries, uint64_t qdim,
                                                        qamd_mem queries_mem, void *stream,
                                             qamd_pq_sharded_query_batch **batch_io);
QAMD_API void qamd_pq_sharded_query_batch_free(qamd_pq_sharded_query_batch *b);
QAMD_API qamd_status qamd_pq_sharded_topk_batch(qamd_pq_sharded *h, const qamd_pq_sharded_query_batch *b,
                                                uint32_t k, int largest, uint32_t *out_ids,
                                                float *out_scores, qamd_mem out_mem, void *stream);
QAMD_API void qamd_pq_sharded_free(qamd_pq_sharded *h);

/* ===================================================================================
 * Selection over an existing score array (device memory), e.g. after a multi-GPU gather.
 * Same ordering contract as the *_topk entry points.
 * =================================================================================== */
QAMD_API qamd_status qamd_topk_scores(const float *scores_dev, uint64_t n, uint32_t k, int largest,
                                      uint32_t *out_ids, float *out_scores, qamd_mem out_mem,
                                      void *stream);

/* The exchange step of a row-sharded top-k on its own, for callers that shard across PROCESSES
 * (one rank per GPU, quantization_amd/sharded.py over torch.distributed / RCCL): after the
 * all-gather of every rank's [n_queries][k] (local id, score) lists into device memory, one kernel
 * per query block merges them -- global id = row_bases[shard] + local id, best first, ties to the
 * lower global id: exactly what a single handle over all rows returns.
 * ids_dev / scores_dev: shard g's lists start at + g * shard_stride elements; n_shards * k <= 8192.
 * row_bases: host array.  Synchronises `stream`. */
QAMD_API qamd_status qamd_topk_merge(const uint32_t *ids_dev, const float *scores_dev,
                                     uint64_t shard_stride, const uint64_t *row_bases,
                                     uint32_t n_shards, uint32_t n_queries, uint32_t k, int largest,
                                     uint32_t *out_ids, float *out_scores, qamd_mem out_mem,
                                     void *stream);

/* ===================================================================================
 * Measurement helper (bench.py's roofline arithmetic).
 * =================================================================================== */
/* Per-row store traffic of the u8 scan kernel: bytes read per scored row (codes + offset). */
QAMD_API uint64_t qamd_u8_scan_bytes_per_row(const qamd_u8 *h);

#ifdef __cplusplus
}
#endif
#endif /* QUANTIZATION_AMD_H */
