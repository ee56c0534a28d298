// Shared host-side plumbing for the MI355X encode-and-score library:
// status/error text, HIP call checking, device buffers, host<->device staging.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdarg>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "json.hpp"

#include "../../include/quantization_amd.h"

namespace qamd {

// Thread-local error text (the String carried by EncodingError / io::Error in the reference).
std::string &last_error();
qamd_status fail(qamd_status st, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

#define QAMD_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::qamd::fail(QAMD_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr,            \
                                hipGetErrorString(_e), __FILE__, __LINE__);                 \
    } while (0)

#define QAMD_TRY(expr)                    \
    do {                                  \
        qamd_status _s = (expr);          \
        if (_s != QAMD_OK) return _s;     \
    } while (0)

int current_device();  // device chosen by qamd_set_device on this thread (default 0)
int device_count();    // cached after the first successful hipGetDeviceCount

// Makes `device` the calling thread's current HIP device for the lifetime of the guard and
// restores the caller's device afterwards (a library must not move the current device under
// its caller -- torch, or another handle's worker).  hipGetDevice is a thread-local read;
// hipSetDevice runs only when the device really differs, so a thread that stays on one GPU
// pays no runtime call beyond the read.
class DeviceGuard {
  public:
    explicit DeviceGuard(int device);
    ~DeviceGuard();
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    qamd_status status() const { return st_; }

  private:
    int prev_ = -1;
    bool switched_ = false;
    qamd_status st_ = QAMD_OK;
};
#define QAMD_ON_DEVICE(dev)                  \
    ::qamd::DeviceGuard _qamd_dev_guard(dev); \
    QAMD_TRY(_qamd_dev_guard.status())

// Developer A/B switches (which kernel serves a call; results never differ) are read from the environment ONLY by the
// tools/lib build (-DQAMD_DEV): the product library ignores them, so a stray variable cannot change which kernel serves
// production queries.  (QAMD_SHARD_LANES, a documented deployment knob of the sharded handle, is not one of these.)
inline const char *dev_env(const char *name) {
#ifdef QAMD_DEV
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// One-time set-up per (call site, device) - hipFuncSetAttribute is a per-device property, and its first call also loads
// the code object.  Two-phase: the device's bit is published only AFTER the set-up has succeeded; first callers that
// arrive together are serialised (one runs the set-up, the others wait for it and then see the bit), so no thread can
// launch with 128-160 KiB of dynamic LDS before the attribute is applied; a failed set-up leaves the bit clear and is
// retried by the next call.  run() returns the set-up's status (QAMD_OK when it had been done before).
class DeviceOnce {
  public:
    template <class F> qamd_status run(F &&setup) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return setup();  // (idempotent: repeat it every time)
        const uint64_t bit = 1ull << dev;
        if (done_.load(std::memory_order_acquire) & bit) return QAMD_OK;
        std::lock_guard<std::mutex> g(mu_);
        if (done_.load(std::memory_order_relaxed) & bit) return QAMD_OK;
        const qamd_status st = setup();
        if (st == QAMD_OK) done_.fetch_or(bit, std::memory_order_release);
        return st;
    }

  private:
    std::atomic<uint64_t> done_{0};
    std::mutex mu_;
};
// Raises a kernel's dynamic-LDS limit once per device: QAMD_LDS_OPT_IN((&kernel<...>), bytes);
#define QAMD_LDS_OPT_IN(kernel, bytes)                                                                            \
    do {                                                                                                          \
        static ::qamd::DeviceOnce _qamd_once;                                                                     \
        QAMD_TRY(_qamd_once.run([]() -> qamd_status {                                                             \
            QAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),                                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)));              \
            return QAMD_OK;                                                                                       \
        }));                                                                                                      \
    } while (0)

// Owning device allocation.
struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : ptr(o.ptr), bytes(o.bytes) { o.ptr = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) {
            release();
            ptr = o.ptr;
            bytes = o.bytes;
            o.ptr = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    qamd_status alloc(size_t n, bool zero = false);
    // bytes [keep, n) are zeroed; [0, keep) is left for the caller to fill completely
    qamd_status alloc_zero_tail(size_t n, size_t keep);
    void release();
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

// Recycles the small device buffers behind encoded-query objects.  The reference's encode_query
// returns a fresh EncodedQuery value per call (encoded_vectors.rs:27); a binding that mirrors that
// creates and drops one query object per search, and a hipMalloc + hipFree pair costs more than a
// 100k-row scan.  A freed buffer is kept (per device and size, a few dozen at most) ONLY when it is
// provably idle -- its last encode has completed and every consumer call was synchronous (host
// outputs) -- so recycling never touches a stream (which the caller may have destroyed by then);
// any other buffer is released with hipFree, which waits for the device.  Contents are not cleared:
// every encode_query overwrites what its consumers read.
qamd_status query_buf_get(size_t bytes, DevBuf &out);
void query_buf_put(DevBuf &buf, bool idle);

inline hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }

// Per-call scratch, stream-ordered (hipMallocAsync / hipFreeAsync on the call's stream).  The
// device's default pool keeps what it has been given (release threshold raised once), so a
// steady stream of calls neither pays hipMalloc nor hands memory back to the driver, whose
// clearing of returned VRAM competes with the kernels that follow.
struct StreamBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;
    StreamBuf() = default;
    StreamBuf(const StreamBuf &) = delete;
    StreamBuf &operator=(const StreamBuf &) = delete;
    ~StreamBuf() { release(); }
    qamd_status alloc(size_t n, hipStream_t s, bool zero = false);
    void release();
    template <typename T> T *as() const { return static_cast<T *>(ptr); }
};

// Orders consumers of an encoded query after the kernel that wrote it.  encode_query only ENQUEUES
// on the caller's stream, while score_point / score_internal / *_query_read run on the null stream
// and a caller may score on another stream: side streams created non-blocking are not ordered
// against the null stream, so the query remembers (stream, event) of its last encode and every
// consumer on a DIFFERENT stream waits on that event first.
struct ReadyEvent {
    hipEvent_t ev = nullptr;
    hipStream_t stream = nullptr;
    bool set = false;
    ReadyEvent() = default;
    ReadyEvent(const ReadyEvent &) = delete;
    ReadyEvent &operator=(const ReadyEvent &) = delete;
    ~ReadyEvent();
    qamd_status record(hipStream_t s);
    qamd_status wait(hipStream_t consumer) const;
    bool complete() const;  // true when nothing was recorded or the recorded work has finished
};

// Copy helpers: `mem` describes the caller side.
qamd_status copy_in(void *dev_dst, const void *src, qamd_mem src_mem, size_t bytes, hipStream_t s);
qamd_status copy_out(void *dst, qamd_mem dst_mem, const void *dev_src, size_t bytes, hipStream_t s);

// 8 KiB of pinned, device-mapped host memory per calling thread (allocated on first use; freed by
// qamd_thread_release() or at thread exit).  The per-pair API calls
// (score_point, score_internal, small score_ids) put their row ids there and let the kernel
// write the scores straight back: no allocation, no explicit copy, one launch + one sync.
constexpr size_t kHostQueryWords = 12288;  // a host query of up to 12288 f32 rides in the scratch too
constexpr size_t kHostDoneAt = 2048 + 8;  // "results are in place" flag of the single-launch top-k
constexpr size_t kHostQueryAt = 2048 + 16;
constexpr size_t kHostScratchWords = kHostQueryAt + kHostQueryWords;  // [0, 1024) ids, [1024, 2048) results,
                                                                     // [2048] a status word, [2064, ...) a query
struct HostScratch {
    uint32_t *host = nullptr;  // what the CPU reads / writes
    uint32_t *dev = nullptr;   // the same memory as the GPU addresses it
};
HostScratch host_scratch();
// The query area of the scratch: encode_query of a HOST query writes the f32 values there and the
// encode kernel reads them over the mapped pointer -- no H2D copy call, no synchronisation.
// acquire waits until the previous kernel that read the area has finished (it normally has);
// release records that on the stream the new reader was launched on.  nullptr when n_floats does
// not fit or the scratch is unavailable (callers then stage through device memory).
float *host_query_acquire(size_t n_floats, const float **dev_view);
void host_query_release(hipStream_t s);

// Grow-only device workspaces per calling thread AND device, handed from call to call: a call that
// only enqueued records the workspace's event at release(), the next acquire() makes its stream wait
// for it (no device-wide synchronisation, no stream handle kept).  For the whole-store calls' scratch (score vector of a top-k, radix-select
// state, PQ partial sums): hipMallocAsync + hipFreeAsync cost ~70 us per buffer and call on this
// runtime, which is most of a top-k on a small store.  A thread that alternates between stores
// on several GPUs keeps one workspace per (slot, device).  Freed by qamd_thread_release() or when
// the thread exits.
enum ThreadWsSlot { WS_SCORES = 0, WS_SELECT = 1, WS_PARTIAL = 2, WS_FUSED = 3, WS_SMALL = 4, WS_SLOTS = 5 };
// `tags` (optional) points at three caller-owned words that live with the buffer and are zeroed
// whenever it is (re)allocated: what the caller has cached inside it.
qamd_status thread_ws_acquire(ThreadWsSlot slot, size_t bytes, hipStream_t s, void **out,
                              uint64_t **tags = nullptr);
// synced: the caller synchronised `s` after its last use of the buffer (nothing to hand over).
void thread_ws_release(ThreadWsSlot slot, hipStream_t s, bool synced = false);
// Frees every workspace and the mapped host scratch of the calling thread (all devices).
void thread_release_all();

// A view of `bytes` of caller memory that kernels on the CURRENT device can read: device memory of
// this device is used in place; host memory and memory of another device are copied into `stage`
// (grown on demand) on stream `s`.  `*staged` tells the caller that `stage` is in use until the
// work it enqueues on `s` has finished.
qamd_status local_view(const void *src, qamd_mem mem, size_t bytes, DevBuf &stage, hipStream_t s, const void **out,
                       bool *staged);

struct DeviceInfo {
    int cu_count = 256;
};
const DeviceInfo &device_info();  // of the current device

inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// serde_json metadata files (save / load): the writer's float formatting, file helpers, and the reader (json.hpp).
std::string json_f32(float v);
bool read_file(const char *path, std::string &out);
bool write_file(const char *path, const void *data, size_t bytes);
void make_parent_dirs(const char *path);
// Reads and parses a metadata file; QAMD_ERR_IO (std::io::Error in the reference: read_to_string / serde_json::from_str,
// encoded_vectors_u8.rs:278-279) with the reader's message when it cannot.
qamd_status read_metadata(const char *meta_path, JsonValue &root);

const char *distance_name(int d);
bool parse_distance(const std::string &s, int &d);
std::string vector_parameters_json(const qamd_vector_parameters &vp);
// `obj`: the VectorParameters struct of a metadata file (encoded_vectors.rs:13-19), any key order
bool parse_vector_parameters(const JsonValue &obj, qamd_vector_parameters &vp, std::string &err);

}  // namespace qamd
