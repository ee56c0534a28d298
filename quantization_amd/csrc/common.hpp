// Shared host-side plumbing for the MI355X encode-and-score library:
// status/error text, HIP call checking, device buffers, host<->device staging.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

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

int current_device();  // device chosen by qamd_set_device on this thread
qamd_status ensure_device(int device);

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

// Copy helpers: `mem` describes the caller side.
qamd_status copy_in(void *dev_dst, const void *src, qamd_mem src_mem, size_t bytes, hipStream_t s);
qamd_status copy_out(void *dst, qamd_mem dst_mem, const void *dev_src, size_t bytes, hipStream_t s);

// 8 KiB of pinned, device-mapped host memory per calling thread (allocated on first use, never
// freed: thread exit may come after the HIP runtime is gone).  The per-pair API calls
// (score_point, score_internal, small score_ids) put their row ids there and let the kernel
// write the scores straight back: no allocation, no explicit copy, one launch + one sync.
constexpr size_t kHostScratchWords = 2048 + 16;  // [0, 1024) ids, [1024, 2048) results, [2048] a status word
struct HostScratch {
    uint32_t *host = nullptr;  // what the CPU reads / writes
    uint32_t *dev = nullptr;   // the same memory as the GPU addresses it
};
HostScratch host_scratch();

// Grow-only device workspaces per calling thread (and device), handed from call to call in
// stream order: release() records an event on the call's stream, the next acquire() makes its own
// stream wait on it.  For the whole-store calls' scratch (score vector of a top-k, radix-select
// state, PQ partial sums): hipMallocAsync + hipFreeAsync cost ~70 us per buffer and call on this
// runtime, which is most of a top-k on a small store.  Never freed (see host_scratch).
enum ThreadWsSlot { WS_SCORES = 0, WS_SELECT = 1, WS_PARTIAL = 2, WS_SLOTS = 3 };
qamd_status thread_ws_acquire(ThreadWsSlot slot, size_t bytes, hipStream_t s, void **out);
void thread_ws_release(ThreadWsSlot slot, hipStream_t s);

struct DeviceInfo {
    int cu_count = 256;
};
const DeviceInfo &device_info();

inline uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// Minimal JSON helpers for the serde_json metadata files (save/load).
std::string json_f32(float v);
bool read_file(const char *path, std::string &out);
bool write_file(const char *path, const void *data, size_t bytes);
void make_parent_dirs(const char *path);
// Finds `"key":` at any depth and parses the scalar after it.
bool json_find_number(const std::string &s, const char *key, double &out);
bool json_find_string(const std::string &s, const char *key, std::string &out);
bool json_find_bool(const std::string &s, const char *key, bool &out);

const char *distance_name(int d);
bool parse_distance(const std::string &s, int &d);
std::string vector_parameters_json(const qamd_vector_parameters &vp);
bool parse_vector_parameters(const std::string &json, qamd_vector_parameters &vp);

}  // namespace qamd
