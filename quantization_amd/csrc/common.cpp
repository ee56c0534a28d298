// Host-side plumbing shared by the three quantizers (see common.hpp).
#include "common.hpp"

#include <sys/stat.h>

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <atomic>
#include <fstream>
#include <mutex>
#include <vector>

namespace qamd {

static thread_local std::string g_last_error;
static thread_local int g_device = 0;

std::string &last_error() { return g_last_error; }

qamd_status fail(qamd_status st, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return st;
}

int current_device() { return g_device; }

int device_count() {
    static std::atomic<int> cached{-1};
    int n = cached.load(std::memory_order_relaxed);
    if (n > 0) return n;
    n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return 0;  // not cached: a later call may find the runtime up
    }
    cached.store(n, std::memory_order_relaxed);
    return n;
}

DeviceGuard::DeviceGuard(int device) {
    const int n = device_count();
    if (n <= 0) {
        st_ = fail(QAMD_ERR_DEVICE, "no HIP device is visible; this library has no CPU fallback");
        return;
    }
    if (device < 0 || device >= n) {
        st_ = fail(QAMD_ERR_ARGUMENTS, "device %d out of range (have %d)", device, n);
        return;
    }
    hipError_t e = hipGetDevice(&prev_);
    if (e == hipSuccess && prev_ != device) {
        e = hipSetDevice(device);
        switched_ = e == hipSuccess;
    }
    if (e != hipSuccess) st_ = fail(QAMD_ERR_DEVICE, "cannot select device %d: %s", device, hipGetErrorString(e));
}

DeviceGuard::~DeviceGuard() {
    if (switched_) (void)hipSetDevice(prev_);
}

qamd_status DevBuf::alloc(size_t n, bool zero) {
    release();
    if (n == 0) n = 16;  // keep a valid pointer for empty stores
    QAMD_HIP(hipMalloc(&ptr, n));
    bytes = n;
    if (zero) {
        // hipMemset on device memory may return before the fill has run; callers go on to use the
        // buffer on their own (possibly non-blocking) stream, so the fill is completed here.
        QAMD_HIP(hipMemsetAsync(ptr, 0, n, nullptr));
        QAMD_HIP(hipStreamSynchronize(nullptr));
    }
    return QAMD_OK;
}

qamd_status DevBuf::alloc_zero_tail(size_t n, size_t keep) {
    QAMD_TRY(alloc(n, false));
    if (keep < bytes) {
        QAMD_HIP(hipMemsetAsync(static_cast<char *>(ptr) + keep, 0, bytes - keep, nullptr));
        QAMD_HIP(hipStreamSynchronize(nullptr));
    }
    return QAMD_OK;
}

void DevBuf::release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    bytes = 0;
}

namespace {
struct QueryBufCache {
    struct Entry { void *ptr; size_t bytes; int dev; };
    std::mutex m;
    std::vector<Entry> free_list;
};
QueryBufCache &query_cache() {
    static QueryBufCache *c = new QueryBufCache;  // leaked on purpose: no destructor-order issues at exit
    return *c;
}
constexpr size_t kQueryCacheEntries = 64;
}  // namespace

qamd_status query_buf_get(size_t bytes, DevBuf &out) {
    out.release();
    int dev = 0;
    QAMD_HIP(hipGetDevice(&dev));
    {
        QueryBufCache &c = query_cache();
        std::lock_guard<std::mutex> lk(c.m);
        for (size_t i = 0; i < c.free_list.size(); i++) {
            if (c.free_list[i].dev == dev && c.free_list[i].bytes == bytes) {
                out.ptr = c.free_list[i].ptr;
                out.bytes = bytes;
                c.free_list[i] = c.free_list.back();
                c.free_list.pop_back();
                return QAMD_OK;
            }
        }
    }
    return out.alloc(bytes, true);
}

void query_buf_put(DevBuf &buf, bool idle) {
    if (!buf.ptr) return;
    hipPointerAttribute_t attr{};
    if (idle && hipPointerGetAttributes(&attr, buf.ptr) == hipSuccess) {
        QueryBufCache &c = query_cache();
        std::lock_guard<std::mutex> lk(c.m);
        if (c.free_list.size() < kQueryCacheEntries) {
            c.free_list.push_back({buf.ptr, buf.bytes, attr.device});
            buf.ptr = nullptr;
            buf.bytes = 0;
            return;
        }
    } else if (idle) {
        (void)hipGetLastError();
    }
    buf.release();
}

qamd_status StreamBuf::alloc(size_t n, hipStream_t s, bool zero) {
    release();
    static std::once_flag once[64];  // per device: its default pool keeps what it is given
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        std::call_once(once[dev], [dev] {
            hipMemPool_t pool = nullptr;
            if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool) {
                uint64_t keep = ~0ull;
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            }
        });
    }
    if (n == 0) n = 16;
    stream = s;
    QAMD_HIP(hipMallocAsync(&ptr, n, s));
    bytes = n;
    if (zero) QAMD_HIP(hipMemsetAsync(ptr, 0, n, s));
    return QAMD_OK;
}

void StreamBuf::release() {
    if (ptr) (void)hipFreeAsync(ptr, stream);
    ptr = nullptr;
    bytes = 0;
}

ReadyEvent::~ReadyEvent() {
    if (ev) (void)hipEventDestroy(ev);
}

qamd_status ReadyEvent::record(hipStream_t s) {
    if (!ev) QAMD_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    QAMD_HIP(hipEventRecord(ev, s));
    stream = s;
    set = true;
    return QAMD_OK;
}

bool ReadyEvent::complete() const {
    if (!set || !ev) return true;
    if (hipEventQuery(ev) == hipSuccess) return true;
    (void)hipGetLastError();  // hipErrorNotReady
    return false;
}

qamd_status ReadyEvent::wait(hipStream_t consumer) const {
    if (!set || consumer == stream) return QAMD_OK;
    // usually the encode finished long ago: a query of the event (~1 us) instead of a wait command on
    // the consumer's stream (~5 us of host time on this runtime).  Not while the consumer's stream is
    // being captured: hipEventQuery is not a capturable call (it would invalidate the capture).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(consumer, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap == hipStreamCaptureStatusNone) {
        if (hipEventQuery(ev) == hipSuccess) return QAMD_OK;
        (void)hipGetLastError();
    }
    QAMD_HIP(hipStreamWaitEvent(consumer, ev, 0));
    return QAMD_OK;
}

qamd_status copy_in(void *dev_dst, const void *src, qamd_mem src_mem, size_t bytes, hipStream_t s) {
    if (bytes == 0) return QAMD_OK;
    if (src_mem == QAMD_MEM_DEVICE) {
        QAMD_HIP(hipMemcpyAsync(dev_dst, src, bytes, hipMemcpyDeviceToDevice, s));
    } else {
        QAMD_HIP(hipMemcpyAsync(dev_dst, src, bytes, hipMemcpyHostToDevice, s));
        QAMD_HIP(hipStreamSynchronize(s));
    }
    return QAMD_OK;
}

qamd_status copy_out(void *dst, qamd_mem dst_mem, const void *dev_src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return QAMD_OK;
    if (dst_mem == QAMD_MEM_DEVICE) {
        QAMD_HIP(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToDevice, s));
    } else {
        QAMD_HIP(hipMemcpyAsync(dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
        QAMD_HIP(hipStreamSynchronize(s));
    }
    return QAMD_OK;
}

qamd_status local_view(const void *src, qamd_mem mem, size_t bytes, DevBuf &stage, hipStream_t s, const void **out,
                       bool *staged) {
    *out = src;
    *staged = false;
    if (bytes == 0) return QAMD_OK;
    bool remote = false;
    if (mem == QAMD_MEM_DEVICE && device_count() > 1) {
        hipPointerAttribute_t attr{};
        int dev = 0;
        if (hipPointerGetAttributes(&attr, src) == hipSuccess && hipGetDevice(&dev) == hipSuccess)
            remote = attr.device != dev;
        else
            (void)hipGetLastError();
    }
    if (mem == QAMD_MEM_DEVICE && !remote) return QAMD_OK;
    if (stage.bytes < bytes) QAMD_TRY(stage.alloc(bytes));
    QAMD_HIP(hipMemcpyAsync(stage.ptr, src, bytes, mem == QAMD_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDefault, s));
    if (mem == QAMD_MEM_HOST) QAMD_HIP(hipStreamSynchronize(s));  // pageable source: the call owns it only now
    *out = stage.ptr;
    *staged = true;
    return QAMD_OK;
}

namespace {

constexpr int kMaxDevices = 64;

struct ThreadWs {
    void *ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;  // recorded behind the last call that only enqueued work on the buffer
    bool in_flight = false;     // `done` may not have completed yet
    uint64_t tags[3] = {0, 0, 0};
};

// Everything the calling thread owns on the GPUs: freed by qamd_thread_release() and by the
// thread_local destructor (a thread that ends while the process lives; at process exit the main
// thread's thread_locals are destroyed before the HIP runtime's own static teardown).
struct ThreadState {
    HostScratch scratch;
    bool scratch_tried = false;
    hipEvent_t query_read = nullptr;  // the last kernel that read the scratch's query area
    bool query_pending = false;
    std::vector<ThreadWs> ws;  // [device][slot], grown on demand

    ThreadWs &at(int dev, ThreadWsSlot slot) {
        const size_t need = (size_t)(dev + 1) * WS_SLOTS;
        if (ws.size() < need) ws.resize(need);
        return ws[(size_t)dev * WS_SLOTS + slot];
    }

    void release() {
        int prev = -1;
        const bool have_prev = hipGetDevice(&prev) == hipSuccess;
        bool moved = false;
        for (size_t i = 0; i < ws.size(); i++) {
            ThreadWs &w = ws[i];
            if (!w.ptr) continue;
            const int dev = (int)(i / WS_SLOTS);
            if (hipSetDevice(dev) == hipSuccess) {
                moved = true;
                (void)hipFree(w.ptr);  // waits for the device: nothing of ours can still be using it
                if (w.done) (void)hipEventDestroy(w.done);
            }
            w = ThreadWs{};
        }
        ws.clear();
        if (moved && have_prev) (void)hipSetDevice(prev);
        if (query_pending && query_read) (void)hipEventSynchronize(query_read);
        if (query_read) (void)hipEventDestroy(query_read);
        query_read = nullptr;
        query_pending = false;
        if (scratch.host) (void)hipHostFree(scratch.host);
        scratch = HostScratch{};
        scratch_tried = false;
        (void)hipGetLastError();
    }

    ~ThreadState() { release(); }
};

ThreadState &thread_state() {
    static thread_local ThreadState st;
    return st;
}

}  // namespace

HostScratch host_scratch() {
    ThreadState &ts = thread_state();
    if (!ts.scratch_tried) {
        ts.scratch_tried = true;
        void *p = nullptr, *d = nullptr;
        if (hipHostMalloc(&p, kHostScratchWords * 4, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess) {
            if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess) {
                ts.scratch.host = static_cast<uint32_t *>(p);
                ts.scratch.dev = static_cast<uint32_t *>(d);
            } else {
                (void)hipHostFree(p);
            }
        }
    }
    return ts.scratch;
}

float *host_query_acquire(size_t n_floats, const float **dev_view) {
    if (n_floats > kHostQueryWords) return nullptr;
    const HostScratch hs = host_scratch();
    if (!hs.host) return nullptr;
    ThreadState &ts = thread_state();
    if (ts.query_pending) {
        if (hipEventSynchronize(ts.query_read) != hipSuccess) return nullptr;
        ts.query_pending = false;
    }
    if (!ts.query_read && hipEventCreateWithFlags(&ts.query_read, hipEventDisableTiming) != hipSuccess) {
        ts.query_read = nullptr;
        return nullptr;
    }
    *dev_view = reinterpret_cast<const float *>(hs.dev + kHostQueryAt);
    return reinterpret_cast<float *>(hs.host + kHostQueryAt);
}

void host_query_release(hipStream_t s) {
    ThreadState &ts = thread_state();
    if (ts.query_read && hipEventRecord(ts.query_read, s) == hipSuccess) ts.query_pending = true;
}

// Hand-off between consecutive calls of one thread.  A call that synchronised its stream before it
// released the workspace (host outputs) leaves nothing behind.  A call that only ENQUEUED records the
// workspace's own event behind its work; the next acquire -- on whatever stream -- makes that stream
// wait for the event unless it has completed already.  Events outlive streams, so nothing is ever
// asked of a stream the caller may have destroyed since, no stream handle is compared, and there is
// no device-wide synchronisation (which would stall other threads' scans and is illegal inside a
// stream capture).  While a stream is being captured nothing is recorded: the captured work runs
// when the graph is launched, and a graph that uses a thread's workspace must not be replayed
// concurrently with other calls of that thread (include/quantization_amd.h, conventions).
qamd_status thread_ws_acquire(ThreadWsSlot slot, size_t bytes, hipStream_t s, void **out, uint64_t **tags) {
    int dev = 0;
    QAMD_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(QAMD_ERR_DEVICE, "device index %d not supported", dev);
    ThreadWs &w = thread_state().at(dev, slot);
    if (bytes == 0) bytes = 16;
    if (w.bytes < bytes) {
        if (w.ptr) {
            (void)hipFree(w.ptr);  // synchronises with whatever still uses it
            w.ptr = nullptr;
        }
        w.bytes = 0;
        w.in_flight = false;
        w.tags[0] = w.tags[1] = w.tags[2] = 0;
        const size_t want = bytes + bytes / 4;  // head room: stores grow, k varies
        QAMD_HIP(hipMalloc(&w.ptr, want));
        w.bytes = want;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (w.in_flight && hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
    if (w.in_flight && cap == hipStreamCaptureStatusNone) {  // (a capture records, it does not run: see above)
        const hipError_t q = hipEventQuery(w.done);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            QAMD_HIP(hipStreamWaitEvent(s, w.done, 0));
        } else if (q != hipSuccess) {
            return fail(QAMD_ERR_DEVICE, "workspace hand-off: hipEventQuery failed: %s", hipGetErrorString(q));
        } else {
            w.in_flight = false;
        }
    }
    *out = w.ptr;
    if (tags) *tags = w.tags;
    return QAMD_OK;
}

void thread_ws_release(ThreadWsSlot slot, hipStream_t s, bool synced) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return;
    ThreadWs &w = thread_state().at(dev, slot);
    if (synced) {
        w.in_flight = false;
        return;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
    // A captured call launches nothing now: whatever an EARLIER enqueue-only call of this thread left in flight (its
    // event, in_flight) is still the state the next eager call must wait for - leave both untouched.
    if (cap != hipStreamCaptureStatusNone) return;
    w.in_flight = false;
    if (!w.done && hipEventCreateWithFlags(&w.done, hipEventDisableTiming) != hipSuccess) {
        w.done = nullptr;
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);  // no event to hand over: finish the work instead
        return;
    }
    if (hipEventRecord(w.done, s) == hipSuccess) w.in_flight = true;
    else (void)hipStreamSynchronize(s);
}

void thread_release_all() { thread_state().release(); }

const DeviceInfo &device_info() {
    static DeviceInfo info[kMaxDevices];
    static std::atomic<uint64_t> known{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    if (!(known.load(std::memory_order_acquire) & (1ull << dev))) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            info[dev].cu_count = p.multiProcessorCount;
        known.fetch_or(1ull << dev, std::memory_order_acq_rel);
    }
    return info[dev];
}

// ---------------------------------------------------------------------------------- JSON
std::string json_f32(float v) {
    if (!std::isfinite(v)) return "null";  // serde_json writes null for NaN/inf
    // serde_json prints floats with ryu: the shortest decimal digits that read back as the same f32, laid out as
    // 12340000000.0 / 12.34 / 0.001234 while -6 < decimal exponent <= 13 and as 1e30 / 1.234e33 outside (ryu's
    // pretty::format32).  Same bytes as the reference's save() for every finite value.
    std::string sign = std::signbit(v) ? "-" : "";
    if (v == 0.0f) return sign + "0.0";
    char buf[64];
    const double a = std::fabs((double)v);
    for (int p = 0; p <= 8; p++) {
        snprintf(buf, sizeof buf, "%.*e", p, a);
        if (strtof(buf, nullptr) == std::fabs(v)) break;
    }
    std::string digits;
    const char *e = strchr(buf, 'e');
    for (const char *c = buf; c < e; c++)
        if (*c != '.') digits += *c;
    const int exp10 = atoi(e + 1), length = (int)digits.size();
    const int kk = exp10 + 1, k = kk - length;  // value = digits * 10^k, 10^(kk-1) <= value < 10^kk
    if (0 <= k && kk <= 13) return sign + digits + std::string((size_t)k, '0') + ".0";
    if (0 < kk && kk <= 13) return sign + digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    if (-6 < kk && kk <= 0) return sign + "0." + std::string((size_t)(-kk), '0') + digits;
    if (length == 1) return sign + digits + "e" + std::to_string(kk - 1);
    return sign + digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
}

bool read_file(const char *path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n > 0) f.read(&out[0], n);
    return bool(f) || f.eof();
}

bool write_file(const char *path, const void *data, size_t bytes) {
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) return false;
    if (bytes) f.write(static_cast<const char *>(data), (std::streamsize)bytes);
    f.flush();
    return bool(f);
}

void make_parent_dirs(const char *path) {
    std::string p(path);
    size_t pos = p.find_last_of('/');
    if (pos == std::string::npos || pos == 0) return;
    std::string dir = p.substr(0, pos);
    for (size_t i = 1; i <= dir.size(); i++) {
        if (i == dir.size() || dir[i] == '/') {
            std::string sub = dir.substr(0, i);
            if (mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST) return;
        }
    }
}

qamd_status read_metadata(const char *meta_path, JsonValue &root) {
    std::string js, err;
    if (!read_file(meta_path, js)) return fail(QAMD_ERR_IO, "cannot read %s", meta_path);
    if (!json_parse(js, root, err)) return fail(QAMD_ERR_IO, "%s: %s", meta_path, err.c_str());
    return QAMD_OK;
}

const char *distance_name(int d) { return d == QAMD_DOT ? "Dot" : d == QAMD_L1 ? "L1" : "L2"; }

bool parse_distance(const std::string &s, int &d) {
    if (s == "Dot") d = QAMD_DOT;
    else if (s == "L1") d = QAMD_L1;
    else if (s == "L2") d = QAMD_L2;
    else return false;
    return true;
}

// serde field order of VectorParameters (encoded_vectors.rs:13-19)
std::string vector_parameters_json(const qamd_vector_parameters &vp) {
    char buf[256];
    snprintf(buf, sizeof buf, "{\"dim\":%llu,\"count\":%llu,\"distance_type\":\"%s\",\"invert\":%s}",
             (unsigned long long)vp.dim, (unsigned long long)vp.count, distance_name(vp.distance_type),
             vp.invert ? "true" : "false");
    return buf;
}

bool parse_vector_parameters(const JsonValue &obj, qamd_vector_parameters &vp, std::string &err) {
    uint64_t dim = 0, count = 0;
    std::string dist;
    bool inv = false;
    if (!json_usize(obj, "dim", dim, err) || !json_usize(obj, "count", count, err) ||
        !json_string(obj, "distance_type", dist, err) || !json_bool(obj, "invert", inv, err))
        return false;
    int d = 0;
    if (!parse_distance(dist, d)) {
        err = "unknown variant `" + dist + "`, expected one of `Dot`, `L1`, `L2`";
        return false;
    }
    vp.dim = dim;
    vp.count = count;
    vp.distance_type = d;
    vp.invert = inv ? 1 : 0;
    return true;
}

}  // namespace qamd

extern "C" {

const char *qamd_last_error(void) { return qamd::last_error().c_str(); }
const char *qamd_version(void) { return "quantization_amd 0.1 (gfx950)"; }

int qamd_device_count(void) { return qamd::device_count(); }

qamd_status qamd_set_device(int device) {
    const int n = qamd::device_count();
    if (n <= 0) return qamd::fail(QAMD_ERR_DEVICE, "no HIP device is visible; this library has no CPU fallback");
    if (device < 0 || device >= n) return qamd::fail(QAMD_ERR_ARGUMENTS, "device %d out of range (have %d)", device, n);
    qamd::g_device = device;
    return QAMD_OK;
}

int qamd_get_device(void) { return qamd::g_device; }

void qamd_thread_release(void) { qamd::thread_release_all(); }
}
