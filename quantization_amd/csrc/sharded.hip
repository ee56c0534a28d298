// Row-sharded stores: ONE process drives several GPUs of a node through one handle.
//
// The reference has no multi-device path; its caller is one process looping score_point over the
// whole store (demos/src/ann_benchmark.rs:245-260) and keeping the best 30 in a heap
// (demos/src/ann_benchmark_data.rs:151-167).  Rows are independent, the metadata (alpha/offset/
// multiplier, centroids) is global and tiny, the query is replicated: so shard g of G owns the
// contiguous row range [g*N/G, (g+1)*N/G) as an ordinary single-device handle on devices[g], and a
// global row id is shard base + local id.  `devices[]` may name one device several times (logical
// shards: how the single-GPU tests exercise the index arithmetic and the merge).
//
// Execution: one persistent worker thread per shard, bound to the shard's device with its own
// stream; a call posts one job per worker (so the G scans / top-ks run concurrently, each
// worker owning its per-thread workspaces) and waits.  Then exactly one exchange:
//   score_all : every shard's scores go straight to their slice of the output — host output: one
//               D2H copy per GPU over its own PCIe link; device output (on devices[0]): a peer copy
//               over xGMI (hipMemcpyAsync, 4 B/row), the single-process form of the score gather.
//   topk      : per-shard exact top-k (k pairs), peer copy of the G*k pairs to devices[0], ONE
//               merge kernel there (bitonic sort on (score key, global id): the single-handle tie
//               rule), result out through mapped host memory — no per-query host merge.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"
#include "topk_device.hpp"
#include "u8_internal.hpp"

using namespace qamd;

namespace qamd {
// pq.hip: find_centroids only (the sharded encoder trains once, then every shard encodes with the result)
qamd_status pq_train_centroids(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                               uint64_t chunk_size, uint32_t max_kmeans_threads, qamd_stop_fn stop, void *stop_user,
                               hipStream_t s, std::vector<float> &centroids, uint32_t *iterations, uint32_t *empties);
}  // namespace qamd

namespace {

// ------------------------------------------------------------------------------------ workers
// Execution model.  Every shard has a job queue served by kLanes worker threads ("lanes"), each bound
// to the shard's device with its own stream and its own per-thread workspaces.  A call from ANY caller
// thread posts one job per shard and waits for its own jobs only, so K threads searching one sharded
// handle (the reference's `&self` methods are called from many search threads at once,
// quantization/src/encoded_vectors.rs:21-35) interleave on the shards' queues instead of taking turns
// on the handle: there is no per-handle lock on the query path.  What a call needs beyond the shard
// handles -- the gather / result buffers on devices[0], the per-shard staging buffers, a root stream
// -- lives in a CallSlot leased for the duration of the call (as many slots as there are concurrent
// callers, created on demand).
// A lane sleeps on the queue's condition variable between jobs, but both sides first poll for ~50 us:
// searches arrive back to back, and a futex wake-up (10-20 us) per shard and call would otherwise be
// a fifth of a 1.25M-row shard scan.
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}
constexpr int kSpinIterations = 4000;
constexpr uint32_t kMaxLanes = 8;

uint32_t lanes_per_shard() {  // QAMD_SHARD_LANES=1..8 (default 3: while one lane waits for its kernel, others enqueue)
    static const uint32_t lanes = [] {
        const char *e = getenv("QAMD_SHARD_LANES");
        const long v = e ? strtol(e, nullptr, 10) : 3;
        return (uint32_t)(v < 1 ? 1 : v > (long)kMaxLanes ? kMaxLanes : v);
    }();
    return lanes;
}

struct Worker;  // a lane

// One fan-out: completed when every job posted for it has run.  Lives on the caller's stack; the
// lane that finishes the last job sets `done` and notifies UNDER the mutex, and the caller leaves
// wait() only after taking that mutex, so no lane touches a Call its owner has already destroyed.
struct Call {
    std::atomic<uint32_t> pending{0};
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    qamd_status status = QAMD_OK;  // first failure
    std::string error;

    void finish_one(qamd_status st, const std::string &err) {
        if (st != QAMD_OK) {
            std::lock_guard<std::mutex> lk(m);
            if (status == QAMD_OK) {
                status = st;
                error = err;
            }
        }
        if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(m);
            done = true;
            cv.notify_all();
        }
    }

    qamd_status wait() {
        for (int i = 0; i < kSpinIterations * 8 && pending.load(std::memory_order_acquire) != 0; i++) cpu_relax();
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        if (status != QAMD_OK) last_error() = error;
        return status;
    }
};

struct Job {
    const std::function<qamd_status(uint32_t, Worker &)> *fn = nullptr;
    uint32_t shard = 0;
    Call *call = nullptr;
};

struct ShardQueue {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> jobs;
    std::atomic<uint32_t> queued{0};  // == jobs.size(), readable without the lock (the lanes' poll)
    bool quit = false;

    void push(const Job &j) {
        {
            std::lock_guard<std::mutex> lk(m);
            jobs.push_back(j);
            queued.store((uint32_t)jobs.size(), std::memory_order_release);
        }
        cv.notify_one();
    }
};

struct Worker {
    int device = 0;
    hipStream_t stream = nullptr;
    ShardQueue *queue = nullptr;
    std::thread th;
    // set-up hand-shake with Pool::start (a lane that cannot select its device or create its stream
    // must not silently run shard jobs on the null stream of whatever device is current)
    std::mutex ready_m;
    std::condition_variable ready_cv;
    bool ready = false;
    qamd_status setup = QAMD_OK;
    std::string setup_error;

    void loop() {
        qamd_status st = QAMD_OK;
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) st = fail(QAMD_ERR_DEVICE, "shard worker: hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
        if (st == QAMD_OK) st = qamd_set_device(device);  // handles this thread creates live on the shard's device
        if (st == QAMD_OK && (e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)) != hipSuccess) {
            stream = nullptr;
            st = fail(QAMD_ERR_DEVICE, "shard worker: cannot create a stream on device %d: %s", device, hipGetErrorString(e));
        }
        {
            std::lock_guard<std::mutex> lk(ready_m);
            setup = st;
            if (st != QAMD_OK) setup_error = last_error();
            ready = true;
            ready_cv.notify_all();
        }
        if (st != QAMD_OK) {
            if (stream) (void)hipStreamDestroy(stream);
            return;
        }
        ShardQueue &q = *queue;
        for (;;) {
            for (int i = 0; i < kSpinIterations && q.queued.load(std::memory_order_acquire) == 0; i++) cpu_relax();
            Job job;
            {
                std::unique_lock<std::mutex> lk(q.m);
                q.cv.wait(lk, [&] { return !q.jobs.empty() || q.quit; });
                if (q.jobs.empty()) break;  // quit, nothing left
                job = q.jobs.front();
                q.jobs.pop_front();
                q.queued.store((uint32_t)q.jobs.size(), std::memory_order_release);
            }
            const qamd_status r = (*job.fn)(job.shard, *this);
            std::string err;
            if (r != QAMD_OK) {
                // a job that failed half-way returned without its own wait: what it had already enqueued on this lane's
                // stream (copies into the call's leased buffers, kernels) must be finished BEFORE the caller hears of the
                // failure and gives the lease back - another caller's job may be handed the same buffers next
                err = last_error();
                (void)hipStreamSynchronize(stream);
                (void)hipGetLastError();
            }
            job.call->finish_one(r, err);
        }
        (void)hipStreamSynchronize(stream);
        (void)hipStreamDestroy(stream);
        thread_release_all();  // this thread's workspaces go with it
    }
};

struct Pool {
    std::vector<std::unique_ptr<ShardQueue>> queues;  // one per shard
    std::vector<std::unique_ptr<Worker>> workers;     // kLanes per shard

    qamd_status start(const int *devices, uint32_t n) {
        const int have = device_count();
        if (have <= 0) return fail(QAMD_ERR_DEVICE, "no HIP device is visible; this library has no CPU fallback");
        for (uint32_t g = 0; g < n; g++)
            if (devices[g] < 0 || devices[g] >= have)
                return fail(QAMD_ERR_ARGUMENTS, "shard %u: device %d out of range (have %d)", g, devices[g], have);
        const uint32_t lanes = lanes_per_shard();
        for (uint32_t g = 0; g < n; g++) {
            queues.emplace_back(new ShardQueue);
            for (uint32_t l = 0; l < lanes; l++) {
                std::unique_ptr<Worker> w(new Worker);
                w->device = devices[g];
                w->queue = queues.back().get();
                Worker *raw = w.get();
                w->th = std::thread([raw] { raw->loop(); });
                workers.push_back(std::move(w));
            }
        }
        for (auto &w : workers) {
            std::unique_lock<std::mutex> lk(w->ready_m);
            w->ready_cv.wait(lk, [&] { return w->ready; });
            if (w->setup != QAMD_OK) {
                last_error() = w->setup_error;
                return w->setup;  // ~Pool joins what was started
            }
        }
        return QAMD_OK;
    }

    // fn(g, lane) for every shard, concurrently with other callers' jobs; first failure wins.
    qamd_status run(const std::function<qamd_status(uint32_t, Worker &)> &fn) {
        Call call;
        const uint32_t G = (uint32_t)queues.size();
        call.pending.store(G, std::memory_order_relaxed);
        for (uint32_t g = 0; g < G; g++) queues[g]->push(Job{&fn, g, &call});
        return call.wait();
    }

    ~Pool() {
        for (auto &q : queues) {
            {
                std::lock_guard<std::mutex> lk(q->m);
                q->quit = true;
            }
            q->cv.notify_all();
        }
        for (auto &w : workers)
            if (w->th.joinable()) w->th.join();
    }
};

// ------------------------------------------------------------------------------------ merge
// One workgroup per query: the G per-shard lists (k pairs each, local ids) become the global best
// k.  Keys are (order-preserving score bits << 32 | global id), all distinct, so the result is the
// single-handle one: best first, ties to the lower global row id.
__global__ __launch_bounds__(1024) void merge_topk_kernel(const uint32_t *__restrict__ ids /*[G] x [Q][k]*/,
                                                         const float *__restrict__ scores /*[G] x [Q][k]*/,
                                                         uint64_t shard_stride /* elements between two shards' lists */,
                                                         const uint64_t *__restrict__ bases /*[G]*/, uint32_t G, uint32_t Q,
                                                         uint32_t k, int largest, uint32_t N /* pow2 >= G*k */,
                                                         uint32_t *__restrict__ out_ids /*[Q][k]*/,
                                                         float *__restrict__ out_scores) {
    extern __shared__ unsigned long long keys[];  // N
    const uint32_t q = blockIdx.x, t = threadIdx.x, total = G * k;
    for (uint32_t i = t; i < N; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < total) {
            const uint32_t g = i / k, j = i - g * k;
            const size_t at = (size_t)g * shard_stride + (size_t)q * k + j;
            const uint32_t id = ids[at];
            if (id != 0xFFFFFFFFu)  // a shard shorter than k pads its list
                key = ((unsigned long long)topk_ordered_bits(scores[at], largest != 0) << 32) | (uint32_t)(bases[g] + id);
        }
        keys[i] = key;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= N; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = t; i < N / 2; i += 1024) {
                const uint32_t a = 2 * i - (i & (stride - 1)), b = a + stride;
                const bool up = (a & size) == 0;
                const unsigned long long x = keys[a], y = keys[b];
                if ((x > y) == up) {
                    keys[a] = y;
                    keys[b] = x;
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = t; i < k; i += 1024) {
        const unsigned long long key = i < N ? keys[i] : ~0ull;
        if (key != ~0ull) {
            out_ids[(size_t)q * k + i] = (uint32_t)(key & 0xFFFFFFFFull);
            out_scores[(size_t)q * k + i] = topk_score_of_key((uint32_t)(key >> 32), largest != 0);
        } else {
            out_ids[(size_t)q * k + i] = 0xFFFFFFFFu;
            out_scores[(size_t)q * k + i] = largest ? -__builtin_huge_valf() : __builtin_huge_valf();
        }
    }
}

uint32_t pow2_at_least(uint32_t v) {
    uint32_t n = 64;
    while (n < v) n <<= 1;
    return n;
}

void shard_bounds(uint64_t count, uint32_t G, std::vector<uint64_t> &base) {
    base.resize(G + 1);
    for (uint32_t g = 0; g <= G; g++) base[g] = (uint64_t)((unsigned __int128)g * count / G);
}

// Which device owns `ptr` (device memory), or -1.
int device_of(const void *ptr) {
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return attr.device;
}

// ------------------------------------------------------------------------------------ generic store
// Ops: the single-device C ABI of one quantizer (identical shapes for u8 / bin / pq).
template <class H, class Qy> struct Ops {
    qamd_status (*encode_query)(const H *, const float *, uint64_t, qamd_mem, void *, Qy **);
    void (*query_free)(Qy *);
    qamd_status (*score_all)(const H *, const Qy *, float *, qamd_mem, void *);
    qamd_status (*topk)(const H *, const Qy *, uint32_t, int, uint32_t *, float *, qamd_mem, void *);
    void (*free_store)(H *);
    // encode_query of a HOST query of this many values launches nothing (the values are kept in the query
    // object until its first consumer, u8.hip): the sharded call then needs no fan-out either
    bool (*host_encode_is_lazy)(uint64_t qdim);
};

template <class H, class Qy> struct ShardedQuery {
    std::vector<Qy *> per_shard;
    const Ops<H, Qy> *ops = nullptr;
    ~ShardedQuery() {
        for (Qy *q : per_shard)
            if (q) ops->query_free(q);
    }
};

// What one in-flight call owns besides the shard handles (leased, see the execution model above).
struct CallSlot {
    hipStream_t root_stream = nullptr;  // on devices[0]: the merge kernel and the result download
    DevBuf gather, result;              // [G][Q][k] ids | [G][Q][k] scores ; [Q][k] ids | scores (devices[0])
    size_t gather_cap = 0, result_cap = 0;
    // per shard, on the shard's device: its k pairs / its scores before the peer copy, a staged query
    std::vector<DevBuf> local_pairs, local_scores, query_stage;
    bool busy = false;
};

// Waits for `s` by polling first: searches on small shards finish within tens of microseconds, less
// than the 20-30 us wake-up latency of a blocking synchronisation on this runtime.  Every lane waits
// for its own stream (in parallel), then the caller for the root stream.  (Tried instead: lanes only
// record an event and the root stream waits for the G events -- hipStreamWaitEvent costs ~5 us of
// host time per event, serially on the caller: slower from 4 shards on.)
qamd_status wait_stream(hipStream_t s) {
    for (int i = 0; i < 5000; i++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return QAMD_OK;
        if (e != hipErrorNotReady) return fail(QAMD_ERR_DEVICE, "hipStreamQuery failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        for (int j = 0; j < 64; j++) cpu_relax();  // ~1-2 us between polls: the query takes runtime locks other threads need
    }
    QAMD_HIP(hipStreamSynchronize(s));
    return QAMD_OK;
}

// The caller's stream and the buffers it hands over.  Sharded calls run on the lanes' own streams,
// which nothing orders against the stream that PRODUCED a caller's device buffer (or that still reads
// the buffer a result will overwrite): when a buffer of the call is device memory, the caller's
// stream is synchronised before the first lane touches it -- NULL meaning the null stream of the
// device that owns the buffer (torch's default stream).  Every sharded call is synchronous, so at
// return the outputs are complete and visible to any stream.
qamd_status order_after_caller(qamd_mem mem, const void *buf, void *stream) {
    if (mem != QAMD_MEM_DEVICE || !buf) return QAMD_OK;
    if (stream) {
        QAMD_HIP(hipStreamSynchronize(as_stream(stream)));
        return QAMD_OK;
    }
    const int dev = device_of(buf);
    QAMD_ON_DEVICE(dev < 0 ? current_device() : dev);
    QAMD_HIP(hipStreamSynchronize(nullptr));
    return QAMD_OK;
}

template <class H, class Qy> struct Sharded {
    const Ops<H, Qy> *ops = nullptr;
    uint64_t count = 0;
    std::vector<int> devices;
    std::vector<uint64_t> base;  // G + 1 row bounds
    std::vector<H *> shards;
    Pool pool;
    DevBuf bases_dev;  // [G] u64 on devices[0], written once
    struct Peer {
        int state = QAMD_PEER_SAME_DEVICE;
        std::string reason = "same device as devices[0]";
    };
    std::vector<Peer> peer;  // how shard g's device reaches devices[0] (recorded at construction, reported on request)
    std::mutex slots_m;
    std::vector<std::unique_ptr<CallSlot>> slots;

    uint32_t G() const { return (uint32_t)shards.size(); }
    int root() const { return devices[0]; }

    // RAII lease of a CallSlot: an idle one, or a new one (one per concurrent caller at most).
    struct Lease {
        Sharded *owner = nullptr;
        CallSlot *slot = nullptr;
        Lease() = default;
        Lease(const Lease &) = delete;
        Lease &operator=(const Lease &) = delete;
        ~Lease() {
            if (!slot) return;
            std::lock_guard<std::mutex> lk(owner->slots_m);
            slot->busy = false;
        }
        CallSlot *operator->() const { return slot; }
    };
    qamd_status lease(Lease &l) {
        {
            std::lock_guard<std::mutex> lk(slots_m);
            for (auto &sl : slots)
                if (!sl->busy) {
                    sl->busy = true;
                    l.owner = this;
                    l.slot = sl.get();
                    return QAMD_OK;
                }
        }
        std::unique_ptr<CallSlot> fresh(new CallSlot);
        fresh->local_pairs.resize(G());
        fresh->local_scores.resize(G());
        fresh->query_stage.resize(G());
        {
            QAMD_ON_DEVICE(root());
            QAMD_HIP(hipStreamCreateWithFlags(&fresh->root_stream, hipStreamNonBlocking));
        }
        fresh->busy = true;
        std::lock_guard<std::mutex> lk(slots_m);
        l.owner = this;
        l.slot = fresh.get();
        slots.push_back(std::move(fresh));
        return QAMD_OK;
    }

    qamd_status init(const int *devs, uint32_t n, uint64_t total) {
        if (!devs || n == 0 || n > 64) return fail(QAMD_ERR_ARGUMENTS, "need 1..64 shards");
        if (total > 0xFFFFFFFFull) return fail(QAMD_ERR_ARGUMENTS, "count exceeds u32 row ids");
        count = total;
        devices.assign(devs, devs + n);
        shard_bounds(total, n, base);
        shards.assign(n, nullptr);
        QAMD_TRY(pool.start(devs, n));
        // direct xGMI copies between the shards' devices and devices[0] (the exchanges are peer copies); where peer
        // access cannot be enabled the same hipMemcpyAsync calls still work, staged through host memory by the runtime:
        // the outcome is recorded per shard (qamd_*_sharded_peer_access), never ignored
        peer.assign(n, Peer{});
        for (uint32_t g = 1; g < n; g++) {
            if (devs[g] == devs[0]) continue;
            int can = 0;
            const hipError_t ce = hipDeviceCanAccessPeer(&can, devs[0], devs[g]);
            if (ce != hipSuccess || !can) {
                peer[g] = Peer{QAMD_PEER_UNAVAILABLE, ce != hipSuccess ? std::string("hipDeviceCanAccessPeer failed: ") + hipGetErrorString(ce)
                                                                      : "hipDeviceCanAccessPeer: no peer access between the devices; copies are staged by the runtime"};
                (void)hipGetLastError();
                continue;
            }
            auto enable = [](int from, int to) {
                DeviceGuard a(from);
                const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
                (void)hipGetLastError();
                return e == hipErrorPeerAccessAlreadyEnabled ? hipSuccess : e;
            };
            const hipError_t e1 = enable(devs[0], devs[g]), e2 = enable(devs[g], devs[0]);
            if (e1 == hipSuccess && e2 == hipSuccess)
                peer[g] = Peer{QAMD_PEER_ENABLED, "hipDeviceEnablePeerAccess in both directions"};
            else
                peer[g] = Peer{QAMD_PEER_FAILED, std::string("hipDeviceEnablePeerAccess failed: ") +
                                                     hipGetErrorString(e1 != hipSuccess ? e1 : e2) + "; copies are staged by the runtime"};
        }
        QAMD_ON_DEVICE(root());
        QAMD_TRY(bases_dev.alloc(n * sizeof(uint64_t)));
        QAMD_TRY(copy_in(bases_dev.ptr, base.data(), QAMD_MEM_HOST, n * sizeof(uint64_t), nullptr));
        return QAMD_OK;
    }

    ~Sharded() {
        for (H *h : shards)
            if (h) ops->free_store(h);
        for (auto &sl : slots) {
            if (sl->root_stream) {
                DeviceGuard g(root());
                (void)hipStreamSynchronize(sl->root_stream);
                (void)hipStreamDestroy(sl->root_stream);
            }
        }
    }

    qamd_status encode_query(const float *query, uint64_t qdim, qamd_mem mem, void *stream, ShardedQuery<H, Qy> **io) {
        std::unique_ptr<ShardedQuery<H, Qy>> fresh;
        ShardedQuery<H, Qy> *q = *io;
        if (!q) {
            fresh.reset(new ShardedQuery<H, Qy>);
            q = fresh.get();
            q->ops = ops;
            q->per_shard.assign(G(), nullptr);
        }
        if (q->per_shard.size() != G()) return fail(QAMD_ERR_ARGUMENTS, "query belongs to another sharded store");
        const int src_dev = mem == QAMD_MEM_DEVICE ? device_of(query) : -1;
        if (mem == QAMD_MEM_HOST && ops->host_encode_is_lazy(qdim)) {  // G host-side copies, no GPU work, no lanes
            for (uint32_t g = 0; g < G(); g++)
                QAMD_TRY(ops->encode_query(shards[g], query, qdim, mem, nullptr, &q->per_shard[g]));
            if (fresh) *io = fresh.release();
            return QAMD_OK;
        }
        QAMD_TRY(order_after_caller(mem, query, stream));
        Lease slot;
        QAMD_TRY(lease(slot));
        QAMD_TRY(pool.run([&](uint32_t g, Worker &w) -> qamd_status {
            const float *src = query;
            if (mem == QAMD_MEM_DEVICE && src_dev != w.device && qdim) {  // a device query lives on ONE GPU
                DevBuf &stage = slot->query_stage[g];
                if (stage.bytes < qdim * 4) QAMD_TRY(stage.alloc(qdim * 4));
                QAMD_HIP(hipMemcpyAsync(stage.ptr, query, qdim * 4, hipMemcpyDefault, w.stream));
                src = stage.template as<float>();
            }
            QAMD_TRY(qamd_set_device(w.device));
            // a host query has been copied out of the caller's buffer when encode_query returns, and the
            // query object orders its consumers after the encode kernel (ReadyEvent): nothing to wait for
            QAMD_TRY(ops->encode_query(shards[g], src, qdim, mem, w.stream, &q->per_shard[g]));
            // a device query: the caller's buffer and the slot's staging copy are in use until the kernel has run
            if (mem == QAMD_MEM_DEVICE) QAMD_TRY(wait_stream(w.stream));
            return QAMD_OK;
        }));
        if (fresh) *io = fresh.release();
        return QAMD_OK;
    }

    qamd_status check(const ShardedQuery<H, Qy> *q) const {
        if (!q || q->per_shard.size() != shards.size()) return fail(QAMD_ERR_ARGUMENTS, "null or foreign sharded query");
        for (Qy *p : q->per_shard)
            if (!p) return fail(QAMD_ERR_ARGUMENTS, "sharded query was never encoded");
        return QAMD_OK;
    }

    // out[base_g + i] = score_point(q, i of shard g).  Device output: on any one GPU.
    qamd_status score_all(const ShardedQuery<H, Qy> *q, float *out, qamd_mem out_mem, void *stream) {
        QAMD_TRY(check(q));
        if (count == 0) return QAMD_OK;
        if (!out) return fail(QAMD_ERR_ARGUMENTS, "out is null");
        const int out_dev = out_mem == QAMD_MEM_DEVICE ? device_of(out) : -1;
        QAMD_TRY(order_after_caller(out_mem, out, stream));
        Lease slot;
        QAMD_TRY(lease(slot));
        return pool.run([&](uint32_t g, Worker &w) -> qamd_status {
            const uint64_t n = base[g + 1] - base[g];
            if (n == 0) return QAMD_OK;
            float *dst = out + base[g];
            if (out_mem == QAMD_MEM_HOST || out_dev == w.device) {
                QAMD_TRY(ops->score_all(shards[g], q->per_shard[g], dst, out_mem, w.stream));
            } else {  // the per-shard score gather over xGMI: scan locally, peer-copy 4 B/row
                DevBuf &local = slot->local_scores[g];
                if (local.bytes < n * 4) QAMD_TRY(local.alloc(n * 4));
                QAMD_TRY(ops->score_all(shards[g], q->per_shard[g], local.template as<float>(), QAMD_MEM_DEVICE, w.stream));
                QAMD_HIP(hipMemcpyAsync(dst, local.ptr, n * 4, hipMemcpyDefault, w.stream));
            }
            return wait_stream(w.stream);
        });
    }

    qamd_status ensure_exchange(CallSlot &sl, uint32_t Q, uint32_t k) {
        const size_t need = (size_t)G() * Q * k * 8, res = (size_t)Q * k * 8;
        if (sl.gather_cap < need) {
            QAMD_TRY(sl.gather.alloc(need + need / 4));
            sl.gather_cap = need + need / 4;
        }
        if (sl.result_cap < res) {
            QAMD_TRY(sl.result.alloc(res + res / 4));
            sl.result_cap = res + res / 4;
        }
        return QAMD_OK;
    }

    // Runs the merge of [G][Q][k] gathered pairs on devices[0] and delivers [Q][k] ids / scores.
    qamd_status merge_and_deliver(CallSlot &sl, uint32_t Q, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                                  qamd_mem out_mem) {
        const uint32_t N = pow2_at_least(G() * k);
        if ((size_t)N * 8 > 64 * 1024) return fail(QAMD_ERR_ARGUMENTS, "shards x k = %u exceeds 8192 merge slots", G() * k);
        const uint32_t *g_ids = sl.gather.as<uint32_t>();
        const float *g_sc = reinterpret_cast<const float *>(g_ids + (size_t)G() * Q * k);
        const HostScratch hs = (out_mem == QAMD_MEM_HOST && Q == 1) ? host_scratch() : HostScratch{};
        uint32_t *ids_dev = out_mem == QAMD_MEM_DEVICE ? out_ids : hs.host ? hs.dev : sl.result.as<uint32_t>();
        float *sc_dev = out_mem == QAMD_MEM_DEVICE ? out_scores
                        : hs.host               ? reinterpret_cast<float *>(hs.dev + 1024)
                                                : reinterpret_cast<float *>(sl.result.as<uint32_t>() + (size_t)Q * k);
        hipLaunchKernelGGL(merge_topk_kernel, dim3(Q), dim3(1024), (size_t)N * 8, sl.root_stream, g_ids, g_sc,
                           (uint64_t)Q * k, bases_dev.as<uint64_t>(), G(), Q, k, largest, N, ids_dev, sc_dev);
        QAMD_HIP(hipGetLastError());
        if (out_mem == QAMD_MEM_HOST && !hs.host) {
            QAMD_HIP(hipMemcpyAsync(out_ids, ids_dev, (size_t)Q * k * 4, hipMemcpyDeviceToHost, sl.root_stream));
            QAMD_HIP(hipMemcpyAsync(out_scores, sc_dev, (size_t)Q * k * 4, hipMemcpyDeviceToHost, sl.root_stream));
        }
        QAMD_TRY(wait_stream(sl.root_stream));
        if (hs.host) {
            memcpy(out_ids, hs.host, (size_t)k * 4);
            memcpy(out_scores, hs.host + 1024, (size_t)k * 4);
        }
        return QAMD_OK;
    }

    // Shard g's [Q][k] pairs -> slot g of the gather buffer on devices[0] (after the shard's top-k).
    // `fn` runs the shard's own top-k with device outputs (ids, scores) on the lane's stream.
#ifdef QAMD_DEV
    static double now_us() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    }
#define QAMD_T(i) { const double t_ = now_us(); g_t[i] += t_ - t_last; t_last = t_; }
#else
#define QAMD_T(i)
#endif
    qamd_status topk_common(uint32_t Q, uint32_t k, int largest, uint32_t *out_ids, float *out_scores, qamd_mem out_mem,
                            void *stream,
                            const std::function<qamd_status(uint32_t, Worker &, uint32_t *, float *)> &fn) {
#ifdef QAMD_DEV
        static thread_local double g_t[8] = {0};
        static thread_local int g_n = 0;
        double t_last = now_us();
        if (++g_n % 300 == 0) {
            fprintf(stderr, "topk_common phases us: lease %.1f run %.1f waits %.1f merge+sync %.1f\n", g_t[0] / g_n, g_t[1] / g_n, g_t[2] / g_n, g_t[3] / g_n);
        }
#endif
        if (k == 0 || Q == 0) return QAMD_OK;
        if (!out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null output");
        if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
        if (out_mem == QAMD_MEM_DEVICE && device_of(out_ids) != root())
            return fail(QAMD_ERR_ARGUMENTS, "device outputs of a sharded top-k must live on devices[0] (%d)", root());
        QAMD_TRY(order_after_caller(out_mem, out_ids, stream));
        Lease slot;
        QAMD_TRY(lease(slot));
        QAMD_ON_DEVICE(root());
        QAMD_TRY(ensure_exchange(*slot.slot, Q, k));
        QAMD_T(0)
        uint32_t *g_ids = slot->gather.template as<uint32_t>();
        float *g_sc = reinterpret_cast<float *>(g_ids + (size_t)G() * Q * k);
        const size_t per = (size_t)Q * k;
        QAMD_TRY(pool.run([&](uint32_t g, Worker &w) -> qamd_status {
            uint32_t *slot_ids = g_ids + g * per;
            float *slot_sc = g_sc + g * per;
            if (w.device == root()) {  // same GPU: the shard's top-k writes its slot directly
                QAMD_TRY(fn(g, w, slot_ids, slot_sc));
            } else {  // k pairs per query over xGMI
                DevBuf &local = slot->local_pairs[g];
                if (local.bytes < per * 8) QAMD_TRY(local.alloc(per * 8));
                uint32_t *l_ids = local.template as<uint32_t>();
                float *l_sc = reinterpret_cast<float *>(l_ids + per);
                QAMD_TRY(fn(g, w, l_ids, l_sc));
                QAMD_HIP(hipMemcpyAsync(slot_ids, l_ids, per * 4, hipMemcpyDefault, w.stream));
                QAMD_HIP(hipMemcpyAsync(slot_sc, l_sc, per * 4, hipMemcpyDefault, w.stream));
            }
            return wait_stream(w.stream);
        }));
        QAMD_T(1)
        QAMD_T(2)
        const qamd_status mst = merge_and_deliver(*slot.slot, Q, k, largest, out_ids, out_scores, out_mem);
        QAMD_T(3)
        return mst;
    }

    qamd_status topk(const ShardedQuery<H, Qy> *q, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                     qamd_mem out_mem, void *stream) {
        QAMD_TRY(check(q));
        return topk_common(1, k, largest, out_ids, out_scores, out_mem, stream,
                           [&](uint32_t g, Worker &w, uint32_t *ids, float *sc) {
                               return ops->topk(shards[g], q->per_shard[g], k, largest, ids, sc, QAMD_MEM_DEVICE, w.stream);
                           });
    }
};

// A shard's slice of caller memory as something its own device can take: host memory and memory of
// the shard's device pass through; memory of ANOTHER device is staged whole (peer copy).
qamd_status shard_source(const void *src, qamd_mem mem, size_t bytes, int device, DevBuf &stage, hipStream_t s,
                         const void **out) {
    *out = src;
    if (mem == QAMD_MEM_HOST || bytes == 0 || device_of(src) == device) return QAMD_OK;
    QAMD_TRY(stage.alloc(bytes));
    QAMD_HIP(hipMemcpyAsync(stage.ptr, src, bytes, hipMemcpyDefault, s));
    QAMD_HIP(hipStreamSynchronize(s));
    *out = stage.ptr;
    return QAMD_OK;
}

// The merge on its own (qamd_topk_merge): what a multi-PROCESS caller runs after its all-gather.
qamd_status merge_lists(const uint32_t *ids, const float *scores, uint64_t shard_stride, const uint64_t *bases_host,
                        uint32_t G, uint32_t Q, uint32_t k, int largest, uint32_t *out_ids, float *out_scores,
                        qamd_mem out_mem, hipStream_t s) {
    if (G == 0 || Q == 0 || k == 0) return QAMD_OK;
    if (k > 1024) return fail(QAMD_ERR_ARGUMENTS, "topk: k=%u exceeds 1024", k);
    const uint32_t N = pow2_at_least(G * k);
    if ((size_t)N * 8 > 64 * 1024) return fail(QAMD_ERR_ARGUMENTS, "shards x k = %u exceeds 8192 merge slots", G * k);
    // bases ride in the calling thread's workspace, results (host output) too
    const size_t off_res = round_up((size_t)G * 8, 256), bytes = off_res + (out_mem == QAMD_MEM_HOST ? (size_t)Q * k * 8 : 0);
    char *ws = nullptr;
    QAMD_TRY(thread_ws_acquire(WS_SELECT, bytes, s, reinterpret_cast<void **>(&ws)));
    qamd_status st = QAMD_OK;
    if (hipMemcpyAsync(ws, bases_host, (size_t)G * 8, hipMemcpyHostToDevice, s) != hipSuccess)
        st = fail(QAMD_ERR_DEVICE, "topk merge: upload of the shard bases failed");
    uint32_t *ids_dev = out_mem == QAMD_MEM_DEVICE ? out_ids : reinterpret_cast<uint32_t *>(ws + off_res);
    float *sc_dev = out_mem == QAMD_MEM_DEVICE ? out_scores : reinterpret_cast<float *>(ws + off_res) + (size_t)Q * k;
    if (st == QAMD_OK) {
        hipLaunchKernelGGL(merge_topk_kernel, dim3(Q), dim3(1024), (size_t)N * 8, s, ids, scores, shard_stride,
                           reinterpret_cast<const uint64_t *>(ws), G, Q, k, largest, N, ids_dev, sc_dev);
        if (hipGetLastError() != hipSuccess) st = fail(QAMD_ERR_DEVICE, "topk merge: launch failed");
    }
    if (st == QAMD_OK && out_mem == QAMD_MEM_HOST) {
        st = copy_out(out_ids, QAMD_MEM_HOST, ids_dev, (size_t)Q * k * 4, s);
        if (st == QAMD_OK) st = copy_out(out_scores, QAMD_MEM_HOST, sc_dev, (size_t)Q * k * 4, s);
    } else if (st == QAMD_OK) {
        // the pageable `bases_host` must stay valid until the copy above has run
        if (hipStreamSynchronize(s) != hipSuccess) st = fail(QAMD_ERR_DEVICE, "topk merge: synchronisation failed");
    }
    thread_ws_release(WS_SELECT, s);
    return st;
}

// Query batches behind a sharded handle: one per-shard batch object each (the queries are replicated),
// encoded by the shards' workers; identical shapes for the three quantizers.
template <class B> struct ShardedBatch {
    std::vector<B *> per_shard;
    uint64_t n_queries = 0;
    void (*free_fn)(B *) = nullptr;
    ~ShardedBatch() {
        for (B *b : per_shard)
            if (b && free_fn) free_fn(b);
    }
};

template <class S, class B, class EncodeFn>
qamd_status sharded_encode_query_batch(S *h, const float *queries, uint64_t n_queries, uint64_t qdim, qamd_mem queries_mem,
                                       void *stream, ShardedBatch<B> *b, EncodeFn encode) {
    if (b->per_shard.size() != h->G()) return fail(QAMD_ERR_ARGUMENTS, "batch belongs to another sharded store");
    const int src_dev = queries_mem == QAMD_MEM_DEVICE ? device_of(queries) : -1;
    const size_t bytes = (size_t)n_queries * qdim * 4;
    QAMD_TRY(order_after_caller(queries_mem, queries, stream));
    typename S::Lease slot;
    QAMD_TRY(h->lease(slot));
    QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
        const float *src = queries;
        if (queries_mem == QAMD_MEM_DEVICE && src_dev != w.device && bytes) {  // device queries live on ONE GPU
            DevBuf &stage = slot->query_stage[g];
            if (stage.bytes < bytes) QAMD_TRY(stage.alloc(bytes));
            QAMD_HIP(hipMemcpyAsync(stage.ptr, queries, bytes, hipMemcpyDefault, w.stream));
            src = stage.template as<float>();
        }
        QAMD_TRY(encode(h->shards[g], src, n_queries, qdim, queries_mem, w.stream, &b->per_shard[g]));
        return wait_stream(w.stream);  // batch objects carry no ready event: finished before anyone scores with them
    }));
    b->n_queries = n_queries;
    return QAMD_OK;
}

bool never_lazy(uint64_t) { return false; }
const Ops<qamd_u8, qamd_u8_query> kU8Ops = {qamd_u8_encode_query, qamd_u8_query_free, qamd_u8_score_all, qamd_u8_topk,
                                            qamd_u8_free, u8_host_encode_is_lazy};
const Ops<qamd_bin, qamd_bin_query> kBinOps = {qamd_bin_encode_query, qamd_bin_query_free, qamd_bin_score_all,
                                               qamd_bin_topk, qamd_bin_free, never_lazy};
const Ops<qamd_pq, qamd_pq_query> kPqOps = {qamd_pq_encode_query, qamd_pq_query_free, qamd_pq_score_all, qamd_pq_topk,
                                            qamd_pq_free, never_lazy};

}  // namespace

struct qamd_u8_sharded : Sharded<qamd_u8, qamd_u8_query> {
    qamd_u8_metadata meta{};
};
struct qamd_u8_sharded_query : ShardedQuery<qamd_u8, qamd_u8_query> {};
struct qamd_u8_sharded_query_batch : ShardedBatch<qamd_u8_query_batch> {};
struct qamd_bin_sharded_query_batch : ShardedBatch<qamd_bin_query_batch> {};
struct qamd_pq_sharded_query_batch : ShardedBatch<qamd_pq_query_batch> {};
struct qamd_bin_sharded : Sharded<qamd_bin, qamd_bin_query> {
    qamd_vector_parameters vp{};
    int store = 0;
};
struct qamd_bin_sharded_query : ShardedQuery<qamd_bin, qamd_bin_query> {};
struct qamd_pq_sharded : Sharded<qamd_pq, qamd_pq_query> {
    qamd_vector_parameters vp{};
    uint64_t chunk_size = 0;
    std::vector<float> centroids;
};
struct qamd_pq_sharded_query : ShardedQuery<qamd_pq, qamd_pq_query> {};

extern "C" {

qamd_status qamd_topk_merge(const uint32_t *ids_dev, const float *scores_dev, uint64_t shard_stride,
                            const uint64_t *row_bases, uint32_t n_shards, uint32_t n_queries, uint32_t k, int largest,
                            uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (!ids_dev || !scores_dev || !row_bases || !out_ids || !out_scores) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    int dev = device_of(ids_dev);
    QAMD_ON_DEVICE(dev < 0 ? current_device() : dev);
    return merge_lists(ids_dev, scores_dev, shard_stride, row_bases, n_shards, n_queries, k, largest, out_ids, out_scores,
                       out_mem, as_stream(stream));
}

// ===================================================================================== u8
qamd_status qamd_u8_sharded_encode(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                                   const float *quantile, const float *alpha_offset, qamd_stop_fn stop, void *stop_user,
                                   const int *devices, uint32_t n_shards, void *stream,
                                   qamd_u8_sharded **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (vp->count > 0 && vp->dim > 0 && !data) return fail(QAMD_ERR_ARGUMENTS, "data is null");
    std::unique_ptr<qamd_u8_sharded> h(new qamd_u8_sharded);
    h->ops = &kU8Ops;
    QAMD_TRY(h->init(devices, n_shards, vp->count));
    QAMD_TRY(order_after_caller(data_mem, data, stream));
    const uint64_t dim = vp->dim;
    float ao[2] = {0.0f, 0.0f};
    if (alpha_offset) {
        ao[0] = alpha_offset[0];
        ao[1] = alpha_offset[1];
    } else if (vp->count) {
        // PASS 1 (encoded_vectors_u8.rs:57): every shard folds min/max over its own rows, the host
        // folds the G results -- min and max do not depend on the order.
        std::vector<float> mn(n_shards, 3.40282347e+38f), mx(n_shards, -3.40282347e+38f);
        QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
            const uint64_t n = h->base[g + 1] - h->base[g];
            if (n == 0 || dim == 0) return QAMD_OK;
            if (stop && stop(stop_user)) return fail(QAMD_ERR_STOPPED, "Stopped");
            return u8_minmax_range(data + h->base[g] * dim, data_mem, n, dim, w.stream, &mn[g], &mx[g]);
        }));
        float lo = 3.40282347e+38f, hi = -3.40282347e+38f;
        for (uint32_t g = 0; g < n_shards; g++) {
            if (mn[g] < lo) lo = mn[g];
            if (mx[g] > hi) hi = mx[g];
        }
        ao[0] = (hi - lo) / 127.0f;  // :228-232
        ao[1] = lo;
        if (quantile) {  // PASS 1b (:58-71): one device sees the whole sample
            bool found = false;
            float qmn = 0.0f, qmx = 0.0f;
            const int qdev = data_mem == QAMD_MEM_DEVICE ? device_of(data) : h->root();
            QAMD_ON_DEVICE(qdev < 0 ? h->root() : qdev);
            QAMD_TRY(u8_quantile_interval(data, data_mem, vp->count, dim, *quantile, nullptr, &found, &qmn, &qmx));
            if (found) {
                ao[0] = (qmx - qmn) / 127.0f;
                ao[1] = qmn;
            }
        }
    }
    // PASS 2 (:73-118): every shard quantizes its own rows with the global (alpha, offset).
    QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
        qamd_vector_parameters svp = *vp;
        svp.count = h->base[g + 1] - h->base[g];
        QAMD_TRY(qamd_set_device(w.device));
        qamd_u8_encoder *e = nullptr;
        QAMD_TRY(qamd_u8_encoder_begin(&svp, nullptr, vp->count ? ao : nullptr, stop, stop_user, w.stream, &e));
        qamd_status st = qamd_u8_encoder_push(e, data + h->base[g] * dim, svp.count, data_mem);
        if (st != QAMD_OK) {
            qamd_u8_encoder_abort(e);
            return st;
        }
        return qamd_u8_encoder_finish(e, &h->shards[g]);
    }));
    QAMD_TRY(qamd_u8_get_metadata(h->shards[0], &h->meta));
    if (vp->count && h->shards[0]->count == 0) {  // an empty first shard holds the empty-store metadata
        for (uint32_t g = 0; g < n_shards; g++)
            if (h->shards[g]->count) {
                QAMD_TRY(qamd_u8_get_metadata(h->shards[g], &h->meta));
                break;
            }
    }
    h->meta.vector_parameters = *vp;
    *out = h.release();
    return QAMD_OK;
}

qamd_status qamd_u8_sharded_from_rows(const uint8_t *rows, qamd_mem rows_mem, const qamd_u8_metadata *meta,
                                      const int *devices, uint32_t n_shards, void *stream,
                                   qamd_u8_sharded **out) {
    if (!meta || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_u8_sharded> h(new qamd_u8_sharded);
    h->ops = &kU8Ops;
    h->meta = *meta;
    QAMD_TRY(h->init(devices, n_shards, meta->vector_parameters.count));
    QAMD_TRY(order_after_caller(rows_mem, rows, stream));
    const uint64_t stride = meta->actual_dim + 4;
    QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
        qamd_u8_metadata sm = *meta;
        sm.vector_parameters.count = h->base[g + 1] - h->base[g];
        DevBuf stage;
        const void *src = nullptr;
        QAMD_TRY(shard_source(rows ? rows + h->base[g] * stride : nullptr, rows_mem, sm.vector_parameters.count * stride,
                              w.device, stage, w.stream, &src));
        QAMD_TRY(qamd_set_device(w.device));
        return qamd_u8_from_rows(static_cast<const uint8_t *>(src), rows_mem, &sm, w.stream, &h->shards[g]);
    }));
    *out = h.release();
    return QAMD_OK;
}

uint32_t qamd_u8_sharded_shard_count(const qamd_u8_sharded *h) { return h ? h->G() : 0; }
qamd_status qamd_u8_sharded_peer_access(const qamd_u8_sharded *h, uint32_t g, int *state, const char **reason) {
    if (!h || !state || g >= h->G()) return fail(QAMD_ERR_ARGUMENTS, "bad shard index");
    *state = h->peer[g].state;
    if (reason) *reason = h->peer[g].reason.c_str();
    return QAMD_OK;
}

qamd_status qamd_u8_sharded_shard(const qamd_u8_sharded *h, uint32_t g, const qamd_u8 **shard, uint64_t *row_begin,
                                  int *device) {
    if (!h || g >= h->G()) return fail(QAMD_ERR_ARGUMENTS, "no such shard");
    if (shard) *shard = h->shards[g];
    if (row_begin) *row_begin = h->base[g];
    if (device) *device = h->devices[g];
    return QAMD_OK;
}

qamd_status qamd_u8_sharded_get_metadata(const qamd_u8_sharded *h, qamd_u8_metadata *out) {
    if (!h || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    *out = h->meta;
    return QAMD_OK;
}

qamd_status qamd_u8_sharded_encode_query(qamd_u8_sharded *h, const float *query, uint64_t qdim, qamd_mem query_mem, void *stream,
                                   qamd_u8_sharded_query **query_io) {
    if (!h || !query_io || (!query && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    ShardedQuery<qamd_u8, qamd_u8_query> *q = *query_io;
    const bool fresh = q == nullptr;
    if (fresh) {
        auto *nq = new qamd_u8_sharded_query;
        nq->ops = &kU8Ops;
        nq->per_shard.assign(h->G(), nullptr);
        q = nq;
    }
    qamd_status st = h->encode_query(query, qdim, query_mem, stream, &q);
    if (st != QAMD_OK) {
        if (fresh) delete static_cast<qamd_u8_sharded_query *>(q);
        return st;
    }
    *query_io = static_cast<qamd_u8_sharded_query *>(q);
    return QAMD_OK;
}

void qamd_u8_sharded_query_free(qamd_u8_sharded_query *q) { delete q; }

qamd_status qamd_u8_sharded_score_all(qamd_u8_sharded *h, const qamd_u8_sharded_query *q, float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return h->score_all(q, out, out_mem, stream);
}

qamd_status qamd_u8_sharded_topk(qamd_u8_sharded *h, const qamd_u8_sharded_query *q, uint32_t k, int largest,
                                 uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return h->topk(q, k, largest, out_ids, out_scores, out_mem, stream);
}

qamd_status qamd_u8_sharded_encode_query_batch(qamd_u8_sharded *h, const float *queries, uint64_t n_queries, uint64_t qdim,
                                               qamd_mem queries_mem, void *stream,
                                   qamd_u8_sharded_query_batch **batch_io) {
    if (!h || !batch_io || (!queries && n_queries && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_u8_sharded_query_batch> fresh;
    qamd_u8_sharded_query_batch *b = *batch_io;
    if (!b) {
        fresh.reset(new qamd_u8_sharded_query_batch);
        b = fresh.get();
        b->per_shard.assign(h->G(), nullptr);
        b->free_fn = qamd_u8_query_batch_free;
    }
    QAMD_TRY(sharded_encode_query_batch(h, queries, n_queries, qdim, queries_mem, stream, b, qamd_u8_encode_query_batch));
    if (fresh) *batch_io = fresh.release();
    return QAMD_OK;
}

void qamd_u8_sharded_query_batch_free(qamd_u8_sharded_query_batch *b) { delete b; }

qamd_status qamd_u8_sharded_topk_batch(qamd_u8_sharded *h, const qamd_u8_sharded_query_batch *b, uint32_t k, int largest,
                                       uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (!h || !b || b->per_shard.size() != h->G()) return fail(QAMD_ERR_ARGUMENTS, "null or foreign argument");
    return h->topk_common((uint32_t)b->n_queries, k, largest, out_ids, out_scores, out_mem, stream,
                          [&](uint32_t g, Worker &w, uint32_t *ids, float *sc) {
                              return qamd_u8_topk_batch(h->shards[g], b->per_shard[g], k, largest, ids, sc, QAMD_MEM_DEVICE,
                                                        w.stream);
                          });
}

void qamd_u8_sharded_free(qamd_u8_sharded *h) { delete h; }

// ===================================================================================== binary
qamd_status qamd_bin_sharded_encode(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                                    qamd_bits_store store, qamd_stop_fn stop, void *stop_user, const int *devices,
                                    uint32_t n_shards, void *stream,
                                   qamd_bin_sharded **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (vp->count > 0 && vp->dim > 0 && !data) return fail(QAMD_ERR_ARGUMENTS, "data is null");
    std::unique_ptr<qamd_bin_sharded> h(new qamd_bin_sharded);
    h->ops = &kBinOps;
    h->vp = *vp;
    h->store = store;
    QAMD_TRY(h->init(devices, n_shards, vp->count));
    QAMD_TRY(order_after_caller(data_mem, data, stream));
    QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
        qamd_vector_parameters svp = *vp;
        svp.count = h->base[g + 1] - h->base[g];
        QAMD_TRY(qamd_set_device(w.device));
        qamd_bin_encoder *e = nullptr;
        QAMD_TRY(qamd_bin_encoder_begin(&svp, store, stop, stop_user, w.stream, &e));
        qamd_status st = qamd_bin_encoder_push(e, data + h->base[g] * vp->dim, svp.count, data_mem);
        if (st != QAMD_OK) {
            qamd_bin_encoder_abort(e);
            return st;
        }
        return qamd_bin_encoder_finish(e, &h->shards[g]);
    }));
    *out = h.release();
    return QAMD_OK;
}

qamd_status qamd_bin_sharded_from_rows(const uint8_t *rows, qamd_mem rows_mem, const qamd_vector_parameters *vp,
                                       qamd_bits_store store, const int *devices, uint32_t n_shards, void *stream,
                                   qamd_bin_sharded **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_bin_sharded> h(new qamd_bin_sharded);
    h->ops = &kBinOps;
    h->vp = *vp;
    h->store = store;
    QAMD_TRY(h->init(devices, n_shards, vp->count));
    QAMD_TRY(order_after_caller(rows_mem, rows, stream));
    const uint64_t stride = qamd_bin_quantized_vector_size(vp, store);
    QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
        qamd_vector_parameters svp = *vp;
        svp.count = h->base[g + 1] - h->base[g];
        DevBuf stage;
        const void *src = nullptr;
        QAMD_TRY(shard_source(rows ? rows + h->base[g] * stride : nullptr, rows_mem, svp.count * stride, w.device, stage,
                              w.stream, &src));
        QAMD_TRY(qamd_set_device(w.device));
        return qamd_bin_from_rows(static_cast<const uint8_t *>(src), rows_mem, &svp, store, w.stream, &h->shards[g]);
    }));
    *out = h.release();
    return QAMD_OK;
}

uint32_t qamd_bin_sharded_shard_count(const qamd_bin_sharded *h) { return h ? h->G() : 0; }
qamd_status qamd_bin_sharded_peer_access(const qamd_bin_sharded *h, uint32_t g, int *state, const char **reason) {
    if (!h || !state || g >= h->G()) return fail(QAMD_ERR_ARGUMENTS, "bad shard index");
    *state = h->peer[g].state;
    if (reason) *reason = h->peer[g].reason.c_str();
    return QAMD_OK;
}

qamd_status qamd_bin_sharded_shard(const qamd_bin_sharded *h, uint32_t g, const qamd_bin **shard, uint64_t *row_begin,
                                   int *device) {
    if (!h || g >= h->G()) return fail(QAMD_ERR_ARGUMENTS, "no such shard");
    if (shard) *shard = h->shards[g];
    if (row_begin) *row_begin = h->base[g];
    if (device) *device = h->devices[g];
    return QAMD_OK;
}

qamd_status qamd_bin_sharded_encode_query(qamd_bin_sharded *h, const float *query, uint64_t qdim, qamd_mem query_mem, void *stream,
                                   qamd_bin_sharded_query **query_io) {
    if (!h || !query_io || (!query && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    ShardedQuery<qamd_bin, qamd_bin_query> *q = *query_io;
    const bool fresh = q == nullptr;
    if (fresh) {
        auto *nq = new qamd_bin_sharded_query;
        nq->ops = &kBinOps;
        nq->per_shard.assign(h->G(), nullptr);
        q = nq;
    }
    qamd_status st = h->encode_query(query, qdim, query_mem, stream, &q);
    if (st != QAMD_OK) {
        if (fresh) delete static_cast<qamd_bin_sharded_query *>(q);
        return st;
    }
    *query_io = static_cast<qamd_bin_sharded_query *>(q);
    return QAMD_OK;
}

void qamd_bin_sharded_query_free(qamd_bin_sharded_query *q) { delete q; }

qamd_status qamd_bin_sharded_score_all(qamd_bin_sharded *h, const qamd_bin_sharded_query *q, float *out,
                                       qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return h->score_all(q, out, out_mem, stream);
}

qamd_status qamd_bin_sharded_topk(qamd_bin_sharded *h, const qamd_bin_sharded_query *q, uint32_t k, int largest,
                                  uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return h->topk(q, k, largest, out_ids, out_scores, out_mem, stream);
}

qamd_status qamd_bin_sharded_encode_query_batch(qamd_bin_sharded *h, const float *queries, uint64_t n_queries,
                                                uint64_t qdim, qamd_mem queries_mem, void *stream,
                                   qamd_bin_sharded_query_batch **batch_io) {
    if (!h || !batch_io || (!queries && n_queries && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_bin_sharded_query_batch> fresh;
    qamd_bin_sharded_query_batch *b = *batch_io;
    if (!b) {
        fresh.reset(new qamd_bin_sharded_query_batch);
        b = fresh.get();
        b->per_shard.assign(h->G(), nullptr);
        b->free_fn = qamd_bin_query_batch_free;
    }
    QAMD_TRY(sharded_encode_query_batch(h, queries, n_queries, qdim, queries_mem, stream, b, qamd_bin_encode_query_batch));
    if (fresh) *batch_io = fresh.release();
    return QAMD_OK;
}

void qamd_bin_sharded_query_batch_free(qamd_bin_sharded_query_batch *b) { delete b; }

qamd_status qamd_bin_sharded_topk_batch(qamd_bin_sharded *h, const qamd_bin_sharded_query_batch *b, uint32_t k, int largest,
                                        uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (!h || !b || b->per_shard.size() != h->G()) return fail(QAMD_ERR_ARGUMENTS, "null or foreign argument");
    return h->topk_common((uint32_t)b->n_queries, k, largest, out_ids, out_scores, out_mem, stream,
                          [&](uint32_t g, Worker &w, uint32_t *ids, float *sc) {
                              return qamd_bin_topk_batch(h->shards[g], b->per_shard[g], k, largest, ids, sc, QAMD_MEM_DEVICE,
                                                         w.stream);
                          });
}

void qamd_bin_sharded_free(qamd_bin_sharded *h) { delete h; }

// ===================================================================================== PQ
qamd_status qamd_pq_sharded_encode(const float *data, qamd_mem data_mem, const qamd_vector_parameters *vp,
                                   uint64_t chunk_size, const float *centroids, uint32_t max_kmeans_threads,
                                   qamd_stop_fn stop, void *stop_user, const int *devices, uint32_t n_shards, void *stream,
                                   qamd_pq_sharded **out) {
    if (!vp || !out) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (chunk_size == 0) return fail(QAMD_ERR_ARGUMENTS, "chunk_size must be > 0");
    if (vp->count > 0 && vp->dim > 0 && !data) return fail(QAMD_ERR_ARGUMENTS, "data is null");
    std::unique_ptr<qamd_pq_sharded> h(new qamd_pq_sharded);
    h->ops = &kPqOps;
    h->vp = *vp;
    h->chunk_size = chunk_size;
    QAMD_TRY(h->init(devices, n_shards, vp->count));
    QAMD_TRY(order_after_caller(data_mem, data, stream));
    if (centroids) {
        h->centroids.assign(centroids, centroids + (size_t)QAMD_PQ_CENTROIDS * vp->dim);
    } else {  // find_centroids (:278-342) once, on the device that holds the data (or devices[0])
        const int tdev = data_mem == QAMD_MEM_DEVICE ? device_of(data) : h->root();
        QAMD_ON_DEVICE(tdev < 0 ? h->root() : tdev);
        QAMD_TRY(pq_train_centroids(data, data_mem, vp, chunk_size, max_kmeans_threads, stop, stop_user, nullptr,
                                    h->centroids, nullptr, nullptr));
    }
    QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
        qamd_vector_parameters svp = *vp;
        svp.count = h->base[g + 1] - h->base[g];
        QAMD_TRY(qamd_set_device(w.device));
        qamd_pq_encoder *e = nullptr;
        QAMD_TRY(qamd_pq_encoder_begin(&svp, chunk_size, h->centroids.data(), max_kmeans_threads, stop, stop_user, w.stream,
                                       &e));
        qamd_status st = qamd_pq_encoder_push(e, data + h->base[g] * vp->dim, svp.count, data_mem);
        if (st != QAMD_OK) {
            qamd_pq_encoder_abort(e);
            return st;
        }
        return qamd_pq_encoder_finish(e, &h->shards[g]);
    }));
    *out = h.release();
    return QAMD_OK;
}

qamd_status qamd_pq_sharded_from_rows(const uint8_t *rows, qamd_mem rows_mem, const qamd_vector_parameters *vp,
                                      uint64_t chunk_size, const float *centroids, const int *devices, uint32_t n_shards, void *stream,
                                   qamd_pq_sharded **out) {
    if (!vp || !out || !centroids) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    if (chunk_size == 0) return fail(QAMD_ERR_ARGUMENTS, "chunk_size must be > 0");
    std::unique_ptr<qamd_pq_sharded> h(new qamd_pq_sharded);
    h->ops = &kPqOps;
    h->vp = *vp;
    h->chunk_size = chunk_size;
    h->centroids.assign(centroids, centroids + (size_t)QAMD_PQ_CENTROIDS * vp->dim);
    QAMD_TRY(h->init(devices, n_shards, vp->count));
    QAMD_TRY(order_after_caller(rows_mem, rows, stream));
    const uint64_t stride = qamd_pq_quantized_vector_size(vp, chunk_size);
    QAMD_TRY(h->pool.run([&](uint32_t g, Worker &w) -> qamd_status {
        qamd_vector_parameters svp = *vp;
        svp.count = h->base[g + 1] - h->base[g];
        DevBuf stage;
        const void *src = nullptr;
        QAMD_TRY(shard_source(rows ? rows + h->base[g] * stride : nullptr, rows_mem, svp.count * stride, w.device, stage,
                              w.stream, &src));
        QAMD_TRY(qamd_set_device(w.device));
        return qamd_pq_from_rows(static_cast<const uint8_t *>(src), rows_mem, &svp, chunk_size, h->centroids.data(),
                                 w.stream, &h->shards[g]);
    }));
    *out = h.release();
    return QAMD_OK;
}

uint32_t qamd_pq_sharded_shard_count(const qamd_pq_sharded *h) { return h ? h->G() : 0; }
qamd_status qamd_pq_sharded_peer_access(const qamd_pq_sharded *h, uint32_t g, int *state, const char **reason) {
    if (!h || !state || g >= h->G()) return fail(QAMD_ERR_ARGUMENTS, "bad shard index");
    *state = h->peer[g].state;
    if (reason) *reason = h->peer[g].reason.c_str();
    return QAMD_OK;
}

qamd_status qamd_pq_sharded_shard(const qamd_pq_sharded *h, uint32_t g, const qamd_pq **shard, uint64_t *row_begin,
                                  int *device) {
    if (!h || g >= h->G()) return fail(QAMD_ERR_ARGUMENTS, "no such shard");
    if (shard) *shard = h->shards[g];
    if (row_begin) *row_begin = h->base[g];
    if (device) *device = h->devices[g];
    return QAMD_OK;
}

qamd_status qamd_pq_sharded_get_centroids(const qamd_pq_sharded *h, float *centroids) {
    if (!h || !centroids) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    memcpy(centroids, h->centroids.data(), h->centroids.size() * sizeof(float));
    return QAMD_OK;
}

qamd_status qamd_pq_sharded_encode_query(qamd_pq_sharded *h, const float *query, uint64_t qdim, qamd_mem query_mem, void *stream,
                                   qamd_pq_sharded_query **query_io) {
    if (!h || !query_io || (!query && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    ShardedQuery<qamd_pq, qamd_pq_query> *q = *query_io;
    const bool fresh = q == nullptr;
    if (fresh) {
        auto *nq = new qamd_pq_sharded_query;
        nq->ops = &kPqOps;
        nq->per_shard.assign(h->G(), nullptr);
        q = nq;
    }
    qamd_status st = h->encode_query(query, qdim, query_mem, stream, &q);
    if (st != QAMD_OK) {
        if (fresh) delete static_cast<qamd_pq_sharded_query *>(q);
        return st;
    }
    *query_io = static_cast<qamd_pq_sharded_query *>(q);
    return QAMD_OK;
}

void qamd_pq_sharded_query_free(qamd_pq_sharded_query *q) { delete q; }

qamd_status qamd_pq_sharded_score_all(qamd_pq_sharded *h, const qamd_pq_sharded_query *q, float *out, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return h->score_all(q, out, out_mem, stream);
}

qamd_status qamd_pq_sharded_topk(qamd_pq_sharded *h, const qamd_pq_sharded_query *q, uint32_t k, int largest,
                                 uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (!h) return fail(QAMD_ERR_ARGUMENTS, "null handle");
    return h->topk(q, k, largest, out_ids, out_scores, out_mem, stream);
}

qamd_status qamd_pq_sharded_encode_query_batch(qamd_pq_sharded *h, const float *queries, uint64_t n_queries, uint64_t qdim,
                                               qamd_mem queries_mem, void *stream,
                                   qamd_pq_sharded_query_batch **batch_io) {
    if (!h || !batch_io || (!queries && n_queries && qdim)) return fail(QAMD_ERR_ARGUMENTS, "null argument");
    std::unique_ptr<qamd_pq_sharded_query_batch> fresh;
    qamd_pq_sharded_query_batch *b = *batch_io;
    if (!b) {
        fresh.reset(new qamd_pq_sharded_query_batch);
        b = fresh.get();
        b->per_shard.assign(h->G(), nullptr);
        b->free_fn = qamd_pq_query_batch_free;
    }
    QAMD_TRY(sharded_encode_query_batch(h, queries, n_queries, qdim, queries_mem, stream, b, qamd_pq_encode_query_batch));
    if (fresh) *batch_io = fresh.release();
    return QAMD_OK;
}

void qamd_pq_sharded_query_batch_free(qamd_pq_sharded_query_batch *b) { delete b; }

qamd_status qamd_pq_sharded_topk_batch(qamd_pq_sharded *h, const qamd_pq_sharded_query_batch *b, uint32_t k, int largest,
                                       uint32_t *out_ids, float *out_scores, qamd_mem out_mem, void *stream) {
    if (!h || !b || b->per_shard.size() != h->G()) return fail(QAMD_ERR_ARGUMENTS, "null or foreign argument");
    return h->topk_common((uint32_t)b->n_queries, k, largest, out_ids, out_scores, out_mem, stream,
                          [&](uint32_t g, Worker &w, uint32_t *ids, float *sc) {
                              return qamd_pq_topk_batch(h->shards[g], b->per_shard[g], k, largest, ids, sc, QAMD_MEM_DEVICE,
                                                        w.stream);
                          });
}

void qamd_pq_sharded_free(qamd_pq_sharded *h) { delete h; }

}  // extern "C"
