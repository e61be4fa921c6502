// nrldpc_capi.hip -- the C ABI of include/nrldpc.h over the gfx950 kernels.
//
// Host-pointer entry points stage through device buffers owned by the handle; *_dev entry points
// launch directly on caller-owned device memory.  No CPU fallback exists anywhere in this library:
// every decode/encode is a HIP kernel launch, and any HIP failure is reported as NRLDPC_ERR_HIP.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "nrldpc.h"
#include "nrldpc_hostpath.h"
#include "nrldpc_kernels.h"
#include "nrldpc_host_quant.h"

#ifndef NRLDPC_HOST_SPIN_US_DEFAULT
#define NRLDPC_HOST_SPIN_US_DEFAULT 300 // see HostPool::spin_us
#endif
#include "nrldpc_sched.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hipfail(hipError_t e, const char* what) {
    return fail(NRLDPC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return hipfail(_e, #expr); \
    } while (0)

// extern "C" bodies must not let C++ exceptions (std::bad_alloc from the staging vectors, std::system_error from the
// copy threads) cross the ABI: they become return codes.
#define NRLDPC_API_BEGIN try {
#define NRLDPC_API_END                                                                                   \
    }                                                                                                    \
    catch (const std::bad_alloc&) { return fail(NRLDPC_ERR_NOMEM, "host allocation failed"); }           \
    catch (const std::exception& ex) { return fail(NRLDPC_ERR_HIP, std::string("internal error: ") + ex.what()); } \
    catch (...) { return fail(NRLDPC_ERR_HIP, "internal error"); }

// Makes the handle's device current for the duration of a call and puts the caller's device back afterwards: a
// caller that drives several GPUs from one thread (or whose framework tracks its own current device) is not disturbed.
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define DEVICE_SCOPE(h)                                  \
    DeviceScope _scope((h)->cfg.device_id);              \
    if (_scope.err != hipSuccess) return hipfail(_scope.err, "hipSetDevice")

template <class T> struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t reserve(size_t want) {
        if (want <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// Pinned host memory for the host-pointer entry points (DMA at PCIe rate; pageable memory is not).
struct PinBuf {
    char* p = nullptr;
    size_t n = 0;
    hipError_t reserve(size_t want) {
        if (want <= n) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault);
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

// A few persistent host threads that move (and quantise / narrow) the caller's pageable arrays into / out of the pinned
// staging buffers: one core does not keep up with PCIe.  On a two-socket host they have to run on the NUMA node that
// HOLDS the caller's array: threads on the other socket read it over the inter-socket links (~120 GB/s for everyone
// together, measured: 855 MB of MATLAB doubles took 7 ms however many threads there were), and the node the calling thread
// happens to run on says little about where its array was first touched.  So the pool asks the kernel where the array's
// pages are (move_pages with no target nodes = query) and moves its threads there, per call.
static int numa_cpus(std::vector<cpu_set_t>* out) { // CPUs of node 0, 1, ... (/sys/devices/system/node/node*/cpulist)
    out->clear();
    for (int node = 0; node < 64; ++node) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
        FILE* f = fopen(path, "r");
        if (!f) break;
        char buf[4096];
        const bool ok = fgets(buf, sizeof buf, f) != nullptr;
        fclose(f);
        cpu_set_t set;
        CPU_ZERO(&set);
        if (ok)
            for (char* p = buf; *p && *p != '\n';) { // "0-63,128-191"
                char* e;
                const long a = strtol(p, &e, 10);
                long b = a;
                if (*e == '-') b = strtol(e + 1, &e, 10);
                for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, &set);
                p = (*e == ',') ? e + 1 : e;
                if (e == p && *e != ',') break;
            }
        out->push_back(set);
    }
    return (int)out->size();
}

// NUMA node that holds most of [p, p + bytes) (sampled), -1 if unknown
static int numa_node_of(const void* p, size_t bytes) {
#ifdef SYS_move_pages
    const long page = sysconf(_SC_PAGESIZE);
    if (page <= 0 || bytes == 0) return -1;
    constexpr int NSAMP = 9;
    void* addr[NSAMP];
    int status[NSAMP];
    const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes - 1;
    for (int i = 0; i < NSAMP; ++i) {
        addr[i] = reinterpret_cast<void*>((lo + (uintptr_t)((double)(hi - lo) * i / (NSAMP - 1))) & ~(uintptr_t)(page - 1));
        status[i] = -1;
    }
    if (syscall(SYS_move_pages, 0, (unsigned long)NSAMP, addr, nullptr, status, 0) != 0) return -1;
    int votes[64] = {};
    for (int i = 0; i < NSAMP; ++i)
        if (status[i] >= 0 && status[i] < 64) ++votes[status[i]];
    int best = -1;
    for (int n = 0; n < 64; ++n)
        if (votes[n] > 0 && (best < 0 || votes[n] > votes[best])) best = n;
    return best;
#else
    (void)p; (void)bytes;
    return -1;
#endif
}

// CPUs this process may actually use: the cgroup CPU quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) when
// there is one, else what the scheduler reports.
static int usable_cpus() {
    long quota = -1, period = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(f);
    } else {
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%ld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%ld", &period) != 1) period = 0; fclose(g); }
    }
    const unsigned hc = std::thread::hardware_concurrency();
    int n = hc ? (int)hc : 8;
    if (quota > 0 && period > 0) n = std::min(n, (int)((quota + period - 1) / period));
    return std::max(1, n);
}

class HostPool {
    // Fork/join on every chunk of a pipelined call: a dozen jobs of 0.1-0.4 ms each within a few milliseconds.  Waking
    // sixteen sleeping threads through a condition variable costs 0.15-0.2 ms per job (measured: 13 chunks x 2 jobs made
    // 4 ms of a 12 ms call), so workers poll the generation counter for a short while after a job before they go back
    // to sleep, and the caller polls the completion counter.
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::function<void(int, int)> job_;
    std::atomic<int> caller_sleeps_{0};
    std::atomic<unsigned> gen_{0};
    std::atomic<int> pending_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> stop_{false};

    // NRLDPC_HOST_SPIN_US: how long an idle copy thread polls for the next job before it sleeps -- WHILE A CALL IS IN FLIGHT
    // (hot_: set by the chunked host path for its duration); between calls the threads sleep at once.  Waking fifteen sleeping
    // threads costs 50-150 us per job, a chunked call is 11 jobs, and round 6 measured what that adds up to: the copy / quantise
    // phase of a 4096-codeword call of doubles takes 4.4 ms with polling threads and 6.0-8.7 ms with sleeping ones (fp16: 1.0 against
    // 2.0-3.0; profiles/r06_host_copy_thread_polling.txt) -- most of what rounds 4 and 5 read as "the host's DRAM drifts".
    // Through round 5 the default was 0: a container with a CPU quota (the MI355X boxes of this build: 256 CPUs visible,
    // cpu.max = 16) charges polling like work, and a throttled process loses 20-30 ms at a time.  Polling only inside a call and
    // only for a bounded time after each job keeps that charge to the gaps between the jobs of a copy-bound call (tens of
    // microseconds each); a device-bound call's long gaps still put the threads to sleep.
    static int spin_us() {
        static const int v = getenv("NRLDPC_HOST_SPIN_US") ? atoi(getenv("NRLDPC_HOST_SPIN_US")) : NRLDPC_HOST_SPIN_US_DEFAULT;
        return v;
    }
    std::atomic<bool> hot_{false};
    static void relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }

public:
    std::vector<cpu_set_t> nodes_;
    cpu_set_t proc_mask_; // affinity of the creating thread = what the host process allows
    int node_ = -1;

    explicit HostPool(int n) {
        CPU_ZERO(&proc_mask_);
        if (sched_getaffinity(0, sizeof(cpu_set_t), &proc_mask_) != 0)
            for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &proc_mask_);
        if (getenv("NRLDPC_HOST_NO_PIN") == nullptr) (void)numa_cpus(&nodes_);
        for (int i = 0; i < n; ++i)
            th_.emplace_back([this, i, n] {
                unsigned seen = 0;
                for (;;) {
                    const auto t0 = std::chrono::steady_clock::now();
                    int polls = 0;
                    while (gen_.load(std::memory_order_acquire) == seen && !stop_.load(std::memory_order_relaxed)) {
                        relax();
                        const bool hot = hot_.load(std::memory_order_relaxed);
                        if (!hot || (((++polls & 255) == 0 || spin_us() == 0) && std::chrono::steady_clock::now() - t0 >= std::chrono::microseconds(spin_us()))) {
                            std::unique_lock<std::mutex> lk(m_);
                            sleepers_.fetch_add(1);
                            cv_.wait(lk, [&] { return stop_.load() || gen_.load() != seen; });
                            sleepers_.fetch_sub(1);
                        }
                    }
                    if (stop_.load()) return;
                    seen = gen_.load(std::memory_order_acquire);
                    job_(i, n); // published before the generation moved; the next run() starts only after this one is done
                    if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1 && caller_sleeps_.load() != 0) {
                        std::lock_guard<std::mutex> lk(m_);
                        cv_done_.notify_all();
                    }
                }
            });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true);
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    // a chunked call is in flight (the workers poll between its jobs) / over (they sleep)
    void set_hot(bool on) { hot_.store(on, std::memory_order_relaxed); }
    // Run the workers on the NUMA node that holds the caller's array (no-op when it is unknown or unchanged).
    void follow(const void* p, size_t bytes) {
        if (nodes_.size() < 2) return;
        const int node = numa_node_of(p, bytes);
        if (node < 0 || node >= (int)nodes_.size() || node == node_) return;
        // never widen what the host process was started with (taskset / numactl / a MATLAB worker's mask): the workers
        // move to the node's CPUs that the process itself may run on, and stay where they are when there is none
        cpu_set_t want;
        CPU_AND(&want, &nodes_[node], &proc_mask_);
        node_ = node;
        if (CPU_COUNT(&want) == 0) return;
        for (auto& t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(cpu_set_t), &want);
    }
    // f(worker, nworkers) on every worker; returns when all are done
    void run(std::function<void(int, int)> f) {
        start(std::move(f));
        finish();
    }
    // the same in two halves: the caller's thread is free between them (it takes finished chunks' results out meanwhile)
    void start(std::function<void(int, int)> f) {
        job_ = std::move(f);
        pending_.store((int)th_.size(), std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_); // a worker between its last poll and cv_.wait holds this lock
            gen_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load() > 0) cv_.notify_all();
    }
    bool done() const { return pending_.load(std::memory_order_acquire) == 0; }
    void finish() {
        // poll for the usual sub-millisecond job; on an oversubscribed host (several handles decoding at once, each with
        // its own pool) give the core away instead
        const auto t0 = std::chrono::steady_clock::now();
        int polls = 0;
        while (pending_.load(std::memory_order_acquire) != 0) {
            relax();
            if (((++polls & 255) == 0 || spin_us() == 0) && std::chrono::steady_clock::now() - t0 >= std::chrono::microseconds(5 * spin_us())) {
                std::unique_lock<std::mutex> lk(m_);
                caller_sleeps_.store(1);
                cv_done_.wait_for(lk, std::chrono::milliseconds(2), [&] { return pending_.load() == 0; });
                caller_sleeps_.store(0);
            }
        }
    }
    // dst[i] = int8 grid value of src[i] (nrldpc_host_quant.h) for n elements; true when a -inf was met
    bool quantise(int8_t* dst, const void* src, size_t n, int kind, float scale) {
        std::atomic<int> neg{0};
        const size_t es = kind == NRLDPC_HQ_F64 ? 8 : kind == NRLDPC_HQ_F16 ? 2 : 4;
        run([=, &neg](int w, int nw) {
            const size_t per = ((n + nw - 1) / nw + 63) & ~(size_t)63, lo = std::min(n, per * w), hi = std::min(n, lo + per);
            if (hi > lo && nrldpc_quantise_i8(dst + lo, static_cast<const char*>(src) + lo * es, hi - lo, kind, scale))
                neg.store(1, std::memory_order_relaxed);
        });
        return neg.load() != 0;
    }
    // the same for the first `act` elements of each of n_rows rows that are `pitch` elements apart in src; dst is COMPACT
    // ([n_rows][act]); the compact range [c_lo, c_hi) only (multiples of 64 or the end) -- the rest of a row is known to be zero
    // and is neither read nor sent (nrldpc.h "Active layers")
    // (two halves, see start / finish: quantise_rows_start, then quantise_rows_finish() = "a -inf was met")
    std::atomic<int> neg_{0};
    void quantise_rows_start(int8_t* dst, const void* src, size_t act, size_t pitch, size_t c_lo, size_t c_hi, int kind, float scale) {
        neg_.store(0, std::memory_order_relaxed);
        const size_t es = kind == NRLDPC_HQ_F64 ? 8 : kind == NRLDPC_HQ_F16 ? 2 : 4, n = c_hi - c_lo;
        start([=](int w, int nw) {
            const size_t per = ((n + nw - 1) / nw + 63) & ~(size_t)63, lo = c_lo + std::min(n, per * w), hi = std::min(c_hi, lo + per);
            for (size_t pos = lo; pos < hi;) {
                const size_t r = pos / act, in_row = pos - r * act, len = std::min(hi - pos, act - in_row);
                if (nrldpc_quantise_i8(dst + pos, static_cast<const char*>(src) + (r * pitch + in_row) * es, len, kind, scale))
                    neg_.store(1, std::memory_order_relaxed);
                pos += len;
            }
        });
    }
    bool quantise_rows_finish() {
        finish();
        return neg_.load() != 0;
    }
    // NRLDPC_LAYERS_AUTO: highest block in [first, nblocks) holding anything but +-0 / NaN in any of n_cw codewords (nrldpc_host_quant.h)
    int top_block(const void* src, int kind, size_t n_cw, int Z, int nblocks, int first) {
        int best = first - 1;
        run([=, &best](int w, int nw) { nrldpc_top_block(src, kind, n_cw, (size_t)w, (size_t)nw, Z, nblocks, first, &best); });
        return best;
    }
    // dst[i] = src[i] for n bytes, or float(dst) = double(src) for n elements when narrow
    void move(void* dst, const void* src, size_t n, bool narrow) {
        run([=](int w, int nw) {
            const size_t per = ((n + nw - 1) / nw + 63) & ~(size_t)63, lo = std::min(n, per * w), hi = std::min(n, lo + per);
            if (narrow) {
                const double* d = static_cast<const double*>(src);
                float* o = static_cast<float*>(dst);
                for (size_t i = lo; i < hi; ++i) o[i] = (float)d[i];
            } else if (hi > lo) {
                memcpy(static_cast<char*>(dst) + lo, static_cast<const char*>(src) + lo, hi - lo);
            }
        });
    }
};

} // namespace

struct nrldpc_codec {
    nrldpc_cfg cfg;
    nrldpc::Schedule sched;
    float alpha = 0.75f, beta = 0.0f; // beta in LLR units; the kernels get beta*scale (the rule of `rule_layers` rows)
    int scale = 8;
    // active layers (ABI revision 5): a property of the call.  layers = what the next call uses: 4..rows, or NRLDPC_LAYERS_AUTO;
    // last_layers = what the last call ran with; rule_auto: cfg.alpha == 0, the check-node rule follows the count in use
    int layers = 0, last_layers = 0, rule_layers = 0;
    bool rule_auto = false;
    // batch counters of the parity-stop kernels that refill their codeword slots (DecArgs::work): a ring, one per launch in flight
    DevBuf<int32_t> d_work;
    unsigned work_seq = 0;
    double host_phases[10] = {}; // of the last chunked host-pointer call (nrldpc_last_host_phases); [0] = 0: none yet
    static constexpr unsigned kWorkRing = 64;
    hipEvent_t work_done[kWorkRing] = {}; // recorded behind the launch that took the slot: a slot is reused only once that launch is over
    DevBuf<int32_t> d_best; // NRLDPC_LAYERS_AUTO on device pointers: the pre-pass kernel's result ...
    PinBuf pin_best;        // ... and where the host reads it
    // device tables
    DevBuf<int32_t> d_rot;
    DevBuf<uint32_t> d_crc; // x^(crc_bits-1-i) mod g, i < crc_bits: the CRC-aided stop's table (early_term = 2)
    DevBuf<uint16_t> d_row_ptr, d_shift;
    DevBuf<uint8_t> d_col;
    // encoder solve order
    int p0_shift = 0, step_row[3] = {0, 0, 0}, step_col[3] = {0, 0, 0}, step_shift[3] = {0, 0, 0};
    int step_nk[3] = {0, 0, 0}, step_kcol[3][3] = {}, step_kshift[3][3] = {}; // already-known core blocks in that row
    // host-entry staging
    DevBuf<char> s_llr;
    DevBuf<int8_t> s_q; // int8 chunks of the pipelined host path, one region per slot
    DevBuf<uint8_t> s_hard, s_bits, s_pk; // s_pk: bit-packed hard decisions (nrldpc_decode_packed)
    DevBuf<int32_t> s_iters;
    DevBuf<float> s_app;
    std::vector<float> h_narrow;
    // argument blocks + workgroup prefix tables of nrldpc_decode_multi_dev: a ring of slots (pinned host copy, device
    // copy, completion event), so that calls in flight on different streams never share a table
    struct MultiSlot { PinBuf pin; DevBuf<char> dev; hipEvent_t done = nullptr; bool used = false; };
    static constexpr int kMultiSlots = 4;
    MultiSlot multi[kMultiSlots];
    int multi_next = 0;
    // side streams for the launches of one nrldpc_decode_multi_dev call: each launch is bound by its slowest
    // workgroups (25 iterations of a codeword that never converges), not by throughput, so the launches of the two
    // base graphs overlap almost perfectly (fork / join with events around the caller's stream)
    static constexpr int kSide = 7;
    hipStream_t side[kSide] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kSide] = {};
    // pipelined host path (large batches): two pinned slots, two streams, copy threads
    static constexpr int kSlots = 4; // chunks the host may run ahead of the device
    PinBuf pin_in[kSlots], pin_out[kSlots], pin_it[kSlots];
    hipStream_t xs[2] = {nullptr, nullptr};
    hipEvent_t xdone[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    HostPool* pool = nullptr;
    int host_threads_hint = 0; // > 0: copy threads of this handle (nrldpc_pool_* shares the cores between its handles)
    // timing
    bool timing = false, have_time = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

size_t llr_elem_bytes(int dtype) { return dtype == NRLDPC_LLR_F16 ? 2 : 4; }

// Substitution order for the core-parity blocks (same derivation as the oracle's encode_one, from the
// dual-diagonal entries at get_3gpp_base_graph.m:30-31,48-50,68-69,87-88 / :339-340,349-350,356-358,367-368).
bool derive_encoder_order(nrldpc_codec* h) {
    const nrldpc::Schedule& s = h->sched;
    const nrldpc::BaseGraph& g = s.g;
    const int kb = g.kb;
    int cnt[4] = {0, 0, 0, 0}, sh0[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int e = g.row_ptr[i]; e < g.row_ptr[i + 1]; ++e)
            if (g.col[e] == kb) { cnt[i] = 1; sh0[i] = s.shift[e]; }
    int a = -1;
    for (int i = 0; i < 4; ++i) {
        if (!cnt[i]) continue;
        int n = 0;
        for (int k = 0; k < 4; ++k) n += (cnt[k] && sh0[k] == sh0[i]);
        if (n & 1) a = sh0[i];
    }
    if (a < 0) return false;
    h->p0_shift = a;
    bool known[4] = {true, false, false, false};
    for (int st = 0; st < 3; ++st) {
        bool found = false;
        for (int i = 0; i < 4 && !found; ++i) {
            int unk = -1, nunk = 0, ush = 0;
            for (int e = g.row_ptr[i]; e < g.row_ptr[i + 1]; ++e) {
                int c = g.col[e] - kb;
                if (c >= 0 && c < 4 && !known[c]) { unk = c; ush = s.shift[e]; ++nunk; }
            }
            if (nunk == 1) {
                h->step_row[st] = i; h->step_col[st] = unk; h->step_shift[st] = ush;
                int nk = 0;
                for (int e = g.row_ptr[i]; e < g.row_ptr[i + 1]; ++e) {
                    const int c = g.col[e] - kb;
                    if (c >= 0 && c < 4 && c != unk) {
                        if (nk == 3) return false;
                        h->step_kcol[st][nk] = c; h->step_kshift[st][nk] = s.shift[e]; ++nk;
                    }
                }
                h->step_nk[st] = nk;
                known[unk] = true; found = true;
            }
        }
        if (!found) return false;
    }
    return true;
}

void begin_timing(nrldpc_codec* h, hipStream_t s) {
    if (h->timing) (void)hipEventRecord(h->ev0, s);
}
void end_timing(nrldpc_codec* h, hipStream_t s) {
    if (h->timing) { (void)hipEventRecord(h->ev1, s); h->have_time = true; }
}

// the check-node rule of a call that decodes nl rows: the handle's own (explicit cfg.alpha), or -- cfg.alpha == 0 -- the library's
// rule for that count, on the grid nrldpc_create puts beta on
void rule_for(const nrldpc_codec* h, int nl, float* alpha, float* beta) {
    *alpha = h->alpha; *beta = h->beta;
    if (!h->rule_auto || nl == h->rule_layers) return;
    float a, b;
    nrldpc::default_rule(h->sched.g.bg, nl, &a, &b);
    *alpha = a;
    *beta = std::nearbyint(2.0f * b * (float)h->scale) / (2.0f * (float)h->scale);
}

// rows a block index found by the NRLDPC_LAYERS_AUTO scans stands for: block c >= kb + 4 is row c - kb's extension column
int layers_of_block(const nrldpc::Schedule& s, int top) { return std::max(4, top - s.g.kb + 1); }

// nl: active layer count of this call (4..rows, resolved by the caller)
nrldpc::DecArgs make_dec_args(const nrldpc_codec* h, const void* d_llr, int batch, uint8_t* d_hard, int32_t* d_iters,
                              float* d_app, int nl, int llr_kind = -1) {
    const nrldpc::Schedule& s = h->sched;
    nrldpc::DecArgs a;
    memset(&a, 0, sizeof a);
    a.llr = d_llr; a.hard = d_hard; a.iters = d_iters; a.app = d_app;
    a.rot = h->d_rot.p;
    a.batch = batch; a.Z = s.Z; a.n_layers = nl; a.max_iter = h->cfg.max_iter; a.ncw = s.ncw; a.sbw = s.sbw;
    a.early_term = h->cfg.early_term ? 1 : 0;
    if (h->cfg.early_term == 2) { a.crc_tab = h->d_crc.p; a.crc_bits = h->cfg.crc_bits; }
    a.need_ext = (a.early_term || d_app) ? 1 : 0;
    a.llr_kind = llr_kind >= 0 ? llr_kind : (h->cfg.llr_dtype == NRLDPC_LLR_F16) ? NRLDPC_K_F16 : NRLDPC_K_F32;
    float alpha, beta;
    rule_for(h, nl, &alpha, &beta);
    a.alpha = alpha; a.scale = (float)h->scale; a.inv_scale = 1.0f / (float)h->scale;
    a.beta = beta * (float)h->scale;
    return a;
}

// NRLDPC_LAYERS_AUTO on device pointers, first half: the pre-pass kernel over `batch` codewords at d_llr and the copy of its
// result into the handle's pinned word, all on `stream`; the caller synchronises the stream and calls auto_layers_read
int auto_layers_enqueue(nrldpc_codec* h, const void* d_llr, int batch, hipStream_t stream, int llr_kind = -1) {
    const nrldpc::Schedule& s = h->sched;
    HIP_TRY(h->d_best.reserve(1));
    HIP_TRY(h->pin_best.reserve(64));
    const int first = s.g.kb + 4, kind = llr_kind >= 0 ? llr_kind : (h->cfg.llr_dtype == NRLDPC_LLR_F16) ? NRLDPC_K_F16 : NRLDPC_K_F32;
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_best.p), first - 1, 1, stream));
    HIP_TRY(nrldpc::launch_top_block(d_llr, kind, batch, s.Z, s.g.ncols, first, h->d_best.p, stream));
    HIP_TRY(hipMemcpyAsync(h->pin_best.p, h->d_best.p, 4, hipMemcpyDeviceToHost, stream));
    return NRLDPC_OK;
}
int auto_layers_read(const nrldpc_codec* h) { return layers_of_block(h->sched, *reinterpret_cast<const int32_t*>(h->pin_best.p)); }

// the layer count of a device-pointer call: the handle's, or -- NRLDPC_LAYERS_AUTO -- read off the data (synchronises `stream`)
int resolve_layers_dev(nrldpc_codec* h, const void* d_llr, int batch, hipStream_t stream, int* nl) {
    if (h->layers != NRLDPC_LAYERS_AUTO) { *nl = h->layers; return NRLDPC_OK; }
    int rc = auto_layers_enqueue(h, d_llr, batch, stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(stream));
    *nl = auto_layers_read(h);
    return NRLDPC_OK;
}

// The batch counter of the next parity-stop launch (the persistent kernels' "next codeword", nrldpc_decode_z64p.h), from a ring of
// 64 per handle.  A slot carries an event recorded behind the launch that used it; a slot whose launch is still running -- more than
// 64 refilling launches in flight over several streams -- is NOT reset under it: this launch then runs without refilling
// (null: every workgroup decodes its own codewords), as it does when the ring cannot be allocated.
int32_t* next_work(nrldpc_codec* h, unsigned* slot) {
    if (h->d_work.reserve(nrldpc_codec::kWorkRing) != hipSuccess) return nullptr;
    const unsigned i = h->work_seq % nrldpc_codec::kWorkRing;
    if (h->work_done[i] && hipEventQuery(h->work_done[i]) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (!h->work_done[i] && hipEventCreateWithFlags(&h->work_done[i], hipEventDisableTiming) != hipSuccess) { h->work_done[i] = nullptr; return nullptr; }
    ++h->work_seq;
    *slot = i;
    return h->d_work.p + i;
}

int decode_launch(nrldpc_codec* h, const void* d_llr, int batch, uint8_t* d_hard, int32_t* d_iters, float* d_app,
                  hipStream_t stream, int nl, int llr_kind = -1) {
    const nrldpc::Schedule& s = h->sched;
    nrldpc::DecArgs a = make_dec_args(h, d_llr, batch, d_hard, d_iters, d_app, nl, llr_kind);
    h->last_layers = nl;
    unsigned wslot = 0;
    a.work = a.early_term ? next_work(h, &wslot) : nullptr; // only the parity-stop kernels refill
    begin_timing(h, stream);
    hipError_t e = nrldpc::launch_decode(s.g.bg, a, s.threads, s.lds_bytes, stream);
    end_timing(h, stream);
    if (a.work) (void)hipEventRecord(h->work_done[wslot], stream);
    if (e != hipSuccess) return hipfail(e, "decode kernel launch");
    return NRLDPC_OK;
}

int encode_launch(nrldpc_codec* h, const uint8_t* d_info, int batch, uint8_t* d_cw, hipStream_t stream) {
    const nrldpc::Schedule& s = h->sched;
    nrldpc::EncArgs a;
    a.info = d_info; a.cw = d_cw;
    a.row_ptr = h->d_row_ptr.p; a.col = h->d_col.p; a.shift = h->d_shift.p;
    a.batch = batch; a.Z = s.Z; a.nrows = s.g.nrows; a.ncols = s.g.ncols; a.kb = s.g.kb; a.nnz = s.g.nnz;
    a.p0_shift = h->p0_shift;
    for (int i = 0; i < 3; ++i) {
        a.step_row[i] = h->step_row[i]; a.step_col[i] = h->step_col[i]; a.step_shift[i] = h->step_shift[i];
        a.step_nk[i] = h->step_nk[i];
        for (int k = 0; k < 3; ++k) { a.step_kcol[i][k] = h->step_kcol[i][k]; a.step_kshift[i][k] = h->step_kshift[i][k]; }
    }
    begin_timing(h, stream);
    hipError_t e = nrldpc::launch_encode(a, stream);
    end_timing(h, stream);
    if (e != hipSuccess) return hipfail(e, "encode kernel launch");
    return NRLDPC_OK;
}

} // namespace

extern "C" {

int nrldpc_set_index(int32_t Z) { return nrldpc::set_index(Z); }
int nrldpc_lifting_size(int32_t K_b, int32_t K_prime) { return nrldpc::lifting_size(K_b, K_prime); }

const char* nrldpc_strerror(int code) {
    switch (code) {
        case NRLDPC_OK: return "ok";
        case NRLDPC_ERR_UNSUPPORTED: return "unsupported parameters";
        case NRLDPC_ERR_ARG: return "invalid argument";
        case NRLDPC_ERR_HIP: return "HIP runtime error";
        case NRLDPC_ERR_NOMEM: return "out of memory";
        default: return "unknown error";
    }
}
const char* nrldpc_last_error(void) { return g_err.c_str(); }
const char* nrldpc_version(void) { return "nrldpc-hip 0.3 (gfx950)"; }
int nrldpc_abi_version(void) { return NRLDPC_ABI_VERSION; }
#ifndef NRLDPC_BUILD_ID
#define NRLDPC_BUILD_ID "unknown"
#endif
const char* nrldpc_build_id(void) { return NRLDPC_BUILD_ID; }
#ifndef NRLDPC_KERNEL_ID
#define NRLDPC_KERNEL_ID "unknown"
#endif
const char* nrldpc_kernel_id(void) { return NRLDPC_KERNEL_ID; }

int nrldpc_default_rule(int32_t bg, int32_t n_layers, float* alpha, float* beta) {
    if (bg != 1 && bg != 2) return fail(NRLDPC_ERR_UNSUPPORTED, "BG must be 1 or 2");
    const int rows = bg == 1 ? NR_BG1_ROWS : NR_BG2_ROWS;
    if (n_layers == 0) n_layers = rows;
    if (n_layers < 4 || n_layers > rows) return fail(NRLDPC_ERR_UNSUPPORTED, "n_layers must be 0 or in 4..rows of the base graph");
    float a, b;
    nrldpc::default_rule(bg, n_layers, &a, &b);
    if (alpha) *alpha = a;
    if (beta) *beta = b;
    return NRLDPC_OK;
}

int nrldpc_create(const nrldpc_cfg* cfg_in, nrldpc_handle* out) {
    NRLDPC_API_BEGIN
    if (!cfg_in || !out) return fail(NRLDPC_ERR_ARG, "null cfg/out");
    *out = nullptr;
    // struct_size is the caller's sizeof(nrldpc_cfg): a caller built against another revision of the header is refused
    // instead of having fields read past the end of its struct
    if (cfg_in->struct_size != (uint32_t)sizeof(nrldpc_cfg))
        return fail(NRLDPC_ERR_ARG, "nrldpc_cfg.struct_size does not match this library (set it to sizeof(nrldpc_cfg); ABI revision mismatch?)");
    const nrldpc_cfg* cfg = cfg_in;
    if (cfg->bg != 1 && cfg->bg != 2) return fail(NRLDPC_ERR_UNSUPPORTED, "BG must be 1 or 2");
    if (nrldpc::set_index(cfg->Z) < 0) return fail(NRLDPC_ERR_UNSUPPORTED, "Invalid lifting size.");
    if (cfg->max_iter < 1 || cfg->max_iter > 2000) return fail(NRLDPC_ERR_UNSUPPORTED, "max_iter must be in 1..2000");
    if (cfg->llr_dtype < NRLDPC_LLR_F32 || cfg->llr_dtype > NRLDPC_LLR_F64)
        return fail(NRLDPC_ERR_UNSUPPORTED, "unknown llr_dtype");
    float alpha = cfg->alpha, beta = cfg->beta;
    const int rows = cfg->bg == 1 ? NR_BG1_ROWS : NR_BG2_ROWS;
    if (cfg->n_layers != NRLDPC_LAYERS_ALL && cfg->n_layers != NRLDPC_LAYERS_AUTO && (cfg->n_layers < 4 || cfg->n_layers > rows))
        return fail(NRLDPC_ERR_UNSUPPORTED, "n_layers must be 0 (all), NRLDPC_LAYERS_AUTO or in 4..rows of the base graph");
    const int rows0 = cfg->n_layers > 0 ? cfg->n_layers : rows; // under NRLDPC_LAYERS_AUTO: until the first call says otherwise
    if (alpha == 0.0f) // the caller leaves the check-node rule to the library: by rate (see nrldpc_default_rule)
        nrldpc::default_rule(cfg->bg, rows0, &alpha, &beta);
    if (!(alpha > 0.0f && alpha <= 1.0f)) return fail(NRLDPC_ERR_UNSUPPORTED, "alpha must be in (0,1]");
    if (!(beta >= 0.0f && beta <= 4.0f)) return fail(NRLDPC_ERR_UNSUPPORTED, "beta must be in [0,4] LLR units");
    if (cfg->early_term < 0 || cfg->early_term > 2) return fail(NRLDPC_ERR_UNSUPPORTED, "early_term must be 0, 1 or 2");
    if (cfg->early_term == 2) { // CRC-aided stop: generator with its x^L term, L = 6 ... 24, over the first crc_bits information bits
        const int kb = cfg->bg == 1 ? 22 : 10;
        if (cfg->crc_len < 6 || cfg->crc_len > 24 || (cfg->crc_poly >> cfg->crc_len) != 1u || !(cfg->crc_poly & 1u))
            return fail(NRLDPC_ERR_UNSUPPORTED, "early_term = 2 needs crc_poly = the generator with its x^crc_len and x^0 terms, crc_len in 6..24");
        if (cfg->crc_bits <= cfg->crc_len || cfg->crc_bits > kb * cfg->Z)
            return fail(NRLDPC_ERR_UNSUPPORTED, "early_term = 2 needs crc_len < crc_bits <= K (payload and CRC of one code block)");
    }
    int scale = cfg->llr_scale == 0 ? 8 : cfg->llr_scale;
    if (scale != 1 && scale != 2 && scale != 4 && scale != 8 && scale != 16 && scale != 32)
        return fail(NRLDPC_ERR_UNSUPPORTED, "llr_scale must be a power of two in 1..32");
    nrldpc_codec* h = new (std::nothrow) nrldpc_codec();
    if (!h) return fail(NRLDPC_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->alpha = alpha;
    // the offset lives on a grid of half fixed-point units (1/(2*llr_scale) LLR): the pipelined kernels round
    // alpha*m - beta inside one fused multiply-add against 2^23 - beta*scale, which has to be exact
    h->beta = std::nearbyint(2.0f * beta * (float)scale) / (2.0f * (float)scale);
    h->scale = scale;
    h->rule_auto = cfg->alpha == 0.0f;
    h->rule_layers = rows0;
    h->layers = cfg->n_layers == NRLDPC_LAYERS_AUTO ? NRLDPC_LAYERS_AUTO : rows0;
    if (!nrldpc::build_schedule(cfg->bg, cfg->Z, rows0, &h->sched)) {
        delete h;
        return fail(NRLDPC_ERR_UNSUPPORTED, "n_layers must be 0 (all), NRLDPC_LAYERS_AUTO or in 4..rows of the base graph");
    }
    if (!derive_encoder_order(h)) {
        delete h;
        return fail(NRLDPC_ERR_UNSUPPORTED, "base graph core is not dual-diagonal");
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        delete h;
        return fail(NRLDPC_ERR_HIP, "no HIP device available (this library has no CPU path)");
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) {
        delete h;
        return fail(NRLDPC_ERR_ARG, "device_id out of range");
    }
#define CREATE_TRY(expr)                                                  \
    do {                                                                  \
        hipError_t _e = (expr);                                           \
        if (_e != hipSuccess) { int rc = hipfail(_e, #expr); nrldpc_destroy(h); return rc; } \
    } while (0)
    DeviceScope scope(cfg->device_id); // the caller's current device is restored on return
    CREATE_TRY(scope.err);
    const nrldpc::Schedule& s = h->sched;
    CREATE_TRY(h->d_rot.reserve(s.rot.size()));
    CREATE_TRY(hipMemcpy(h->d_rot.p, s.rot.data(), s.rot.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> sh16(s.shift.begin(), s.shift.end());
    CREATE_TRY(h->d_row_ptr.reserve(s.g.nrows + 1));
    CREATE_TRY(h->d_col.reserve(s.g.nnz));
    CREATE_TRY(h->d_shift.reserve(s.g.nnz));
    CREATE_TRY(hipMemcpy(h->d_row_ptr.p, s.g.row_ptr, (s.g.nrows + 1) * 2, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_col.p, s.g.col, s.g.nnz, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_shift.p, sh16.data(), s.g.nnz * 2, hipMemcpyHostToDevice));
    if (cfg->early_term == 2) {
        // tab[i] = x^(n-1-i) mod g: bit i of the block (first bit = highest power, as comm.CRCDetector reads it,
        // NRLDPCDecoder.m:113-115) contributes this to the remainder; built from the last bit backwards by x -> x*x mod g
        const int n = cfg->crc_bits, L = cfg->crc_len;
        std::vector<uint32_t> tab((size_t)n);
        uint32_t r = 1u; // x^0
        const uint32_t top = 1u << L;
        for (int i = n - 1; i >= 0; --i) {
            tab[(size_t)i] = r;
            r <<= 1;
            if (r & top) r ^= cfg->crc_poly;
        }
        CREATE_TRY(h->d_crc.reserve((size_t)n));
        CREATE_TRY(hipMemcpy(h->d_crc.p, tab.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        h->sched.lds_bytes += 4 * (size_t)nrldpc::CRC_SLOTS * h->sched.ncw + 16; // the run-time-Z kernel's slots, behind its flags
    }
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
#undef CREATE_TRY
    *out = h;
    return NRLDPC_OK;
    NRLDPC_API_END
}

void nrldpc_destroy(nrldpc_handle h) {
    if (!h) return;
    DeviceScope scope(h->cfg.device_id);
    h->d_rot.release(); h->d_crc.release(); h->d_best.release(); h->pin_best.release(); h->d_work.release();
    h->d_row_ptr.release(); h->d_col.release(); h->d_shift.release();
    h->s_llr.release(); h->s_q.release(); h->s_hard.release(); h->s_bits.release(); h->s_pk.release(); h->s_iters.release(); h->s_app.release();
    for (auto& m : h->multi) { m.pin.release(); m.dev.release(); if (m.done) (void)hipEventDestroy(m.done); }
    for (int i = 0; i < nrldpc_codec::kSide; ++i) {
        if (h->side[i]) (void)hipStreamDestroy(h->side[i]);
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (auto ev : h->work_done)
        if (ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < nrldpc_codec::kSlots; ++i) {
        h->pin_in[i].release(); h->pin_out[i].release(); h->pin_it[i].release();
        if (h->xdone[i]) (void)hipEventDestroy(h->xdone[i]);
    }
    for (int i = 0; i < 2; ++i)
        if (h->xs[i]) (void)hipStreamDestroy(h->xs[i]);
    delete h->pool;
    delete h;
}

int nrldpc_get_dims(nrldpc_handle h, nrldpc_dims* out) {
    if (!h || !out) return fail(NRLDPC_ERR_ARG, "null handle/out");
    if (out->struct_size != (uint32_t)sizeof(nrldpc_dims))
        return fail(NRLDPC_ERR_ARG, "nrldpc_dims.struct_size does not match this library (set it to sizeof(nrldpc_dims) before the call)");
    const nrldpc::Schedule& s = h->sched;
    out->nrows = s.g.nrows; out->ncols = s.g.ncols; out->kb = s.g.kb; out->i_ls = s.ils;
    out->K = s.g.kb * s.Z; out->N_cw = s.g.ncols * s.Z; out->n_layers = h->layers;
    rule_for(h, h->layers != NRLDPC_LAYERS_AUTO ? h->layers : h->last_layers ? h->last_layers : s.g.nrows, &out->alpha, &out->beta);
    return NRLDPC_OK;
}

int nrldpc_set_layers(nrldpc_handle h, int32_t n_layers) {
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    const int rows = h->sched.g.nrows;
    if (n_layers == NRLDPC_LAYERS_AUTO) h->layers = NRLDPC_LAYERS_AUTO;
    else if (n_layers == NRLDPC_LAYERS_ALL) h->layers = rows;
    else if (n_layers >= 4 && n_layers <= rows) h->layers = n_layers;
    else return fail(NRLDPC_ERR_UNSUPPORTED, "n_layers must be 0 (all), NRLDPC_LAYERS_AUTO or in 4..rows of the base graph");
    return NRLDPC_OK;
}

int nrldpc_set_llr_dtype(nrldpc_handle h, int32_t llr_dtype) {
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    if (llr_dtype < NRLDPC_LLR_F32 || llr_dtype > NRLDPC_LLR_F64) return fail(NRLDPC_ERR_UNSUPPORTED, "unknown llr_dtype");
    h->cfg.llr_dtype = llr_dtype;
    return NRLDPC_OK;
}

int nrldpc_last_layers(nrldpc_handle h, int32_t* n_layers) {
    if (!h || !n_layers) return fail(NRLDPC_ERR_ARG, "null handle/out");
    *n_layers = h->last_layers;
    return NRLDPC_OK;
}

int nrldpc_count_layers(int32_t bg, int32_t Z, const void* llr, int32_t batch, int32_t llr_dtype) {
    nrldpc::BaseGraph g;
    if (!nrldpc::base_graph(bg, &g) || nrldpc::set_index(Z) < 0) { (void)fail(NRLDPC_ERR_UNSUPPORTED, "invalid BG / lifting size"); return -1; }
    if (batch < 0 || (batch > 0 && !llr) || llr_dtype < NRLDPC_LLR_F32 || llr_dtype > NRLDPC_LLR_F64) { (void)fail(NRLDPC_ERR_ARG, "invalid llr / batch / llr_dtype"); return -1; }
    const int kind = llr_dtype == NRLDPC_LLR_F64 ? NRLDPC_HQ_F64 : llr_dtype == NRLDPC_LLR_F16 ? NRLDPC_HQ_F16 : NRLDPC_HQ_F32;
    int best = g.kb + 3;
    nrldpc_top_block(llr, kind, (size_t)batch, 0, 1, Z, g.ncols, g.kb + 4, &best);
    return std::max(4, best - g.kb + 1);
}

int nrldpc_set_timing(nrldpc_handle h, int32_t enabled) {
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    h->timing = enabled != 0;
    h->have_time = false;
    return NRLDPC_OK;
}

int nrldpc_last_kernel_ms(nrldpc_handle h, float* ms) {
    if (!h || !ms) return fail(NRLDPC_ERR_ARG, "null handle/out");
    if (!h->timing || !h->have_time) return fail(NRLDPC_ERR_ARG, "no timed launch recorded");
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return NRLDPC_OK;
}

int nrldpc_last_host_phases(nrldpc_handle h, double* out10) {
    if (!h || !out10) return fail(NRLDPC_ERR_ARG, "null handle/out");
    if (h->host_phases[0] == 0) return fail(NRLDPC_ERR_ARG, "no chunked host-pointer call recorded on this handle");
    for (int i = 0; i < 10; ++i) out10[i] = h->host_phases[i];
    return NRLDPC_OK;
}

int nrldpc_decode_dev(nrldpc_handle h, const void* d_llr, int32_t batch, uint8_t* d_hard, int32_t* d_iters_out,
                      float* d_app_out, void* stream) {
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    if (batch < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (batch == 0) return NRLDPC_OK;
    if (!d_llr || !d_hard) return fail(NRLDPC_ERR_ARG, "null llr/hard pointer");
    if (h->cfg.llr_dtype == NRLDPC_LLR_F64) return fail(NRLDPC_ERR_ARG, "f64 LLRs are accepted by the host entry point only");
    if (d_app_out && h->cfg.early_term == 2) return fail(NRLDPC_ERR_UNSUPPORTED, "soft output is not available with the CRC-aided stop (early_term = 2)");
    DEVICE_SCOPE(h);
    int nl = 0;
    const int rc = resolve_layers_dev(h, d_llr, batch, static_cast<hipStream_t>(stream), &nl);
    if (rc) return rc;
    return decode_launch(h, d_llr, batch, d_hard, d_iters_out, d_app_out, static_cast<hipStream_t>(stream), nl);
}

int nrldpc_decode_multi_dev(int32_t n, const nrldpc_handle* hs, const void* const* d_llr, const int32_t* batch,
                            uint8_t* const* d_hard, int32_t* const* d_iters, void* stream) {
    NRLDPC_API_BEGIN
    if (n < 0) return fail(NRLDPC_ERR_ARG, "negative configuration count");
    if (n == 0) return NRLDPC_OK;
    if (!hs || !d_llr || !batch || !d_hard) return fail(NRLDPC_ERR_ARG, "null array");
    for (int i = 0; i < n; ++i) {
        if (!hs[i]) return fail(NRLDPC_ERR_ARG, "null handle");
        if (batch[i] < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
        if (batch[i] > 0 && (!d_llr[i] || !d_hard[i])) return fail(NRLDPC_ERR_ARG, "null llr/hard pointer");
        if (hs[i]->cfg.llr_dtype == NRLDPC_LLR_F64) return fail(NRLDPC_ERR_ARG, "f64 LLRs are accepted by the host entry point only");
        if (hs[i]->cfg.device_id != hs[0]->cfg.device_id) return fail(NRLDPC_ERR_ARG, "all handles of one call must live on one device");
    }
    nrldpc_codec* own = hs[0]; // its scratch holds the tables
    DEVICE_SCOPE(own);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // A configuration whose (BG, Z) has a compile-time-Z kernel (0.55-0.7x the time per codeword) and at least
    // NRLDPC_MULTI_Z64_MIN_ROWS check rows of work (default 512*384: it fills the chip; measured on BASELINE config 4,
    // whose buckets hold ~80 codewords: own launches from 16*384 / 64*384 / 256*384 rows on take 1.47 / 0.89 / 0.58 ms
    // against 0.57 ms with none) gets a launch of its own on that kernel; everything else shares ONE
    // launch of the run-time-Z kernel per (base graph, LLR type): argument blocks, then the workgroup prefix table.
    // All launches of the call are spread over the caller's stream and three side streams (fork / join with events):
    // a bucket's own launch fills only part of the chip, and the shared launches end in a tail of codewords that
    // never converge.
    static const long env_rows = getenv("NRLDPC_MULTI_Z64_MIN_ROWS") ? atol(getenv("NRLDPC_MULTI_Z64_MIN_ROWS")) : 512L * 384L;
    static const bool one_stream = getenv("NRLDPC_MULTI_ONE_STREAM") != nullptr; // A/B
    // The shared launches: one per (base graph, LLR type, workgroup class).  A launch has ONE workgroup size and ONE dynamic-LDS
    // size, and the run-time-Z body keeps every wave of the workgroup to the end: launched together, a 256-thread configuration
    // (27 KB of LDS, four workgroups per CU by its schedule) would run at the residency of a 512-thread one.  Classes: up to 256
    // threads, up to 512, above.  NRLDPC_MULTI_CLASSES=0: one launch per (base graph, LLR type) as before (A/B).
    static const bool env_classes = !(getenv("NRLDPC_MULTI_CLASSES") && atoi(getenv("NRLDPC_MULTI_CLASSES")) == 0);
    constexpr int kClasses = 6;
    static int edges[kClasses] = {256, 512, 768, 768, 768, 768}; // NRLDPC_MULTI_CLASS_EDGES=a,b,...: ascending thread counts (experiments)
    static const bool edges_read = [] {
        const char* e = getenv("NRLDPC_MULTI_CLASS_EDGES");
        for (int i = 0; e && *e && i < kClasses - 1; ++i) {
            edges[i] = atoi(e);
            for (int j = i + 1; j < kClasses; ++j) edges[j] = 768;
            e = strchr(e, ',');
            if (e) ++e;
        }
        return true;
    }();
    (void)edges_read;
    struct Group { std::vector<nrldpc::DecArgs> args; std::vector<int32_t> start; size_t lds = 0; int grid = 0, threads = 0; };
    Group g[2][2][kClasses];
    std::vector<int> routed;
    // layer count of every configuration: its handle's, or (NRLDPC_LAYERS_AUTO) read off its codewords -- all pre-pass kernels
    // are queued first and the stream is synchronised ONCE for the lot
    std::vector<int> nls((size_t)n, 0);
    {
        bool any_auto = false;
        for (int i = 0; i < n; ++i) {
            if (batch[i] == 0) continue;
            if (hs[i]->layers != NRLDPC_LAYERS_AUTO) { nls[i] = hs[i]->layers; continue; }
            for (int j = 0; j < i; ++j)
                if (hs[j] == hs[i] && batch[j] > 0) return fail(NRLDPC_ERR_ARG, "a handle under NRLDPC_LAYERS_AUTO may appear once per nrldpc_decode_multi_dev call");
            const int rc = auto_layers_enqueue(hs[i], d_llr[i], batch[i], st);
            if (rc) return rc;
            any_auto = true;
        }
        if (any_auto) {
            HIP_TRY(hipStreamSynchronize(st));
            for (int i = 0; i < n; ++i)
                if (batch[i] > 0 && hs[i]->layers == NRLDPC_LAYERS_AUTO) nls[i] = auto_layers_read(hs[i]);
        }
    }
    // (Ordering the shared launches' configurations largest lifting size first -- so that the longest workgroups do not start
    // last -- was measured on BASELINE configs[3]: 0.66 ms against 0.61-0.63 ms in the caller's order, twice; not kept.)
    for (int i = 0; i < n; ++i) {
        if (batch[i] == 0) continue;
        nrldpc_codec* h = hs[i];
        const nrldpc::Schedule& s = h->sched;
        // (a handle with the CRC-aided stop always gets a launch of its own: the shared kernel is built without it)
        if (h->cfg.early_term == 2 || ((long)batch[i] * s.Z >= env_rows && nrldpc::has_z64_kernel(s.g.bg, s.Z))) { routed.push_back(i); continue; }
        int cls = kClasses - 1;
        if (env_classes)
            for (cls = 0; cls < kClasses - 1 && s.threads > edges[cls]; ++cls) {}
        Group& q = g[s.g.bg - 1][h->cfg.llr_dtype == NRLDPC_LLR_F16 ? 1 : 0][cls];
        q.args.push_back(make_dec_args(h, d_llr[i], batch[i], d_hard[i], d_iters ? d_iters[i] : nullptr, nullptr, nls[i]));
        h->last_layers = nls[i];
        q.start.push_back(q.grid);
        q.grid += (batch[i] + s.ncw - 1) / s.ncw;
        q.lds = std::max(q.lds, s.lds_bytes);
        q.threads = std::max(q.threads, s.threads);
    }
    std::vector<char> host;
    size_t off[2][2][kClasses][2];
    int nlaunch = (int)routed.size();
    for (int b = 0; b < 2; ++b)
        for (int d = 0; d < 2; ++d)
            for (int c = 0; c < kClasses; ++c) {
                Group& q = g[b][d][c];
                if (q.args.empty()) continue;
                ++nlaunch;
                q.start.push_back(q.grid);
                host.resize((host.size() + 15) & ~(size_t)15);
                off[b][d][c][0] = host.size();
                host.insert(host.end(), reinterpret_cast<const char*>(q.args.data()),
                            reinterpret_cast<const char*>(q.args.data() + q.args.size()));
                off[b][d][c][1] = host.size();
                host.insert(host.end(), reinterpret_cast<const char*>(q.start.data()),
                            reinterpret_cast<const char*>(q.start.data() + q.start.size()));
            }
    if (nlaunch == 0) return NRLDPC_OK;
    // table slot: reused only after the launches that read it have completed (calls on different streams may overlap)
    nrldpc_codec::MultiSlot& m = own->multi[own->multi_next];
    own->multi_next = (own->multi_next + 1) % nrldpc_codec::kMultiSlots;
    if (!m.done) HIP_TRY(hipEventCreateWithFlags(&m.done, hipEventDisableTiming));
    if (m.used) HIP_TRY(hipEventSynchronize(m.done));
    if (!host.empty()) {
        HIP_TRY(m.pin.reserve(host.size()));
        HIP_TRY(m.dev.reserve(host.size()));
        memcpy(m.pin.p, host.data(), host.size());
        HIP_TRY(hipMemcpyAsync(m.dev.p, m.pin.p, host.size(), hipMemcpyHostToDevice, st));
    }
    const bool fan = nlaunch > 1 && !one_stream;
    static const int env_side = getenv("NRLDPC_MULTI_STREAMS") ? atoi(getenv("NRLDPC_MULTI_STREAMS")) : 5;
    const int nside = std::max(1, std::min({env_side, (int)nrldpc_codec::kSide, nlaunch - 1}));
    if (fan) { // fork: the side streams start after the table copy (and everything the caller queued before it)
        if (!own->ev_fork) HIP_TRY(hipEventCreateWithFlags(&own->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(own->ev_fork, st));
        for (int i = 0; i < nside; ++i) {
            if (!own->side[i]) HIP_TRY(hipStreamCreateWithFlags(&own->side[i], hipStreamNonBlocking));
            if (!own->ev_join[i]) HIP_TRY(hipEventCreateWithFlags(&own->ev_join[i], hipEventDisableTiming));
            HIP_TRY(hipStreamWaitEvent(own->side[i], own->ev_fork, 0));
        }
    }
    int rc = NRLDPC_OK, k = 0;
    auto next_stream = [&]() -> hipStream_t { const int q = k++ % (nside + 1); return (!fan || q == 0) ? st : own->side[q - 1]; };
    // shared launches first (the longest: their tail of never-converging small codewords), then the buckets' own
    // (the largest class first: its workgroups are the longest)
    for (int c = kClasses - 1; c >= 0 && rc == NRLDPC_OK; --c)
        for (int b = 0; b < 2 && rc == NRLDPC_OK; ++b)
            for (int d = 0; d < 2 && rc == NRLDPC_OK; ++d) {
                const Group& q = g[b][d][c];
                if (q.args.empty()) continue;
                hipError_t e = nrldpc::launch_decode_multi_wg(
                    b + 1, d ? NRLDPC_K_F16 : NRLDPC_K_F32, reinterpret_cast<const nrldpc::DecArgs*>(m.dev.p + off[b][d][c][0]),
                    reinterpret_cast<const int32_t*>(m.dev.p + off[b][d][c][1]), (int)q.args.size(), q.grid, q.threads, q.lds, next_stream());
                if (e != hipSuccess) rc = hipfail(e, "multi-configuration decode launch");
            }
    for (size_t r = 0; r < routed.size() && rc == NRLDPC_OK; ++r) {
        const int i = routed[r];
        nrldpc_codec* h = hs[i];
        nrldpc::DecArgs a = make_dec_args(h, d_llr[i], batch[i], d_hard[i], d_iters ? d_iters[i] : nullptr, nullptr, nls[i]);
        h->last_layers = nls[i];
        unsigned wslot = 0;
        a.work = a.early_term ? next_work(h, &wslot) : nullptr;
        hipStream_t ls = next_stream();
        hipError_t e = nrldpc::launch_decode(h->sched.g.bg, a, h->sched.threads, h->sched.lds_bytes, ls);
        if (a.work) (void)hipEventRecord(h->work_done[wslot], ls);
        if (e != hipSuccess) rc = hipfail(e, "decode kernel launch");
    }
    if (fan)
        for (int i = 0; i < nside; ++i) { // join: the caller's stream continues after every side launch
            (void)hipEventRecord(own->ev_join[i], own->side[i]);
            (void)hipStreamWaitEvent(st, own->ev_join[i], 0);
        }
    (void)hipEventRecord(m.done, st); // also on a failed launch: the copy above is in flight
    m.used = true;
    return rc;
    NRLDPC_API_END
}

int nrldpc_quantise_llr(int8_t* dst, const void* src, int64_t n, int32_t llr_dtype, int32_t llr_scale) {
    if (n <= 0 || !dst || !src) return 0;
    const int kind = llr_dtype == NRLDPC_LLR_F64 ? NRLDPC_HQ_F64 : llr_dtype == NRLDPC_LLR_F16 ? NRLDPC_HQ_F16 : NRLDPC_HQ_F32;
    return nrldpc_quantise_i8(dst, src, (size_t)n, kind, (float)llr_scale) ? 1 : 0;
}

} // extern "C"

namespace {
// nrldpc_decode / nrldpc_decode_packed: `packed` = the hard decisions leave as one BIT per bit ([batch][ceil(K/8)] bytes, least
// significant bit first) instead of one byte per bit -- packed on the device, so that an eighth of the bytes crosses PCIe and
// goes through the copy into the caller's array
// nl_call > 0: the layer count of this call whatever the handle says (nrldpc_pool_*: found once for the whole batch)
// full_scan: under NRLDPC_LAYERS_AUTO scan the whole batch before the first chunk (the restart of a call whose chunk-by-chunk scan
// met a codeword that reaches higher than the first chunk's did)
int decode_host(nrldpc_handle h, const void* llr, int32_t batch, uint8_t* hard, int32_t* iters_out, float* app_out, bool packed,
                int nl_call = 0, bool full_scan = false) {
    NRLDPC_API_BEGIN
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    if (batch < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (batch == 0) return NRLDPC_OK;
    if (!llr || !hard) return fail(NRLDPC_ERR_ARG, "null llr/hard pointer");
    if (app_out && h->cfg.early_term == 2) return fail(NRLDPC_ERR_UNSUPPORTED, "soft output is not available with the CRC-aided stop (early_term = 2)");
    DEVICE_SCOPE(h);
    const nrldpc::Schedule& s = h->sched;
    const size_t ncw = (size_t)s.g.ncols * s.Z, K = (size_t)s.g.kb * s.Z;
    const size_t KO = packed ? (K + 7) / 8 : K; // bytes of one codeword's hard decisions in the caller's array
    const size_t cap = (h->cfg.max_batch > batch) ? (size_t)h->cfg.max_batch : (size_t)batch;
    const bool f64 = h->cfg.llr_dtype == NRLDPC_LLR_F64;
    const size_t eb = llr_elem_bytes(h->cfg.llr_dtype); // on the device (MATLAB doubles are narrowed to f32 on the host)
    HIP_TRY(h->s_llr.reserve(cap * ncw * eb));
    HIP_TRY(h->s_hard.reserve(cap * K));
    if (packed) HIP_TRY(h->s_pk.reserve(cap * KO));
    if (iters_out) HIP_TRY(h->s_iters.reserve(cap));
    if (app_out) HIP_TRY(h->s_app.reserve(cap * ncw));

    // Batches above 8 MB: chunks (NRLDPC_HOST_CHUNK_MB of wire bytes, default 32 / 16 as int8; NRLDPC_HOST_THREADS copy threads,
    // default 16, pinned to the caller's NUMA node unless NRLDPC_HOST_NO_PIN is set; NRLDPC_HOST_PIPELINE=0 disables)
    // flow caller array -> pinned slot (copy threads; quantised to int8 on the way when the kernel reads that) -> H2D ->
    // decode -> D2H -> pinned slot -> caller array through four slots on two alternating streams, so that host copies,
    // both DMA directions and the kernels of neighbouring chunks overlap and the host runs up to three chunks ahead of
    // the device.  Same results as one launch (NRLDPC_HOST_TRACE=1: phase times of each call on stderr).
    const size_t in_bytes = (size_t)batch * ncw * eb;
    static const int env_chunk_mb = getenv("NRLDPC_HOST_CHUNK_MB") ? atoi(getenv("NRLDPC_HOST_CHUNK_MB")) : 32;
    static const int env_threads = getenv("NRLDPC_HOST_THREADS") ? atoi(getenv("NRLDPC_HOST_THREADS")) : 16;
    // The copy threads quantise while they copy (1 byte per LLR on the wire instead of 2 / 4; nrldpc_host_quant.h) and a
    // small kernel expands each chunk to fp16 on the device (nrldpc_expand.hip).  NRLDPC_HOST_I8=0: A/B against the native format.
    const bool i8 = !(getenv("NRLDPC_HOST_I8") && atoi(getenv("NRLDPC_HOST_I8")) == 0);
    const int hq_kind = f64 ? NRLDPC_HQ_F64 : h->cfg.llr_dtype == NRLDPC_LLR_F16 ? NRLDPC_HQ_F16 : NRLDPC_HQ_F32;
    const size_t host_eb = f64 ? 8 : eb; // element size of the caller's array
    static const int env_pipe = getenv("NRLDPC_HOST_PIPELINE") ? atoi(getenv("NRLDPC_HOST_PIPELINE")) : 1;
    int nl = nl_call > 0 ? nl_call : h->layers; // NRLDPC_LAYERS_AUTO is resolved below, from the caller's array
    if (env_pipe && in_bytes >= ((size_t)8 << 20) && !app_out && !h->timing) {
        constexpr int NS = nrldpc_codec::kSlots;
        // chunk: ~NRLDPC_HOST_CHUNK_MB of wire bytes (int8 when the copy threads quantise), at least four chunks per call,
        // and -- where that leaves more than one -- whole rounds of 512 codewords (two per CU of a 256-CU device: a
        // 642-codeword chunk ran as one full round and a quarter-full one)
        const size_t wire_eb = i8 ? 1 : eb;
        const size_t chunk_bytes = std::min<size_t>(((size_t)std::max(1, env_chunk_mb) << 20) / (i8 ? 2 : 1), (size_t)batch * ncw * wire_eb / 4);
        int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)batch, chunk_bytes / (ncw * wire_eb)));
        if (chunk >= 512) chunk -= chunk % 512;
        if (!h->pool) {
            // the caller's own thread works too (enqueues, waits for events): leave it a CPU of the quota
            const int want = h->host_threads_hint > 0 ? h->host_threads_hint : std::max(1, env_threads);
            h->pool = new (std::nothrow) HostPool(std::max(1, std::min(want, usable_cpus() - 1)));
            if (!h->pool) return fail(NRLDPC_ERR_NOMEM, "host thread pool");
        }
        h->pool->follow(llr, (size_t)batch * ncw * host_eb);
        struct Hot { // the copy threads poll between the jobs of this call, and sleep again when it returns (whichever way)
            HostPool* p;
            explicit Hot(HostPool* q) : p(q) { p->set_hot(true); }
            ~Hot() { p->set_hot(false); }
        } hot_guard(h->pool);
        // NRLDPC_LAYERS_AUTO, on the copy threads (the all-zero column blocks are read here, once).  Chunk by chunk: the count is
        // taken from the FIRST chunk, and each later chunk's tail above it is checked right before that chunk is quantised -- the
        // device already works on the earlier chunks meanwhile, where a scan of the whole batch up front was a serial phase as long
        // as the quantisation itself (MATLAB doubles at R = 8/9: 5.4 ms a call against 4.0 ms with every row).  A call is rate-
        // matched alike throughout in practice; if a later chunk does reach higher, the call starts again with a full scan.
        const int scan_first = std::min(batch, chunk);
        const bool lazy_scan = nl == NRLDPC_LAYERS_AUTO && !full_scan && scan_first < batch;
        double t_scan = 0;
        auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        if (nl == NRLDPC_LAYERS_AUTO) {
            const double t0 = now_ms();
            nl = layers_of_block(s, h->pool->top_block(llr, hq_kind, (size_t)(lazy_scan ? scan_first : batch), s.Z, s.g.ncols, s.g.kb + 4));
            t_scan += now_ms() - t0;
        }
        // what no active layer reads (extension columns kb + nl ...) is neither quantised nor sent: a row goes out as its first
        // `act` LLRs (int8 wire format only; the kernels never touch the rest of the row -- stale staging bytes at worst)
        const size_t act = i8 ? std::min(ncw, (size_t)(s.g.kb + nl) * (size_t)s.Z) : ncw;
        for (int i = 0; i < NS; ++i) {
            HIP_TRY(h->pin_in[i].reserve((size_t)chunk * ncw * eb));
            HIP_TRY(h->pin_out[i].reserve((size_t)chunk * K));
            if (iters_out) HIP_TRY(h->pin_it[i].reserve((size_t)chunk * 4));
            if (!h->xdone[i]) HIP_TRY(hipEventCreateWithFlags(&h->xdone[i], hipEventDisableTiming));
        }
        const size_t q_slot = ((size_t)chunk * ncw + 255) & ~(size_t)255; // bytes of one int8 chunk on the device
        if (i8) HIP_TRY(h->s_q.reserve(q_slot * NS));
        for (int i = 0; i < 2; ++i)
            if (!h->xs[i]) HIP_TRY(hipStreamCreateWithFlags(&h->xs[i], hipStreamNonBlocking));
        // Chunk k = codewords [starts[k], starts[k+1]).  The first chunk's copy + H2D is the one stretch of a call in which the
        // device has nothing to do, so a call of several rounds opens with two half chunks (one workgroup per CU each).
        // (NRLDPC_HOST_RAMP=1; off by default: measured +-4 % either way on the fp16 path, within the run-to-run spread)
        static const bool env_ramp = getenv("NRLDPC_HOST_RAMP") && atoi(getenv("NRLDPC_HOST_RAMP")) != 0;
        std::vector<int> starts(1, 0);
        for (int pos = 0, k = 0; pos < batch; ++k) {
            const int want = (env_ramp && k < 2 && chunk >= 512 && batch >= 3 * chunk) ? chunk / 2 : chunk;
            pos += std::min(want, batch - pos);
            starts.push_back(pos);
        }
        const int nchunks = (int)starts.size() - 1;
        // an early error return must not leave copies or kernels of this call in flight on the two streams
        struct Quiesce {
            hipStream_t* xs; bool armed = true;
            ~Quiesce() { if (armed) for (int i = 0; i < 2; ++i) if (xs[i]) (void)hipStreamSynchronize(xs[i]); }
        } quiesce{h->xs};
        static const bool trace = getenv("NRLDPC_HOST_TRACE") != nullptr; // phase times of one call on stderr
        double t_quant = 0, t_wait = 0, t_out = 0, t_enq = 0;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        int drained = 0; // chunks 0 .. drained-1 are back in the caller's arrays
        int cur = -1;    // the chunk whose results are on their way out, and how many of its bytes are (it may go out in pieces:
        size_t cur_done = 0; // by the copy threads while they are idle, by this thread while they quantise the next chunk)
        auto drain_step = [&](size_t budget, bool use_pool) -> int { // up to `budget` bytes of chunk `drained`: pinned slot -> caller array
            const int k = drained, sl = k % NS, c0 = starts[k], n = starts[k + 1] - c0;
            const size_t total = (size_t)n * KO;
            if (cur != k) {
                const double t0 = now();
                HIP_TRY(hipEventSynchronize(h->xdone[sl]));
                t_wait += now() - t0;
                cur = k; cur_done = 0;
            }
            const double t1 = now();
            const size_t len = std::min(budget, total - cur_done);
            uint8_t* dst = hard + (size_t)c0 * KO + cur_done;
            const char* src = h->pin_out[sl].p + cur_done;
            if (use_pool && len >= ((size_t)1 << 20)) h->pool->move(dst, src, len, false); // (bit-packed: an eighth of the bytes, not worth a fan-out)
            else memcpy(dst, src, len);
            cur_done += len;
            if (cur_done == total) {
                if (iters_out) memcpy(iters_out + c0, h->pin_it[sl].p, (size_t)n * 4);
                ++drained;
            }
            t_out += now() - t1;
            return NRLDPC_OK;
        };
        auto ready = [&]() { return drained == cur || hipEventQuery(h->xdone[drained % NS]) == hipSuccess; };
        for (int k = 0; k < nchunks; ++k) {
            const int sl = k % NS, st = k & 1, c0 = starts[k], n = starts[k + 1] - c0;
            // the slot's previous chunk has to be out (that also frees pin_in[sl]); chunks that happen to be finished
            // are taken out now (the copy threads are idle) rather than in one lump at the end
            while (drained <= k - NS || (drained < k && ready())) {
                int rc = drain_step((size_t)-1, true); if (rc) return rc;
            }
            const size_t off = (size_t)c0 * ncw;
            if (lazy_scan && c0 + n > scan_first && nl < s.g.nrows) { // (the part of) this chunk the first scan did not see: anything above the count?
                const double t0 = now_ms();
                const int c1 = std::max(c0, scan_first);
                const int top = h->pool->top_block(static_cast<const char*>(llr) + (size_t)c1 * ncw * host_eb, hq_kind, (size_t)(c0 + n - c1), s.Z,
                                                   s.g.ncols, s.g.kb + nl);
                t_scan += now_ms() - t0;
                if (top >= s.g.kb + nl) { // yes: every chunk already sent was decoded with too few rows -- start again, whole-batch scan first
                    for (int i = 0; i < 2; ++i) HIP_TRY(hipStreamSynchronize(h->xs[i]));
                    quiesce.armed = false;
                    return decode_host(h, llr, batch, hard, iters_out, app_out, packed, nl_call, true);
                }
            }
            char* d_in = h->s_llr.p + off * eb;
            int kind = -1; // the handle's own format
            const double tq0 = now();
            // The first chunk is the one stretch of a call in which the device has nothing to do: it is quantised and sent in four
            // pieces, so that its H2D copy overlaps its own quantisation (later chunks overlap the previous chunk's kernel anyway).
            bool as_i8 = i8;
            if (i8) {
                int8_t* d_q = h->s_q.p + q_slot * sl;
                int8_t* pin = reinterpret_cast<int8_t*>(h->pin_in[sl].p);
                const size_t total = (size_t)n * act; // compact: [n][act]
                const int parts = (k == 0 && total >= ((size_t)1 << 20)) ? 4 : 1;
                bool sent = false;
                for (int pi = 0; pi < parts && as_i8; ++pi) {
                    const size_t lo = (total * pi / parts) & ~(size_t)63, hi = pi + 1 == parts ? total : ((total * (pi + 1) / parts) & ~(size_t)63);
                    h->pool->quantise_rows_start(pin, static_cast<const char*>(llr) + off * host_eb, act, ncw, lo, hi, hq_kind, (float)h->scale);
                    // while the copy threads quantise, this thread takes finished chunks' results out, a piece at a time (round 4
                    // ran the two one after the other; with MATLAB doubles the copy threads are the longest phase of a call)
                    while (!h->pool->done() && drained < k && ready()) {
                        int rc = drain_step((size_t)256 << 10, false);
                        if (rc) { (void)h->pool->quantise_rows_finish(); return rc; }
                    }
                    if (h->pool->quantise_rows_finish()) {
                        as_i8 = false; // a -inf: int8 has no code for it (pieces already sent are simply not used)
                    } else {
                        HIP_TRY(hipMemcpyAsync(d_q + lo, pin + lo, hi - lo, hipMemcpyHostToDevice, h->xs[st]));
                        sent = true;
                    }
                }
                if (as_i8) {
                    kind = NRLDPC_K_F16; // whatever the handle's format: this chunk reaches the decoder as fp16
                    HIP_TRY(nrldpc::launch_expand_i8_rows(d_q, d_in, (size_t)n, act, ncw, 1.0f / (float)h->scale, h->xs[st]));
                } else if (sent) {
                    // the pieces already queued still read this pinned slot, which the fallback below overwrites
                    HIP_TRY(hipStreamSynchronize(h->xs[st]));
                }
            }
            if (!as_i8) { // the handle's own format (NRLDPC_HOST_I8=0, or a chunk that holds a -inf)
                if (f64) h->pool->move(h->pin_in[sl].p, static_cast<const double*>(llr) + off, (size_t)n * ncw, true);
                else h->pool->move(h->pin_in[sl].p, static_cast<const char*>(llr) + off * eb, (size_t)n * ncw * eb, false);
                HIP_TRY(hipMemcpyAsync(d_in, h->pin_in[sl].p, (size_t)n * ncw * eb, hipMemcpyHostToDevice, h->xs[st]));
            }
            const double tq1 = now();
            int rc = decode_launch(h, d_in, n, h->s_hard.p + (size_t)c0 * K, iters_out ? h->s_iters.p + c0 : nullptr, nullptr, h->xs[st], nl, kind);
            if (rc) return rc;
            t_quant += tq1 - tq0; t_enq -= tq1;
            if (packed) {
                HIP_TRY(nrldpc::launch_pack_bits(h->s_hard.p + (size_t)c0 * K, h->s_pk.p + (size_t)c0 * KO, n, (int)K, h->xs[st]));
                HIP_TRY(hipMemcpyAsync(h->pin_out[sl].p, h->s_pk.p + (size_t)c0 * KO, (size_t)n * KO, hipMemcpyDeviceToHost, h->xs[st]));
            } else {
                HIP_TRY(hipMemcpyAsync(h->pin_out[sl].p, h->s_hard.p + (size_t)c0 * K, (size_t)n * K, hipMemcpyDeviceToHost, h->xs[st]));
            }
            if (iters_out) HIP_TRY(hipMemcpyAsync(h->pin_it[sl].p, h->s_iters.p + c0, (size_t)n * 4, hipMemcpyDeviceToHost, h->xs[st]));
            HIP_TRY(hipEventRecord(h->xdone[sl], h->xs[st]));
            t_enq += now();
        }
        while (drained < nchunks) { int rc = drain_step((size_t)-1, true); if (rc) return rc; }
        {
            double* ph = h->host_phases;
            ph[0] = nchunks; ph[1] = chunk; ph[2] = nl; ph[3] = t_scan; ph[4] = t_quant; ph[5] = t_enq; ph[6] = t_wait; ph[7] = t_out;
            ph[8] = h->pool->node_; ph[9] = sched_getcpu();
        }
        if (trace)
            fprintf(stderr, "[nrldpc host path] %d chunks of %d, %d layers (%zu of %zu LLRs per codeword on the wire; scan %.2f ms): copy/quantise in %.2f ms (incl. H2D enqueue), launch+D2H enqueue %.2f, wait for device %.2f, copy out %.2f; copy threads on NUMA node %d (caller on CPU %d)\n",
                    nchunks, chunk, nl, act, ncw, t_scan, t_quant, t_enq, t_wait, t_out, h->pool->node_, sched_getcpu());
        quiesce.armed = false; // every chunk was drained behind its event
        return NRLDPC_OK;
    }

    if (nl == NRLDPC_LAYERS_AUTO) { // a small batch: the caller's thread scans it
        int best = s.g.kb + 3;
        nrldpc_top_block(llr, hq_kind, (size_t)batch, 0, 1, s.Z, s.g.ncols, s.g.kb + 4, &best);
        nl = layers_of_block(s, best);
    }
    // The reference's own call pattern -- one step() = the C code blocks of one transport block (NRLDPCDecoder.m:257-266), i.e. a
    // call of one to a few codewords -- is bound by latency, not by bytes: three hipMemcpyAsync calls on pageable memory (each
    // staged and synchronised inside the runtime) and the kernel launches were 0.09 ms of a 0.12 ms call.  Small calls therefore go
    // ZERO-COPY: the caller's LLRs are narrowed / copied into a pinned buffer that the decoder kernel reads over PCIe in its
    // prologue (hipHostMalloc'd memory is device-accessible at the same address), hard decisions (bit-packed by the pack kernel
    // when asked) and iteration counts are written by the kernels straight into pinned memory, and the call is: one CPU copy in,
    // one or two launches, one stream synchronisation, one CPU copy out.  NRLDPC_HOST_ZEROCOPY_KB: largest input (device format)
    // that goes this way, default 2048; 0 = off (A/B).
    static const size_t zc_max = (size_t)(getenv("NRLDPC_HOST_ZEROCOPY_KB") ? atol(getenv("NRLDPC_HOST_ZEROCOPY_KB")) : 2048) << 10;
    if (!app_out && !h->timing && in_bytes <= zc_max) {
        HIP_TRY(h->pin_in[0].reserve(in_bytes));
        HIP_TRY(h->pin_out[0].reserve((size_t)batch * K)); // one byte per bit from the kernel; bit-packed output is packed here, on the CPU
        if (iters_out) HIP_TRY(h->pin_it[0].reserve((size_t)batch * 4));
        if (!h->xs[0]) HIP_TRY(hipStreamCreateWithFlags(&h->xs[0], hipStreamNonBlocking));
        if (f64) {
            const double* d = static_cast<const double*>(llr);
            float* o = reinterpret_cast<float*>(h->pin_in[0].p);
            for (size_t i = 0; i < (size_t)batch * ncw; ++i) o[i] = (float)d[i];
        } else {
            memcpy(h->pin_in[0].p, llr, in_bytes);
        }
        uint8_t* d_out = reinterpret_cast<uint8_t*>(h->pin_out[0].p);
        int rc = decode_launch(h, h->pin_in[0].p, batch, d_out, iters_out ? reinterpret_cast<int32_t*>(h->pin_it[0].p) : nullptr, nullptr, h->xs[0], nl);
        if (rc) { (void)hipStreamSynchronize(h->xs[0]); return rc; }
        HIP_TRY(hipStreamSynchronize(h->xs[0]));
        if (packed) { // a few kilobytes: packing them here saves the pack kernel's launch (one launch per call instead of two)
            for (size_t b = 0; b < (size_t)batch; ++b) {
                const uint8_t* src = d_out + b * K;
                uint8_t* dst = hard + b * KO;
                size_t k = 0;
                for (; k + 8 <= K; k += 8) {
                    uint64_t x;
                    memcpy(&x, src + k, 8);
                    dst[k >> 3] = (uint8_t)(((x & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
                }
                if (k < K) {
                    unsigned v = 0;
                    for (size_t j = k; j < K; ++j) v |= (unsigned)(src[j] & 1u) << (j - k);
                    dst[k >> 3] = (uint8_t)v;
                }
            }
        } else {
            memcpy(hard, d_out, (size_t)batch * K);
        }
        if (iters_out) memcpy(iters_out, h->pin_it[0].p, (size_t)batch * 4);
        return NRLDPC_OK;
    }
    const void* src = llr;
    if (f64) { // MATLAB doubles: narrow on the host (halves PCIe bytes)
        h->h_narrow.resize((size_t)batch * ncw);
        const double* d = static_cast<const double*>(llr);
        for (size_t i = 0; i < (size_t)batch * ncw; ++i) h->h_narrow[i] = (float)d[i];
        src = h->h_narrow.data();
    }
    HIP_TRY(hipMemcpyAsync(h->s_llr.p, src, (size_t)batch * ncw * eb, hipMemcpyHostToDevice, nullptr));
    // F64 was narrowed to f32 above; the kernel sees f32 in that case.
    int rc = decode_launch(h, h->s_llr.p, batch, h->s_hard.p, iters_out ? h->s_iters.p : nullptr,
                           app_out ? h->s_app.p : nullptr, nullptr, nl);
    if (rc) return rc;
    if (packed) {
        HIP_TRY(nrldpc::launch_pack_bits(h->s_hard.p, h->s_pk.p, batch, (int)K, nullptr));
        HIP_TRY(hipMemcpyAsync(hard, h->s_pk.p, (size_t)batch * KO, hipMemcpyDeviceToHost, nullptr));
    } else {
        HIP_TRY(hipMemcpyAsync(hard, h->s_hard.p, (size_t)batch * K, hipMemcpyDeviceToHost, nullptr));
    }
    if (iters_out) HIP_TRY(hipMemcpyAsync(iters_out, h->s_iters.p, (size_t)batch * 4, hipMemcpyDeviceToHost, nullptr));
    if (app_out) HIP_TRY(hipMemcpyAsync(app_out, h->s_app.p, (size_t)batch * ncw * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return NRLDPC_OK;
    NRLDPC_API_END
}
} // namespace

extern "C" {

int nrldpc_decode(nrldpc_handle h, const void* llr, int32_t batch, uint8_t* hard, int32_t* iters_out, float* app_out) {
    return decode_host(h, llr, batch, hard, iters_out, app_out, false);
}

int nrldpc_decode_packed(nrldpc_handle h, const void* llr, int32_t batch, uint8_t* hard_packed, int32_t* iters_out) {
    return decode_host(h, llr, batch, hard_packed, iters_out, nullptr, true);
}

int nrldpc_decode_packed_layers(nrldpc_handle h, const void* llr, int32_t batch, uint8_t* hard_packed, int32_t* iters_out,
                                int32_t n_layers) {
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    const int saved = h->layers; // the handle is driven by one caller thread: the count of this call, then the handle's own again
    int rc = nrldpc_set_layers(h, n_layers);
    if (rc != NRLDPC_OK) return rc;
    rc = decode_host(h, llr, batch, hard_packed, iters_out, nullptr, true);
    h->layers = saved;
    return rc;
}

} // extern "C"

// ---- multi-GPU pool: N handles, N host threads, a queue of chunks ------------------------------------------------
struct nrldpc_pool {
    std::vector<nrldpc_handle> hs;
    std::vector<std::thread> th;
    std::vector<int32_t> split;
    int chunks_per_device = 2;
    size_t ncw = 0, K = 0, eb = 0;
    // one job at a time (the pool, like a handle, is driven by one caller thread)
    std::mutex m;
    std::condition_variable cv, cv_done;
    unsigned gen = 0;
    bool stop = false;
    const char* llr = nullptr; uint8_t* hard = nullptr; int32_t* iters = nullptr;
    int batch = 0, chunk = 0, nchunks = 0, next = 0, running = 0, rc = NRLDPC_OK;
    bool packed = false; // hard decisions leave bit-packed (nrldpc_pool_decode_packed)
    int layers = 0;      // as nrldpc_codec::layers (nrldpc_pool_set_layers); nl_call: what this call runs with
    int nl_call = 0;
    // NRLDPC_LAYERS_AUTO, device form: every shard scans its slice, the last one to arrive takes the maximum, all launch with it
    int scan_arrived = 0, scan_best = 0;
    std::condition_variable cv_scan;
    std::string err;
    // device-resident job (nrldpc_pool_decode_dev): shard i launches on its own stream and waits for it
    bool dev_job = false;
    const void* const* dv_llr = nullptr; const int32_t* dv_batch = nullptr; uint8_t* const* dv_hard = nullptr;
    int32_t* const* dv_iters = nullptr;
    std::vector<hipStream_t> streams;

    int decode_dev_shard(int i) {
        if (dv_batch[i] <= 0) return dv_batch[i] < 0 ? fail(NRLDPC_ERR_ARG, "negative batch") : NRLDPC_OK;
        nrldpc_handle h = hs[i];
        DEVICE_SCOPE(h);
        if (!streams[i]) HIP_TRY(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
        // The shard's stream is private and non-blocking, so it is ordered against nothing the caller queued: wait for
        // everything already submitted to this device (whatever stream produced d_llr[i]) before the launch.  The call is
        // synchronous anyway -- it returns when every shard is done -- so this costs no overlap the caller could have had.
        HIP_TRY(hipDeviceSynchronize());
        if (h->cfg.llr_dtype == NRLDPC_LLR_F64) return fail(NRLDPC_ERR_ARG, "f64 LLRs are accepted by the host entry point only");
        const int r = decode_launch(h, dv_llr[i], dv_batch[i], dv_hard[i], dv_iters ? dv_iters[i] : nullptr, nullptr, streams[i], nl_call);
        if (r != NRLDPC_OK) return r;
        HIP_TRY(hipStreamSynchronize(streams[i]));
        return NRLDPC_OK;
    }
    // first phase of a device-form call under NRLDPC_LAYERS_AUTO: this shard's highest non-zero column block (first - 1: none)
    int scan_dev_shard(int i, int* top) {
        nrldpc_handle h = hs[i];
        *top = h->sched.g.kb + 3;
        if (dv_batch[i] <= 0 || h->cfg.llr_dtype == NRLDPC_LLR_F64) return NRLDPC_OK; // (errors are reported by the decode phase)
        DEVICE_SCOPE(h);
        if (!streams[i]) HIP_TRY(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
        HIP_TRY(hipDeviceSynchronize());
        const int r = auto_layers_enqueue(h, dv_llr[i], dv_batch[i], streams[i]);
        if (r != NRLDPC_OK) return r;
        HIP_TRY(hipStreamSynchronize(streams[i]));
        *top = *reinterpret_cast<const int32_t*>(h->pin_best.p);
        return NRLDPC_OK;
    }

    void worker(int i) {
        unsigned seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
            }
            if (dev_job) {
                if (layers == NRLDPC_LAYERS_AUTO) {
                    int top = 0;
                    const int rs = scan_dev_shard(i, &top);
                    std::unique_lock<std::mutex> lk(m);
                    if (rs != NRLDPC_OK && rc == NRLDPC_OK) { rc = rs; err = nrldpc_last_error(); }
                    scan_best = std::max(scan_best, top);
                    if (++scan_arrived == (int)hs.size()) {
                        nl_call = layers_of_block(hs[0]->sched, scan_best);
                        cv_scan.notify_all();
                    } else {
                        cv_scan.wait(lk, [&] { return scan_arrived == (int)hs.size(); });
                    }
                    if (rc != NRLDPC_OK) { // some shard's scan failed: nobody launches
                        split[i] = 0;
                        if (--running == 0) cv_done.notify_all();
                        continue;
                    }
                }
                const int r = decode_dev_shard(i);
                std::lock_guard<std::mutex> lk(m);
                if (r != NRLDPC_OK && rc == NRLDPC_OK) { rc = r; err = nrldpc_last_error(); }
                split[i] = dv_batch[i] > 0 ? dv_batch[i] : 0;
            } else for (;;) {
                int k;
                {
                    std::lock_guard<std::mutex> lk(m);
                    if (next >= nchunks || rc != NRLDPC_OK) break;
                    k = next++;
                }
                const int c0 = k * chunk, n = std::min(chunk, batch - c0);
                const int r = decode_host(hs[i], llr + (size_t)c0 * ncw * eb, n, hard + (size_t)c0 * (packed ? (K + 7) / 8 : K),
                                          iters ? iters + c0 : nullptr, nullptr, packed, nl_call);
                std::lock_guard<std::mutex> lk(m);
                if (r != NRLDPC_OK && rc == NRLDPC_OK) { rc = r; err = nrldpc_last_error(); }
                split[i] += n;
            }
            std::lock_guard<std::mutex> lk(m);
            if (--running == 0) cv_done.notify_all();
        }
    }
};

extern "C" {

int nrldpc_pool_create(const nrldpc_cfg* cfg, const int32_t* device_ids, int32_t n_devices, int32_t chunks_per_device,
                       nrldpc_pool_handle* out) {
    NRLDPC_API_BEGIN
    if (!cfg || !device_ids || !out) return fail(NRLDPC_ERR_ARG, "null cfg/device_ids/out");
    *out = nullptr;
    // before anything reads *cfg as this library's struct: a caller built against another revision may have passed a smaller one
    if (cfg->struct_size != (uint32_t)sizeof(nrldpc_cfg))
        return fail(NRLDPC_ERR_ARG, "nrldpc_cfg.struct_size does not match this library (set it to sizeof(nrldpc_cfg); ABI revision mismatch?)");
    if (n_devices < 1 || n_devices > 64) return fail(NRLDPC_ERR_ARG, "n_devices must be in 1..64");
    if (chunks_per_device < 1 || chunks_per_device > 64) return fail(NRLDPC_ERR_ARG, "chunks_per_device must be in 1..64");
    nrldpc_pool* p = new nrldpc_pool();
    p->chunks_per_device = chunks_per_device;
    for (int i = 0; i < n_devices; ++i) {
        nrldpc_cfg c = *cfg;
        c.device_id = device_ids[i];
        nrldpc_handle h = nullptr;
        const int rc = nrldpc_create(&c, &h);
        if (rc != NRLDPC_OK) {
            for (auto q : p->hs) nrldpc_destroy(q);
            delete p;
            return rc; // text of nrldpc_create's failure is already in place
        }
        h->host_threads_hint = std::max(2, 16 / n_devices);
        p->hs.push_back(h);
    }
    const nrldpc::Schedule& s = p->hs[0]->sched;
    p->ncw = (size_t)s.g.ncols * s.Z; p->K = (size_t)s.g.kb * s.Z;
    p->eb = cfg->llr_dtype == NRLDPC_LLR_F64 ? 8 : cfg->llr_dtype == NRLDPC_LLR_F16 ? 2 : 4; // in the caller's array
    p->layers = p->hs[0]->layers;
    p->split.assign(n_devices, 0);
    p->streams.assign(n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) p->th.emplace_back([p, i] { p->worker(i); });
    *out = p;
    return NRLDPC_OK;
    NRLDPC_API_END
}

} // extern "C"
namespace {
int pool_decode_host(nrldpc_pool_handle p, const void* llr, int32_t batch, uint8_t* hard, int32_t* iters_out, bool packed) {
    NRLDPC_API_BEGIN
    if (!p) return fail(NRLDPC_ERR_ARG, "null pool");
    if (batch < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (batch == 0) return NRLDPC_OK;
    if (!llr || !hard) return fail(NRLDPC_ERR_ARG, "null llr/hard pointer");
    std::unique_lock<std::mutex> lk(p->m);
    p->packed = packed;
    p->nl_call = p->layers;
    if (p->layers == NRLDPC_LAYERS_AUTO) { // once for the whole batch, before it is cut: a few short-lived threads over the caller's array
        const nrldpc::Schedule& s = p->hs[0]->sched;
        const int kind = p->eb == 8 ? NRLDPC_HQ_F64 : p->eb == 2 ? NRLDPC_HQ_F16 : NRLDPC_HQ_F32;
        const int nt = std::max(1, std::min({8, usable_cpus(), batch}));
        int best = s.g.kb + 3;
        std::vector<std::thread> ts;
        for (int t = 1; t < nt; ++t)
            ts.emplace_back([&, t] { nrldpc_top_block(llr, kind, (size_t)batch, (size_t)t, (size_t)nt, s.Z, s.g.ncols, s.g.kb + 4, &best); });
        nrldpc_top_block(llr, kind, (size_t)batch, 0, (size_t)nt, s.Z, s.g.ncols, s.g.kb + 4, &best);
        for (auto& t : ts) t.join();
        p->nl_call = layers_of_block(s, best);
    }
    const int want = (int)p->hs.size() * p->chunks_per_device;
    p->chunk = std::max(1, (batch + want - 1) / want);
    p->nchunks = (batch + p->chunk - 1) / p->chunk;
    p->llr = static_cast<const char*>(llr); p->hard = hard; p->iters = iters_out; p->batch = batch;
    p->next = 0; p->rc = NRLDPC_OK; p->err.clear();
    p->dev_job = false;
    std::fill(p->split.begin(), p->split.end(), 0);
    p->running = (int)p->th.size();
    ++p->gen;
    p->cv.notify_all();
    p->cv_done.wait(lk, [&] { return p->running == 0; });
    if (p->rc != NRLDPC_OK) return fail(p->rc, p->err);
    return NRLDPC_OK;
    NRLDPC_API_END
}
} // namespace
extern "C" {

int nrldpc_pool_decode(nrldpc_pool_handle p, const void* llr, int32_t batch, uint8_t* hard, int32_t* iters_out) {
    return pool_decode_host(p, llr, batch, hard, iters_out, false);
}
int nrldpc_pool_decode_packed(nrldpc_pool_handle p, const void* llr, int32_t batch, uint8_t* hard_packed, int32_t* iters_out) {
    return pool_decode_host(p, llr, batch, hard_packed, iters_out, true);
}

int nrldpc_pool_set_layers(nrldpc_pool_handle p, int32_t n_layers) {
    if (!p) return fail(NRLDPC_ERR_ARG, "null pool");
    std::lock_guard<std::mutex> lk(p->m);
    for (auto h : p->hs) {
        const int rc = nrldpc_set_layers(h, n_layers);
        if (rc != NRLDPC_OK) return rc;
    }
    p->layers = p->hs[0]->layers;
    return NRLDPC_OK;
}

int nrldpc_pool_decode_dev(nrldpc_pool_handle p, const void* const* d_llr, const int32_t* batch, uint8_t* const* d_hard,
                           int32_t* const* d_iters) {
    NRLDPC_API_BEGIN
    if (!p) return fail(NRLDPC_ERR_ARG, "null pool");
    if (!d_llr || !batch || !d_hard) return fail(NRLDPC_ERR_ARG, "null d_llr/batch/d_hard array");
    for (size_t i = 0; i < p->hs.size(); ++i)
        if (batch[i] > 0 && (!d_llr[i] || !d_hard[i])) return fail(NRLDPC_ERR_ARG, "null device pointer for a shard with work");
    std::unique_lock<std::mutex> lk(p->m);
    p->dev_job = true;
    p->nl_call = p->layers; p->scan_arrived = 0; p->scan_best = 0;
    p->dv_llr = d_llr; p->dv_batch = batch; p->dv_hard = d_hard; p->dv_iters = d_iters;
    p->rc = NRLDPC_OK; p->err.clear();
    std::fill(p->split.begin(), p->split.end(), 0);
    p->running = (int)p->th.size();
    ++p->gen;
    p->cv.notify_all();
    p->cv_done.wait(lk, [&] { return p->running == 0; });
    p->dev_job = false;
    if (p->rc != NRLDPC_OK) return fail(p->rc, p->err);
    return NRLDPC_OK;
    NRLDPC_API_END
}

int nrldpc_pool_set_timing(nrldpc_pool_handle p, int32_t enabled) {
    if (!p) return fail(NRLDPC_ERR_ARG, "null pool");
    std::lock_guard<std::mutex> lk(p->m);
    for (auto h : p->hs) {
        const int rc = nrldpc_set_timing(h, enabled);
        if (rc != NRLDPC_OK) return rc;
    }
    return NRLDPC_OK;
}

int nrldpc_pool_last_kernel_ms(nrldpc_pool_handle p, float* ms) {
    NRLDPC_API_BEGIN
    if (!p || !ms) return fail(NRLDPC_ERR_ARG, "null pool/out");
    std::lock_guard<std::mutex> lk(p->m);
    for (size_t i = 0; i < p->hs.size(); ++i) {
        nrldpc_handle h = p->hs[i];
        ms[i] = 0.0f;
        if (!h->timing) return fail(NRLDPC_ERR_ARG, "timing is not enabled (nrldpc_pool_set_timing)");
        if (!h->have_time) continue; // this shard has not launched since timing was enabled
        DEVICE_SCOPE(h);
        HIP_TRY(hipEventSynchronize(h->ev1));
        HIP_TRY(hipEventElapsedTime(&ms[i], h->ev0, h->ev1));
    }
    return NRLDPC_OK;
    NRLDPC_API_END
}

int nrldpc_pool_size(nrldpc_pool_handle p) { return p ? (int)p->hs.size() : 0; }

int nrldpc_pool_last_split(nrldpc_pool_handle p, int32_t* counts) {
    if (!p || !counts) return fail(NRLDPC_ERR_ARG, "null pool/counts");
    std::lock_guard<std::mutex> lk(p->m);
    for (size_t i = 0; i < p->split.size(); ++i) counts[i] = p->split[i];
    return NRLDPC_OK;
}

void nrldpc_pool_destroy(nrldpc_pool_handle p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->m);
        p->stop = true;
    }
    p->cv.notify_all();
    for (auto& t : p->th) t.join();
    for (size_t i = 0; i < p->hs.size(); ++i)
        if (p->streams[i]) { DeviceScope scope(p->hs[i]->cfg.device_id); (void)hipStreamDestroy(p->streams[i]); }
    for (auto h : p->hs) nrldpc_destroy(h);
    delete p;
}

static uint32_t crc_poly_for(int len, bool code_block) {
    // get_3gpp_crc_polynomial.m:3-14
    if (len == 16) return 0x11021u;
    return code_block ? 0x1800063u : 0x1864CFBu;
}

// x^n mod g as an L-bit register value
static uint32_t xpow_mod(uint32_t poly, int L, long n) {
    const uint32_t top = 1u << (L - 1), mask = (1u << L) - 1u;
    uint32_t v = 1u; // x^0
    for (long i = 0; i < n; ++i) {
        const bool carry = v & top;
        v = (v << 1) & mask;
        if (carry) v ^= poly & mask;
    }
    return v;
}

static void xpow_matrix(uint32_t* M, uint32_t poly, int L, long n) {
    const uint32_t top = 1u << (L - 1), mask = (1u << L) - 1u;
    uint32_t v = xpow_mod(poly, L, n); // x^n
    for (int b = 0; b < 24; ++b) {
        M[b] = (b < L) ? v : 0u; // x^(n+b)
        const bool carry = v & top;
        v = (v << 1) & mask;
        if (carry) v ^= poly & mask;
    }
}

// Plan for wave-parallel CRCs over messages of up to `len` bits; seg / seg_tail are the segment lengths of
// the cross-code-block fold (nrldpc_crc.hip).  An odd chunk keeps the 64 lanes' LDS reads on distinct banks.
static void make_crc_plan(nrldpc::CrcPlan* pl, uint32_t poly, int L, int len, int seg, int seg_tail) {
    pl->poly = poly; pl->L = L;
    pl->chunk = ((len + 63) / 64) | 1;
    for (int s = 0; s < 6; ++s) xpow_matrix(pl->shiftmat[s], poly, L, (long)pl->chunk << s);
    xpow_matrix(pl->horner, poly, L, seg);
    xpow_matrix(pl->horner_tail, poly, L, seg_tail);
}

static int check_tb_params(const nrldpc_tb_params* p) {
    if (!p) return fail(NRLDPC_ERR_ARG, "null parameters");
    if (p->C < 1 || p->C > NRLDPC_MAX_C) return fail(NRLDPC_ERR_UNSUPPORTED, "C out of range (1..160)");
    if (nrldpc::set_index(p->Z) < 0 || (p->bg != 1 && p->bg != 2)) return fail(NRLDPC_ERR_UNSUPPORTED, "invalid BG / lifting size");
    if (p->Q_m < 1 || p->N_cb < 1 || p->N_cb > p->N || p->K_prime > p->K) return fail(NRLDPC_ERR_UNSUPPORTED, "inconsistent block parameters");
    return NRLDPC_OK;
}

static int make_rm_args(const nrldpc_tb_params* p, const float* d_g_tilde, int32_t n_tb, float* d_harq, void* d_cw_llr,
                        int32_t out_dtype, nrldpc::RmArgs* out) {
    // G == 0 is a legal draw of the reference's own sweep (testbench.m:35 with a small A): nothing was transmitted, the
    // decoder input is all zeros / fillers, and an empty g_tilde has no address
    if (!d_g_tilde && p->G > 0) return fail(NRLDPC_ERR_ARG, "null pointer");
    nrldpc::RmArgs& a = *out;
    memset(&a, 0, sizeof a);
    a.g = d_g_tilde; a.harq = d_harq; a.out = d_cw_llr; a.out_f16 = out_dtype == NRLDPC_LLR_F16;
    a.n_tb = n_tb; a.C = p->C; a.G = p->G; a.Z = p->Z; a.K = p->K; a.Kp = p->K_prime; a.N = p->N; a.N_cb = p->N_cb;
    a.k0 = p->k_0; a.Qm = p->Q_m;
    int off = 0;
    for (int r = 0; r < p->C; ++r) {
        if (p->E_r[r] < 0 || p->E_r[r] % p->Q_m) return fail(NRLDPC_ERR_UNSUPPORTED, "E_r must be a non-negative multiple of Q_m");
        a.E[r] = p->E_r[r]; a.off[r] = off; off += p->E_r[r];
    }
    if (off != p->G) return fail(NRLDPC_ERR_ARG, "sum(E_r) must equal G");
    return NRLDPC_OK;
}

int nrldpc_rate_recover_dev(const nrldpc_tb_params* p, const float* d_g_tilde, int32_t n_tb, float* d_harq,
                            void* d_cw_llr, int32_t out_dtype, void* stream) {
    int rc = check_tb_params(p);
    if (rc) return rc;
    if (n_tb < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (n_tb == 0) return NRLDPC_OK;
    if (!d_cw_llr) return fail(NRLDPC_ERR_ARG, "null pointer");
    if (out_dtype != NRLDPC_LLR_F32 && out_dtype != NRLDPC_LLR_F16) return fail(NRLDPC_ERR_ARG, "out_dtype must be f32 or f16");
    nrldpc::RmArgs a;
    rc = make_rm_args(p, d_g_tilde, n_tb, d_harq, d_cw_llr, out_dtype, &a);
    if (rc) return rc;
    hipError_t e = nrldpc::launch_rate_recover(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hipfail(e, "rate-recovery kernel launch");
    return NRLDPC_OK;
}

static int crc_check_common(const nrldpc_tb_params* p, const uint8_t* d_c_hat, int32_t n_tb, uint8_t* d_b_hat, int32_t* d_ok,
                            int32_t* d_cb_pass, const uint8_t* cbgti_flags, int keep_b_hat, int sticky, void* stream) {
    int rc = check_tb_params(p);
    if (rc) return rc;
    if (n_tb < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (n_tb == 0) return NRLDPC_OK;
    if (!d_c_hat || !d_b_hat || !d_ok) return fail(NRLDPC_ERR_ARG, "null pointer");
    if (sticky && !d_cb_pass) return fail(NRLDPC_ERR_ARG, "the sticky code-block flags (d_cb_pass) are required");
    if ((p->tb_crc_len != 16 && p->tb_crc_len != 24) || (p->cb_crc_len != 0 && p->cb_crc_len != 24))
        return fail(NRLDPC_ERR_UNSUPPORTED, "CRC lengths must be 16/24 (TB) and 0/24 (CB)");
    if (p->B != p->A + p->tb_crc_len || p->C * (p->K_prime - p->cb_crc_len) != p->B)
        return fail(NRLDPC_ERR_ARG, "B must equal A + L and C*(K' - L_cb)");
    nrldpc::CrcArgs a;
    a.c_hat = d_c_hat; a.b_hat = d_b_hat; a.ok = d_ok; a.cb_pass = d_cb_pass;
    a.n_tb = n_tb; a.C = p->C; a.K = p->K; a.Kp = p->K_prime; a.Lcb = p->cb_crc_len; a.A = p->A; a.B = p->B;
    a.keep_b_hat = keep_b_hat ? 1 : 0; a.sticky = sticky ? 1 : 0;
    for (int r = 0; r < NRLDPC_MAX_C; ++r) a.cbgti[r] = (cbgti_flags && r < p->C) ? (cbgti_flags[r] ? 1 : 0) : 1;
    const int pay = p->K_prime - p->cb_crc_len;
    make_crc_plan(&a.tb, crc_poly_for(p->tb_crc_len, false), p->tb_crc_len, pay, pay, pay);
    make_crc_plan(&a.cb, crc_poly_for(24, true), 24, p->K_prime, 0, 0);
    hipError_t e = nrldpc::launch_crc_check(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hipfail(e, "CRC kernel launch");
    return NRLDPC_OK;
}

int nrldpc_crc_check_dev(const nrldpc_tb_params* p, const uint8_t* d_c_hat, int32_t n_tb, uint8_t* d_b_hat,
                         int32_t* d_ok, int32_t* d_cb_pass, void* stream) {
    return crc_check_common(p, d_c_hat, n_tb, d_b_hat, d_ok, d_cb_pass, nullptr, 0, 0, stream);
}

int nrldpc_crc_check_harq_dev(const nrldpc_tb_params* p, const uint8_t* d_c_hat, int32_t n_tb, uint8_t* d_b_hat,
                              int32_t* d_ok, int32_t* d_cb_pass, const uint8_t* cbgti_flags, int32_t keep_b_hat,
                              void* stream) {
    return crc_check_common(p, d_c_hat, n_tb, d_b_hat, d_ok, d_cb_pass, cbgti_flags, keep_b_hat, 1, stream);
}

int nrldpc_crc_attach_dev(const nrldpc_tb_params* p, const uint8_t* d_a, int32_t n_tb, uint8_t* d_c, void* stream) {
    int rc = check_tb_params(p);
    if (rc) return rc;
    if (n_tb < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (n_tb == 0) return NRLDPC_OK;
    if (!d_a || !d_c) return fail(NRLDPC_ERR_ARG, "null pointer");
    if ((p->tb_crc_len != 16 && p->tb_crc_len != 24) || (p->cb_crc_len != 0 && p->cb_crc_len != 24))
        return fail(NRLDPC_ERR_UNSUPPORTED, "CRC lengths must be 16/24 (TB) and 0/24 (CB)");
    if (p->B != p->A + p->tb_crc_len || p->C * (p->K_prime - p->cb_crc_len) != p->B)
        return fail(NRLDPC_ERR_ARG, "B must equal A + L and C*(K' - L_cb)");
    nrldpc::CrcAttachArgs a;
    a.a = d_a; a.c = d_c; a.n_tb = n_tb; a.C = p->C; a.K = p->K; a.Kp = p->K_prime; a.Lcb = p->cb_crc_len;
    a.A = p->A; a.B = p->B;
    const int pay = p->K_prime - p->cb_crc_len;
    if (pay < p->tb_crc_len) return fail(NRLDPC_ERR_UNSUPPORTED, "code block shorter than the transport-block CRC");
    make_crc_plan(&a.tb, crc_poly_for(p->tb_crc_len, false), p->tb_crc_len, pay, pay, pay - p->tb_crc_len);
    make_crc_plan(&a.cb, crc_poly_for(24, true), 24, pay, 0, 0);
    hipError_t e = nrldpc::launch_crc_attach(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hipfail(e, "CRC attach kernel launch");
    return NRLDPC_OK;
}

int nrldpc_rate_match_dev(const nrldpc_tb_params* p, const uint8_t* d_cw, int32_t n_tb, uint8_t* d_g, void* stream) {
    int rc = check_tb_params(p);
    if (rc) return rc;
    if (n_tb < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (n_tb == 0) return NRLDPC_OK;
    if (p->G == 0) return NRLDPC_OK; // nothing to transmit (testbench.m:35 can draw G = 0); g has no address then
    if (!d_cw || !d_g) return fail(NRLDPC_ERR_ARG, "null pointer");
    nrldpc::TxRmArgs a;
    a.cw = d_cw; a.g = d_g; a.n_tb = n_tb; a.C = p->C; a.G = p->G; a.Z = p->Z; a.K = p->K; a.Kp = p->K_prime;
    a.N = p->N; a.N_cb = p->N_cb; a.k0 = p->k_0; a.Qm = p->Q_m;
    int off = 0;
    for (int r = 0; r < p->C; ++r) {
        if (p->E_r[r] < 0 || p->E_r[r] % p->Q_m) return fail(NRLDPC_ERR_UNSUPPORTED, "E_r must be a non-negative multiple of Q_m");
        a.E[r] = p->E_r[r]; a.off[r] = off; off += p->E_r[r];
    }
    if (off != p->G) return fail(NRLDPC_ERR_ARG, "sum(E_r) must equal G");
    hipError_t e = nrldpc::launch_rate_match(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hipfail(e, "rate-matching kernel launch");
    return NRLDPC_OK;
}

int nrldpc_awgn_llr_dev(const uint8_t* d_g, int64_t n_bits, int32_t Q_m, float EsN0_dB, uint64_t seed,
                        uint64_t first_symbol, float* d_g_tilde, void* stream) {
    if (Q_m != 1 && Q_m != 2 && Q_m != 4 && Q_m != 6 && Q_m != 8) return fail(NRLDPC_ERR_UNSUPPORTED, "Unsupported modulation");
    if (n_bits < 0 || n_bits % Q_m) return fail(NRLDPC_ERR_ARG, "n_bits must be a non-negative multiple of Q_m");
    if (n_bits == 0) return NRLDPC_OK;
    if (!d_g || !d_g_tilde) return fail(NRLDPC_ERR_ARG, "null pointer");
    if (!(EsN0_dB > -60.0f && EsN0_dB < 80.0f)) return fail(NRLDPC_ERR_ARG, "EsN0_dB out of range");
    nrldpc::ChanArgs a;
    a.g = d_g; a.llr = d_g_tilde; a.n_sym = n_bits / Q_m; a.seed = seed; a.first_symbol = first_symbol; a.Qm = Q_m;
    const double n0 = std::pow(10.0, -(double)EsN0_dB / 10.0); // plot_BLER_vs_SNR.m:106
    static const double mean_sq[5] = {1.0, 1.0, 5.0, 21.0, 85.0};  // mean(level^2) of a rail with 0,1,2,3,4 bits
    a.sigma = (float)std::sqrt(n0 / 2.0); a.inv_n0 = (float)(1.0 / n0);
    a.inv_norm = (float)(1.0 / std::sqrt(2.0 * mean_sq[Q_m / 2]));
    hipError_t e = nrldpc::launch_awgn_llr(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hipfail(e, "channel kernel launch");
    return NRLDPC_OK;
}

} // extern "C"
namespace nrldpc { // nrldpc_channel.hip (declared here, not in nrldpc_kernels.h: that header is part of the decoder kernels' identity, nrldpc_kernel_id)
hipError_t launch_payload_bits(uint64_t seed, uint64_t first_block, int32_t n_tb, int32_t A, uint8_t* out, hipStream_t stream);
}
extern "C" {

int nrldpc_payload_bits_dev(uint64_t seed, uint64_t first_block, int32_t n_tb, int32_t A, uint8_t* d_a, void* stream) {
    if (n_tb < 0 || A < 1) return fail(NRLDPC_ERR_ARG, "n_tb must be non-negative and A positive");
    if (n_tb == 0) return NRLDPC_OK;
    if (!d_a) return fail(NRLDPC_ERR_ARG, "null pointer");
    hipError_t e = nrldpc::launch_payload_bits(seed, first_block, n_tb, A, d_a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return hipfail(e, "payload kernel launch");
    return NRLDPC_OK;
}

int nrldpc_encode_dev(nrldpc_handle h, const uint8_t* d_info, int32_t batch, uint8_t* d_cw, void* stream) {
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    if (batch < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (batch == 0) return NRLDPC_OK;
    if (!d_info || !d_cw) return fail(NRLDPC_ERR_ARG, "null info/cw pointer");
    DEVICE_SCOPE(h);
    return encode_launch(h, d_info, batch, d_cw, static_cast<hipStream_t>(stream));
}

int nrldpc_encode(nrldpc_handle h, const uint8_t* info, int32_t batch, uint8_t* cw) {
    if (!h) return fail(NRLDPC_ERR_ARG, "null handle");
    if (batch < 0) return fail(NRLDPC_ERR_ARG, "negative batch");
    if (batch == 0) return NRLDPC_OK;
    if (!info || !cw) return fail(NRLDPC_ERR_ARG, "null info/cw pointer");
    DEVICE_SCOPE(h);
    const nrldpc::Schedule& s = h->sched;
    const size_t ncw = (size_t)s.g.ncols * s.Z, K = (size_t)s.g.kb * s.Z;
    HIP_TRY(h->s_bits.reserve((size_t)batch * K));
    HIP_TRY(h->s_hard.reserve((size_t)batch * ncw));
    HIP_TRY(hipMemcpyAsync(h->s_bits.p, info, (size_t)batch * K, hipMemcpyHostToDevice, nullptr));
    int rc = encode_launch(h, h->s_bits.p, batch, h->s_hard.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(cw, h->s_hard.p, (size_t)batch * ncw, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return NRLDPC_OK;
}

} // extern "C"
