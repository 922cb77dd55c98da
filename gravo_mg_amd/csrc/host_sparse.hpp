// host_sparse.hpp -- dependency-free host sparse containers and the Galerkin triple product.
//
// The reference keeps everything in Eigen::SparseMatrix<double> (CSC, int32, sorted inner indices;
// gravomg/include/gravomg/multigrid_solver.h:105-108).  Eigen is not available here, so the host
// side owns a minimal compressed-storage type.  A `Compressed` is "outer-compressed": read as CSC
// it is the matrix, read as CSR it is the transpose.  System matrices are symmetric, so the same
// arrays serve both views (the reference's Gauss-Seidel relies on exactly that,
// gravomg/src/multigrid_solver.cpp:1200-1208).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <sched.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace gmg {

// std::allocator whose value-less construct() default-initialises: resize(n) of a vector of ints / doubles allocates
// without writing, so a 36-70 MB index / value array is first touched by whoever fills it (the threaded copies) instead
// of being zero-filled, page fault by page fault, on the calling thread (~0.17 ms per MB).
// Large blocks are 2 MiB-aligned and marked MADV_HUGEPAGE: first-touching fresh 4 KiB pages costs ~0.17 ms per MB and
// does not scale with threads (the faults serialise on the process's memory map), so a 36 MB pattern fetched from the
// device took 8 ms where the copy itself needs 1.5.
template <class T>
struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    using std::allocator<T>::allocator;
    static constexpr size_t kHuge = (size_t)2 << 20;
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes >= 2 * kHuge) {
            void* q = nullptr;
            const size_t rounded = (bytes + kHuge - 1) & ~(kHuge - 1);
            if (posix_memalign(&q, kHuge, rounded) == 0) {
                (void)madvise(q, rounded, MADV_HUGEPAGE);
                return static_cast<T*>(q);
            }
            throw std::bad_alloc();
        }
        void* q = std::malloc(std::max<size_t>(bytes, 1));
        if (!q) throw std::bad_alloc();
        return static_cast<T*>(q);
    }
    void deallocate(T* p, size_t) noexcept { std::free(p); }
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... Args> void construct(U* p, Args&&... args) { ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...); }
};
using IndexVec = std::vector<int, default_init_allocator<int>>;
using ValueVec = std::vector<double, default_init_allocator<double>>;

struct Compressed {
    int n_outer = 0;              // number of compressed vectors (columns for CSC, rows for CSR)
    int n_inner = 0;              // length of each compressed vector
    std::vector<int> ptr;         // n_outer + 1
    IndexVec idx;                 // nnz, ascending inside each outer vector
    ValueVec val;                 // nnz
    int nnz() const { return ptr.empty() ? 0 : ptr[n_outer]; }
    void assign(int nouter, int ninner, const int* p, const int* i, const double* v) {
        n_outer = nouter; n_inner = ninner;
        ptr.assign(p, p + nouter + 1);
        idx.assign(i, i + ptr[nouter]);
        val.assign(v, v + ptr[nouter]);
    }
};

// ---- the library's environment switches: ALL of them, read once per process (first use), none on a per-solve path ------------
//   GMG_HOST_THREADS=N      host threads of this process (default: the CPUs it may use, see cpu_budget)
//   LOCAL_WORLD_SIZE=N      set by torch.distributed.run: ranks sharing the node; a rank takes 1/N of the CPUs
//   GMG_LDLT_THREADS=N      threads of the coarsest LDL^T (factorisation and back-substitution teams); 1 = no team
//   GMG_TRACE=setup,ldlt,ctor   phase timers of gmg_set_system / the host factorisation / the drop-in constructor on stderr
//   GMG_POLL=0              wait for device results with copy + hipStreamSynchronize instead of polling pinned memory
//   GMG_P2P_TIMEOUT_S=S     device-side time-out of a peer-to-peer exchange (default 4 s)
//   GMG_SEGV_BACKTRACE=1    native stack of a fatal signal on stderr (installed at load time, engine.hip)
//   GMG_P2P_FENCE_FREE=1    default of gmg_p2p_set_fences for new plans: 0 = the fence-free gfx942 / gfx950 publication (kernels.hip.hpp::publish_order)
//   GMG_PUBLISH_FENCED=1    device -> pinned host publications (residual sums, coarsest right-hand side) behind a system-scope release fence
//   GMG_P2P_SHARED_DEVICE=1 gmg_p2p_connect accepts ranks that sit on the SAME device (functional tests on a one-GPU box; exchange kernels of
//                           such ranks wait for each other on one device: milliseconds per exchange, never a measurement)
// Everything else that used to be an A/B switch is either a gmg_config field or gone.
struct EnvSwitches {
    int host_threads = 0, local_world = 0, ldlt_threads = 0;
    bool trace_setup = false, trace_ldlt = false, trace_ctor = false, poll = true, segv_backtrace = false;
    bool block_smallest_last = true;       // GMG_BLOCK_COLOURING=bfs: in-block colouring in breadth-first instead of smallest-last order (host_plan.hpp::make_block_ordering)
    double p2p_timeout_s = 0.0;
    bool p2p_fence_free = false, publish_fenced = false, p2p_shared_device = false;
    static const EnvSwitches& get() {
        static const EnvSwitches v = [] {
            EnvSwitches e;
            auto num = [](const char* name) { const char* s = std::getenv(name); return s ? std::atof(s) : -1.0; };
            if (num("GMG_HOST_THREADS") > 0) e.host_threads = (int)num("GMG_HOST_THREADS");
            if (num("LOCAL_WORLD_SIZE") > 1) e.local_world = (int)num("LOCAL_WORLD_SIZE");
            if (num("GMG_LDLT_THREADS") > 0) e.ldlt_threads = (int)num("GMG_LDLT_THREADS");
            if (const char* t = std::getenv("GMG_BLOCK_COLOURING")) e.block_smallest_last = std::string(t) != "bfs";
            if (const char* t = std::getenv("GMG_TRACE")) { const std::string s(t); e.trace_setup = s.find("setup") != std::string::npos; e.trace_ldlt = s.find("ldlt") != std::string::npos; e.trace_ctor = s.find("ctor") != std::string::npos; }
            e.poll = num("GMG_POLL") != 0.0;
            if (num("GMG_P2P_TIMEOUT_S") > 0) e.p2p_timeout_s = num("GMG_P2P_TIMEOUT_S");
            e.segv_backtrace = num("GMG_SEGV_BACKTRACE") > 0;
            e.p2p_fence_free = num("GMG_P2P_FENCE_FREE") > 0;
            e.publish_fenced = num("GMG_PUBLISH_FENCED") > 0;
            e.p2p_shared_device = num("GMG_P2P_SHARED_DEVICE") > 0;
            return e;
        }();
        return v;
    }
};

// CPUs this process may actually use: hardware threads, cut down to the scheduler affinity mask and to the cgroup CPU
// quota.  A container on a 256-thread host with a 16-CPU quota reports 256 hardware threads; running 64-128 threads
// there exhausts the quota within a period and the kernel then stalls the WHOLE process for tens of milliseconds
// (seen as set-ups three times slower than usual and as DMAs that start late on some boxes).
inline int cpu_budget() {
    static const int v = [] {
        long t = (long)std::thread::hardware_concurrency();
        if (t <= 0) t = 1;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) t = std::min<long>(t, c); }
        auto apply_v2 = [&](const std::string& file) {
            if (FILE* f = std::fopen(file.c_str(), "r")) {
                char q[64]; long long period = 0;
                if (std::fscanf(f, "%63s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
                    const long long quota = std::atoll(q);
                    if (quota > 0) t = std::min<long>(t, (long)std::max<long long>(1, (quota + period - 1) / period));
                }
                std::fclose(f);
            }
        };
        auto read_ll = [](const char* file, long long& out) {
            FILE* f = std::fopen(file, "r");
            if (!f) return false;
            const bool ok = std::fscanf(f, "%lld", &out) == 1;
            std::fclose(f);
            return ok;
        };
        apply_v2("/sys/fs/cgroup/cpu.max");
        if (FILE* f = std::fopen("/proc/self/cgroup", "r")) {          // cgroup v2 line "0::/path": every ancestor's limit applies
            char line[1024];
            while (std::fgets(line, sizeof(line), f)) {
                if (std::strncmp(line, "0::", 3) != 0) continue;
                std::string path(line + 3);
                while (!path.empty() && (path.back() == '\n' || path.back() == '/')) path.pop_back();
                while (!path.empty()) {
                    apply_v2("/sys/fs/cgroup" + path + "/cpu.max");
                    const size_t cut = path.find_last_of('/');
                    path = cut == std::string::npos ? std::string() : path.substr(0, cut);
                }
            }
            std::fclose(f);
        }
        long long quota = 0, period = 0;                                 // cgroup v1
        if (read_ll("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", quota) && read_ll("/sys/fs/cgroup/cpu/cpu.cfs_period_us", period) && quota > 0 && period > 0)
            t = std::min<long>(t, (long)std::max<long long>(1, (quota + period - 1) / period));
        // one process per GPU: the ranks of a node share its CPUs (torchrun exports LOCAL_WORLD_SIZE); every rank sizes its worker
        // pool, its staging threads and its back-substitution team to ITS share -- 8 ranks x 16 threads on a 16-CPU quota is the
        // throttling case above
        const EnvSwitches& env = EnvSwitches::get();
        if (env.local_world > 1) t = std::max<long>(1, t / env.local_world);
        if (env.host_threads > 0) t = env.host_threads;
        return (int)std::max<long>(1, t);
    }();
    return v;
}

inline int hw_threads() { return std::min(cpu_budget(), 128); }

// ---- worker pool -----------------------------------------------------------------------------------------------
// The host has hundreds of cores and the setup issues dozens of short parallel loops: spawning std::threads per loop
// costs milliseconds (20-30 us per thread).  One lazily created pool serves them all.  A caller runs the first range
// itself and, while waiting for the others, executes queued work (so loops may nest and pool threads may submit).
class WorkerPool {
public:
    static WorkerPool& instance() {
        static WorkerPool* p = new WorkerPool();      // leaked on purpose: no destruction order problems at exit
        return *p;
    }
    struct Batch {
        std::atomic<int> pending{0};
    };
    void submit(Batch& b, std::function<void()> fn) {
        b.pending.fetch_add(1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> g(m_);
            ensure_workers();
            q_.emplace_back(&b, std::move(fn));
        }
        cv_.notify_one();
    }
    void wait(Batch& b) {
        while (b.pending.load(std::memory_order_acquire) > 0) {
            std::pair<Batch*, std::function<void()>> job;
            {
                std::unique_lock<std::mutex> g(m_);
                if (q_.empty()) { g.unlock(); std::this_thread::yield(); continue; }
                job = std::move(q_.front());
                q_.pop_front();
            }
            job.second();
            job.first->pending.fetch_sub(1, std::memory_order_release);
        }
    }
    int size() const { return n_workers_; }

private:
    WorkerPool() {
        n_workers_ = std::max(1, std::min(cpu_budget(), 64));
        pthread_atfork(nullptr, nullptr, [] { instance().after_fork(); });
    }
    void after_fork() {         // the child of a fork() has no worker threads: start over
        new (&m_) std::mutex();
        new (&cv_) std::condition_variable();
        q_.clear();
        started_ = false;
    }
    void ensure_workers() {     // called with m_ held
        if (started_) return;
        started_ = true;
        for (int i = 0; i < n_workers_; ++i)
            std::thread([this] {
                for (;;) {
                    std::pair<Batch*, std::function<void()>> job;
                    {
                        std::unique_lock<std::mutex> g(m_);
                        cv_.wait(g, [this] { return !q_.empty(); });
                        job = std::move(q_.front());
                        q_.pop_front();
                    }
                    job.second();
                    job.first->pending.fetch_sub(1, std::memory_order_release);
                }
            }).detach();
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::pair<Batch*, std::function<void()>>> q_;
    int n_workers_ = 1;
    bool started_ = false;
};

// f(lo, hi, t) over [0, n) split into `nthreads` contiguous ranges run on the worker pool (range 0 on the caller);
// n < min_serial items run inline (raise or lower it with the weight of an item).
template <class F>
inline void parallel_ranges(int n, int nthreads, F&& f, int min_serial = 4096) {
    if (nthreads <= 1 || n < min_serial || n < 2) { f(0, n, 0); return; }
    WorkerPool& pool = WorkerPool::instance();
    WorkerPool::Batch batch;
    const int chunk = (n + nthreads - 1) / nthreads;
    for (int t = 1; t < nthreads; ++t) {
        const int lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        pool.submit(batch, [&f, lo, hi, t] { f(lo, hi, t); });
    }
    f(0, std::min(n, chunk), 0);
    pool.wait(batch);
}

// Transpose of the compressed layout (CSC <-> CSR of the same matrix); output inner indices sorted.
inline Compressed transpose(const Compressed& a) {
    Compressed t;
    t.n_outer = a.n_inner; t.n_inner = a.n_outer;
    t.ptr.assign((size_t)a.n_inner + 1, 0);
    const int nnz = a.nnz();
    t.idx.resize(nnz); t.val.resize(nnz);
    for (int p = 0; p < nnz; ++p) t.ptr[a.idx[p] + 1]++;
    for (int i = 0; i < a.n_inner; ++i) t.ptr[i + 1] += t.ptr[i];
    std::vector<int> next(t.ptr.begin(), t.ptr.end() - 1);
    for (int j = 0; j < a.n_outer; ++j)
        for (int p = a.ptr[j]; p < a.ptr[j + 1]; ++p) {
            int q = next[a.idx[p]]++;
            t.idx[q] = j; t.val[q] = a.val[p];
        }
    return t;
}

// Galerkin coarse operator  Ac = U^T * A * U   (gravomg/src/multigrid_solver.cpp:1387-1392).
//   A  : symmetric n x n, compressed (either view)
//   U  : n x nc prolongation in CSC (the reference's storage); U rows hold <= 3 entries
// Row-wise fused triple product: for coarse row p, Ac[p,q] = sum_{i in U^T[p]} u_ip sum_j a_ij u_jq.
// No n x nc intermediate is formed; coarse rows are independent -> threaded over p.
// Output: symmetric nc x nc Compressed with sorted indices (pattern == exact symbolic product).
inline Compressed galerkin_rap(const Compressed& A, const Compressed& Ucsc, int nthreads) {
    const int n = A.n_outer, nc = Ucsc.n_outer;
    Compressed Ucsr = transpose(Ucsc);     // rows of U: n vectors with <= 3 entries
    (void)n;
    std::vector<std::vector<int>> tidx(nthreads > 0 ? nthreads : 1);
    std::vector<std::vector<double>> tval(tidx.size());
    std::vector<int> rowcnt((size_t)nc, 0);
    std::vector<int> tlo(tidx.size() + 1, 0);
    int T = (int)tidx.size();
    if (nc < 4096) T = 1;
    int chunk = (nc + T - 1) / T;
    auto work = [&](int lo, int hi, int t) {
        std::vector<double> acc((size_t)nc, 0.0);
        std::vector<int> mark((size_t)nc, -1), list;
        auto& oi = tidx[t]; auto& ov = tval[t];
        for (int p = lo; p < hi; ++p) {
            list.clear();
            for (int a = Ucsc.ptr[p]; a < Ucsc.ptr[p + 1]; ++a) {
                const int i = Ucsc.idx[a];
                const double uip = Ucsc.val[a];
                for (int b = A.ptr[i]; b < A.ptr[i + 1]; ++b) {
                    const int j = A.idx[b];
                    const double w = uip * A.val[b];
                    for (int c = Ucsr.ptr[j]; c < Ucsr.ptr[j + 1]; ++c) {
                        const int q = Ucsr.idx[c];
                        if (mark[q] != p) { mark[q] = p; acc[q] = 0.0; list.push_back(q); }
                        acc[q] += w * Ucsr.val[c];
                    }
                }
            }
            std::sort(list.begin(), list.end());
            rowcnt[p] = (int)list.size();
            for (int q : list) { oi.push_back(q); ov.push_back(acc[q]); }
        }
    };
    if (T == 1) work(0, nc, 0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) {
            int lo = t * chunk, hi = std::min(nc, lo + chunk);
            if (lo >= hi) break;
            pool.emplace_back(work, lo, hi, t);
        }
        for (auto& th : pool) th.join();
    }
    Compressed C;
    C.n_outer = nc; C.n_inner = nc;
    C.ptr.assign((size_t)nc + 1, 0);
    for (int p = 0; p < nc; ++p) C.ptr[p + 1] = C.ptr[p] + rowcnt[p];
    C.idx.resize(C.ptr[nc]); C.val.resize(C.ptr[nc]);
    size_t off = 0;
    for (int t = 0; t < T; ++t) {
        std::copy(tidx[t].begin(), tidx[t].end(), C.idx.begin() + off);
        std::copy(tval[t].begin(), tval[t].end(), C.val.begin() + off);
        off += tidx[t].size();
    }
    return C;
}

}  // namespace gmg
