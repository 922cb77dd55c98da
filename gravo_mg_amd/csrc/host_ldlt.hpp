// host_ldlt.hpp -- sparse LDL^T for the coarsest level, kept on the host as the north star asks.
//
// Replaces the reference's `Eigen::SimplicialLDLT<Eigen::SparseMatrix<double>> coarsestSolver`
// (gravomg/include/gravomg/multigrid_solver.h:145; factor at gravomg/src/multigrid_solver.cpp:1401,
// back-substitution once per V-cycle at :1075).  Eigen is a third-party dependency that is not
// present; this is an independent implementation of the same mathematical object: a fill-reducing
// symmetric permutation (minimum degree on the explicit elimination graph -- the coarsest level has
// 1 000 ... ~8 000 unknowns) followed by a sparse LDL^T.  Two implementations: SparseLDLT (simplicial,
// up-looking, driven by the elimination tree: the compact reference) and SupernodalLDLT (dense panels, the one the
// engine uses: 3-4x faster factorisation and 2x faster back-substitution on the ~21-entries-per-row Galerkin operators).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <iterator>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <new>
#include <string>

#include "host_sparse.hpp"
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace gmg {

class SparseLDLT {
public:
    int n = 0;
    bool ok = false;
    std::vector<int> perm;       // new -> old
    std::vector<int> Lp, Li;     // strictly lower unit factor, by columns
    std::vector<double> Lx, D;

    // A: symmetric, compressed, sorted indices, full (both triangles) storage.
    // reuse_perm: the sparsity pattern is the one of the previous call (the engine's pattern cache says so): keep `perm`.
    bool factor(const Compressed& A, bool reuse_perm = false) {
        ok = false;
        if (!(reuse_perm && n == A.n_outer && (int)perm.size() == n)) {
            n = A.n_outer;
            if (n > kMinDegreeMax) nested_dissection(A); else min_degree(A);
        }
        std::vector<int> inv(n);
        for (int i = 0; i < n; ++i) inv[perm[i]] = i;
        // Upper triangle (incl. diagonal) of C = P A P^T, by columns, unsorted rows are fine.
        std::vector<int> Cp(n + 1, 0), Ci;
        std::vector<double> Cx;
        for (int k = 0; k < n; ++k) {
            int old = perm[k];
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p)
                if (inv[A.idx[p]] <= k) Cp[k + 1]++;
        }
        for (int k = 0; k < n; ++k) Cp[k + 1] += Cp[k];
        Ci.resize(Cp[n]); Cx.resize(Cp[n]);
        for (int k = 0; k < n; ++k) {
            int q = Cp[k], old = perm[k];
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) {
                int i = inv[A.idx[p]];
                if (i <= k) { Ci[q] = i; Cx[q] = A.val[p]; ++q; }
            }
        }
        // symbolic: elimination tree + column counts of L
        std::vector<int> parent(n, -1), flag(n, -1), lnz(n, 0);
        for (int k = 0; k < n; ++k) {
            flag[k] = k;
            for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
                int i = Ci[p];
                while (i < k && flag[i] != k) {
                    if (parent[i] < 0) parent[i] = k;
                    lnz[i]++;
                    flag[i] = k;
                    i = parent[i];
                }
            }
        }
        Lp.assign(n + 1, 0);
        for (int k = 0; k < n; ++k) Lp[k + 1] = Lp[k] + lnz[k];
        Li.assign(Lp[n], 0); Lx.assign(Lp[n], 0.0); D.assign(n, 0.0);
        // numeric, up-looking: row k of L is the solution of a sparse triangular system
        std::vector<double> y(n, 0.0);
        std::vector<int> pattern(n), fill(n, 0);
        std::fill(flag.begin(), flag.end(), -1);
        for (int k = 0; k < n; ++k) {
            int top = n;
            flag[k] = k;
            for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
                int i = Ci[p];
                y[i] += Cx[p];
                int len = 0;
                while (i < k && flag[i] != k) { pattern[len++] = i; flag[i] = k; i = parent[i]; }
                while (len > 0) pattern[--top] = pattern[--len];
            }
            double dk = y[k];
            y[k] = 0.0;
            for (; top < n; ++top) {
                int i = pattern[top];
                double yi = y[i];
                y[i] = 0.0;
                int pend = Lp[i] + fill[i];
                for (int p = Lp[i]; p < pend; ++p) y[Li[p]] -= Lx[p] * yi;
                double lki = yi / D[i];
                dk -= lki * yi;
                Li[pend] = k; Lx[pend] = lki;
                fill[i]++;
            }
            if (dk == 0.0 || !std::isfinite(dk)) return false;
            D[k] = dk;
        }
        ok = true;
        return true;
    }

    // x = A^{-1} b for one column; work must hold n doubles.
    void solve(const double* b, double* x, double* work) const {
        double* y = work;
        for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
        for (int j = 0; j < n; ++j) {
            double yj = y[j];
            for (int p = Lp[j]; p < Lp[j + 1]; ++p) y[Li[p]] -= Lx[p] * yj;
        }
        for (int j = 0; j < n; ++j) y[j] /= D[j];
        for (int j = n - 1; j >= 0; --j) {
            double s = y[j];
            for (int p = Lp[j]; p < Lp[j + 1]; ++p) s -= Lx[p] * y[Li[p]];
            y[j] = s;
        }
        for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    }

    // d right-hand sides at once (columns b + c*ldb -> x + c*ldx): every entry of L is loaded once and used d times.
    // Per column the operations and their order are those of solve(), so the results are bitwise the same.
    // work: n * min(d, 4) doubles.
    void solve_multi(const double* b, size_t ldb, double* x, size_t ldx, int d, double* work) const {
        for (int c0 = 0; c0 < d; c0 += 4) {
            const int dc = std::min(4, d - c0);
            switch (dc) {
                case 1: solve(b + c0 * ldb, x + c0 * ldx, work); break;
                case 2: solve_block<2>(b + c0 * ldb, ldb, x + c0 * ldx, ldx, work); break;
                case 3: solve_block<3>(b + c0 * ldb, ldb, x + c0 * ldx, ldx, work); break;
                default: solve_block<4>(b + c0 * ldb, ldb, x + c0 * ldx, ldx, work); break;
            }
        }
    }

    template <int DC>
    void solve_block(const double* b, size_t ldb, double* x, size_t ldx, double* y) const {
        for (int i = 0; i < n; ++i) for (int c = 0; c < DC; ++c) y[(size_t)i * DC + c] = b[c * ldb + perm[i]];
        for (int j = 0; j < n; ++j) {
            double yj[DC];
            for (int c = 0; c < DC; ++c) yj[c] = y[(size_t)j * DC + c];
            for (int p = Lp[j]; p < Lp[j + 1]; ++p) {
                const double l = Lx[p];
                double* t = y + (size_t)Li[p] * DC;
                for (int c = 0; c < DC; ++c) t[c] -= l * yj[c];
            }
        }
        for (int j = 0; j < n; ++j) for (int c = 0; c < DC; ++c) y[(size_t)j * DC + c] /= D[j];
        for (int j = n - 1; j >= 0; --j) {
            double sj[DC];
            for (int c = 0; c < DC; ++c) sj[c] = y[(size_t)j * DC + c];
            for (int p = Lp[j]; p < Lp[j + 1]; ++p) {
                const double l = Lx[p];
                const double* t = y + (size_t)Li[p] * DC;
                for (int c = 0; c < DC; ++c) sj[c] -= l * t[c];
            }
            for (int c = 0; c < DC; ++c) y[(size_t)j * DC + c] = sj[c];
        }
        for (int i = 0; i < n; ++i) for (int c = 0; c < DC; ++c) x[c * ldx + perm[i]] = y[(size_t)i * DC + c];
    }

    // the fill-reducing permutation alone (what factor() starts with)
    void compute_ordering(const Compressed& A) {
        n = A.n_outer;
        if (n > kMinDegreeMax) nested_dissection(A); else min_degree(A);
    }

    long factor_nnz() const { return (long)Lp.empty() ? 0 : Lp[n]; }

    static constexpr int kLeaf = 320;             // nested dissection stops at regions of this size
    static constexpr int kMinDegreeMax = 2500;    // up to here the exact minimum-degree ordering (bit-row elimination graph) gives ~6 % less fill

private:
    // Nested dissection with breadth-first level-set separators: O(E log n), fill O(n log n) on mesh-like graphs.
    // A region is split by the middle level of a BFS started at a pseudo-peripheral vertex (edges of a BFS only join
    // equal or adjacent levels, so a whole level separates); the two halves are ordered first, the separator last.
    // order(region) = order(a) ++ order(b) ++ separator: every region owns a contiguous slice of perm, so the two halves are ordered by
    // independent tasks (the first split walks the whole graph on one thread, the rest runs on up to 16); the result does not depend
    // on how the work was spread.
    struct NdShared {
        const Compressed* A;
        std::unique_ptr<std::atomic<int>[]> region;       // vertex -> id of the region it currently belongs to (-1: in a separator)
        std::vector<int> level;                           // BFS levels (entries of a region are only touched by the task that owns it)
        std::atomic<int> next_region{1};
        int* perm;
    };
    void nested_dissection(const Compressed& A) {
        perm.assign(n, 0);
        NdShared S;
        S.A = &A;
        S.region.reset(new std::atomic<int>[(size_t)std::max(n, 1)]);
        for (int i = 0; i < n; ++i) S.region[i].store(0, std::memory_order_relaxed);
        S.level.assign(n, -1);
        S.perm = perm.data();
        std::vector<int> all(n);
        for (int i = 0; i < n; ++i) all[i] = i;
        nd_order(S, std::move(all), 0, 0, 0);
    }
    static void nd_order(NdShared& S, std::vector<int> nodes, int id, int off, int depth) {
        const Compressed& A = *S.A;
        int* out = S.perm + off;
        auto in_region = [&](int w, int rid) { return S.region[w].load(std::memory_order_relaxed) == rid; };
        if ((int)nodes.size() <= kLeaf) {      // leaf region: exact minimum degree on its own subgraph
            const std::vector<int> ord = leaf_min_degree(A, nodes, S.region.get(), id);
            std::copy(ord.begin(), ord.end(), out);
            return;
        }
        std::vector<int> queue;
        queue.reserve(nodes.size());
        auto bfs = [&](int start) {      // levels inside region `id`; the visit order ends up in `queue`
            queue.clear(); queue.push_back(start); S.level[start] = 0;
            for (size_t h = 0; h < queue.size(); ++h) {
                const int v = queue[h];
                for (int p = A.ptr[v]; p < A.ptr[v + 1]; ++p) { const int w = A.idx[p]; if (in_region(w, id) && S.level[w] < 0) { S.level[w] = S.level[v] + 1; queue.push_back(w); } }
            }
        };
        // two halves: the first on another thread while the region is big and the tree of tasks still narrow
        auto both = [&](std::vector<int>&& first, int id1, int off1, std::vector<int>&& second, int id2, int off2) {
            if (depth < 4 && first.size() > 600 && second.size() > 600) {
                auto other = std::async(std::launch::async, [&S, id1, off1, depth, f = std::move(first)]() mutable { nd_order(S, std::move(f), id1, off1, depth + 1); });
                nd_order(S, std::move(second), id2, off2, depth + 1);
                other.get();
            } else {
                nd_order(S, std::move(first), id1, off1, depth + 1);
                nd_order(S, std::move(second), id2, off2, depth + 1);
            }
        };
        for (int v : nodes) S.level[v] = -1;
        bfs(nodes[0]);
        if (queue.size() < nodes.size()) {
            // disconnected region: peel this component off and handle both parts independently (order: the rest, then the component)
            std::vector<int> comp(queue), rest;
            const int idc = S.next_region.fetch_add(2), idr = idc + 1;
            for (int v : comp) S.region[v].store(idc, std::memory_order_relaxed);
            for (int v : nodes) if (S.level[v] < 0) { rest.push_back(v); S.region[v].store(idr, std::memory_order_relaxed); }
            const int nrest = (int)rest.size();
            both(std::move(rest), idr, off, std::move(comp), idc, off + nrest);
            return;
        }
        const int far = queue.back();
        for (int v : nodes) S.level[v] = -1;
        bfs(far);
        const int deep = S.level[queue.back()];
        if (deep < 2) { std::copy(nodes.begin(), nodes.end(), out); return; }   // clique-like: no separator
        // middle level by vertex count
        std::vector<int> cnt(deep + 1, 0);
        for (int v : queue) cnt[S.level[v]]++;
        int mid = 1, acc = cnt[0];
        while (mid < deep - 1 && acc + cnt[mid] < (int)queue.size() / 2) { acc += cnt[mid]; ++mid; }
        std::vector<int> a, b, sep;
        const int ida = S.next_region.fetch_add(2), idb = ida + 1;
        for (int v : queue) {
            if (S.level[v] < mid) { a.push_back(v); S.region[v].store(ida, std::memory_order_relaxed); }
            else if (S.level[v] > mid) { b.push_back(v); S.region[v].store(idb, std::memory_order_relaxed); }
            else { sep.push_back(v); S.region[v].store(-1, std::memory_order_relaxed); }
        }
        const int na = (int)a.size(), nb = (int)b.size();
        std::copy(sep.begin(), sep.end(), out + na + nb);        // the separator is eliminated last
        both(std::move(a), ida, off, std::move(b), idb, off + na);
    }

    // Exact minimum degree restricted to the vertices of one region (ids local to `nodes`).
    // (adjacency as bit rows over the region's own vertices, <= kLeaf / 64 words each, like min_degree below: eliminating v ORs its
    // row into each neighbour's.  Ties go to the vertex that comes first in `nodes`.  The sorted-list version this replaces spent
    // 1-3 ms per leaf in set unions; the orderings are the same.)
    __attribute__((target("popcnt"))) static std::vector<int> leaf_min_degree(const Compressed& A, const std::vector<int>& nodes, const std::atomic<int>* region, int id) {
        const int m = (int)nodes.size();
        const int W = (m + 63) / 64;
        std::vector<std::pair<int, int>> key(m);
        for (int i = 0; i < m; ++i) key[i] = {nodes[i], i};
        std::sort(key.begin(), key.end());
        auto find = [&](int g) { auto it = std::lower_bound(key.begin(), key.end(), std::make_pair(g, -1)); return it->second; };
        std::vector<uint64_t> bits((size_t)m * W, 0);
        auto row = [&](int v) { return bits.data() + (size_t)v * W; };
        for (int i = 0; i < m; ++i) {
            const int g = nodes[i];
            for (int p = A.ptr[g]; p < A.ptr[g + 1]; ++p) {
                const int w = A.idx[p];
                if (w != g && region[w].load(std::memory_order_relaxed) == id) { const int j = find(w); row(i)[j >> 6] |= 1ull << (j & 63); row(j)[i >> 6] |= 1ull << (i & 63); }
            }
        }
        std::vector<int> deg(m, 0), out;
        for (int v = 0; v < m; ++v) { const uint64_t* r = row(v); for (int q = 0; q < W; ++q) deg[v] += __builtin_popcountll(r[q]); }
        std::vector<char> done(m, 0);
        out.reserve(m);
        for (int step = 0; step < m; ++step) {
            int v = -1, best = m + 1;
            for (int j = 0; j < m; ++j) if (!done[j] && deg[j] < best) { best = deg[j]; v = j; }
            done[v] = 1;
            out.push_back(nodes[v]);
            const uint64_t* rv = row(v);
            for (int wq = 0; wq < W; ++wq) {
                uint64_t mask = rv[wq];
                while (mask) {
                    const int u = (wq << 6) + __builtin_ctzll(mask);
                    mask &= mask - 1;
                    uint64_t* ru = row(u);                   // adj[u] = (adj[u] | adj[v]) \ {u, v}
                    int d = deg[u];
                    for (int q = 0; q < W; ++q) {
                        const uint64_t add = rv[q] & ~ru[q];
                        if (add) { d += __builtin_popcountll(add); ru[q] |= add; }
                    }
                    ru[u >> 6] &= ~(1ull << (u & 63));
                    ru[v >> 6] &= ~(1ull << (v & 63));
                    deg[u] = d - 2;
                }
            }
        }
        return out;
    }

    // Minimum-degree ordering on the explicit elimination graph, adjacency kept as bit rows (n <= kMinDegreeMax, so a
    // row is <= 40 words and the whole graph sits in L2): eliminating v ORs its row into each neighbour's row.  Ties go
    // to the lowest index.  O(n^2/64 * mean degree) word operations -- a few ms at n = 2 000.
    __attribute__((target("popcnt"))) void min_degree(const Compressed& A) {
        const int W = (n + 63) / 64;
        std::vector<uint64_t> bits((size_t)n * W, 0);
        auto row = [&](int v) { return bits.data() + (size_t)v * W; };
        for (int j = 0; j < n; ++j)
            for (int p = A.ptr[j]; p < A.ptr[j + 1]; ++p) {
                const int i = A.idx[p];
                if (i != j) { row(j)[i >> 6] |= 1ull << (i & 63); row(i)[j >> 6] |= 1ull << (j & 63); }
            }
        std::vector<int> deg(n, 0);
        for (int v = 0; v < n; ++v) { const uint64_t* r = row(v); for (int w = 0; w < W; ++w) deg[v] += __builtin_popcountll(r[w]); }
        std::vector<char> done(n, 0);
        perm.resize(n);
        for (int step = 0; step < n; ++step) {
            int v = -1, best = n + 1;
            for (int j = 0; j < n; ++j) if (!done[j] && deg[j] < best) { best = deg[j]; v = j; }
            done[v] = 1;
            perm[step] = v;
            const uint64_t* rv = row(v);
            for (int w = 0; w < W; ++w) {
                uint64_t m = rv[w];
                while (m) {
                    const int u = (w << 6) + __builtin_ctzll(m);
                    m &= m - 1;
                    uint64_t* ru = row(u);        // adj[u] = (adj[u] | adj[v]) \ {u, v}; rv holds u, ru holds v
                    int d = deg[u];
                    for (int q = 0; q < W; ++q) {
                        const uint64_t add = rv[q] & ~ru[q];
                        if (add) { d += __builtin_popcountll(add); ru[q] |= add; }
                    }
                    ru[u >> 6] &= ~(1ull << (u & 63));
                    ru[v >> 6] &= ~(1ull << (v & 63));
                    deg[u] = d - 2;               // u itself (came in with rv) and v (was a neighbour) leave
                }
            }
        }
    }
};

// ---- a second thread for the back-substitution of a V-cycle ------------------------------------------------------
// A small team of spinning threads for the coarsest back-substitution.  A batch of independent jobs is handed over through two
// words the threads spin on (a hand-over costs a cache-line transfer, ~0.1-0.2 us; the worker pool's queue + condition variable
// would cost tens).  The helpers spin only while the team is ARMED -- for the duration of a solve -- and sleep on a condition
// variable otherwise; run() on an unarmed team executes every job on the caller.  Whoever claims a job runs it (the caller takes
// part): a helper that lost its core to another thread for a scheduler tick -- milliseconds -- costs nothing, the others and the
// caller do its share.  Results never depend on who ran what (the jobs are independent and the caller combines them in job order).
class SpinTeam {
public:
    explicit SpinTeam(int helpers = 1) {
        helpers = std::max(1, std::min(helpers, 15));
        th_.reserve((size_t)helpers);
        for (int i = 0; i < helpers; ++i) th_.emplace_back([this] { loop(); });
    }
    ~SpinTeam() {
        { std::lock_guard<std::mutex> lk(m_); quit_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    SpinTeam(const SpinTeam&) = delete;
    SpinTeam& operator=(const SpinTeam&) = delete;
    int helpers() const { return (int)th_.size(); }
    void arm() { { std::lock_guard<std::mutex> lk(m_); ++armed_; } cv_.notify_all(); }
    void disarm() { std::lock_guard<std::mutex> lk(m_); --armed_; }
    bool armed() const { return armed_.load(std::memory_order_acquire) > 0; }
    // Keep the helpers on cores that share their last-level cache with the CALLING thread (not the same core): the threads hand
    // a few hundred cache lines of the solution vector back and forth per solve, which costs several times more across L3 domains
    // (measured: 52 us per solve with the helper next door, 74 us across CCDs of an EPYC 9575F).  Best effort: topology from
    // sysfs, silently skipped where it is not readable or the affinity call is refused.
    void stay_near_caller() {
        const int cpu = sched_getcpu();
        if (cpu < 0 || cpu == near_cpu_) return;
        near_cpu_ = cpu;
        auto read_list = [](const std::string& path, std::vector<int>& out) {
            out.clear();
            FILE* f = std::fopen(path.c_str(), "r");
            if (!f) return;
            char buf[4096];
            if (std::fgets(buf, sizeof buf, f)) {
                const char* p = buf;
                while (*p) {
                    char* e = nullptr;
                    long a = std::strtol(p, &e, 10);
                    if (e == p) break;
                    long b = a;
                    if (*e == '-') { p = e + 1; b = std::strtol(p, &e, 10); }
                    for (long c = a; c <= b && out.size() < 4096; ++c) out.push_back((int)c);
                    p = *e == ',' ? e + 1 : e;
                    if (*e != ',') break;
                }
            }
            std::fclose(f);
        };
        const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(cpu);
        std::vector<int> l3, sib;
        read_list(base + "/cache/index3/shared_cpu_list", l3);
        read_list(base + "/topology/thread_siblings_list", sib);
        cpu_set_t set;
        CPU_ZERO(&set);
        int n = 0;
        for (int c : l3)
            if (c != cpu && std::find(sib.begin(), sib.end(), c) == sib.end() && c < CPU_SETSIZE) { CPU_SET(c, &set); ++n; }
        if (n >= (int)th_.size())          // (fewer cores next door than helpers: leave the placement to the scheduler)
            for (auto& t : th_) (void)pthread_setaffinity_np(t.native_handle(), sizeof set, &set);
    }
    // fn(arg, j) for j = 0 .. njobs - 1, each exactly once, on the helpers and the calling thread; returns when all are done
    void run(void (*fn)(void*, int), void* arg, int njobs) {
        if (njobs <= 0) return;
        if (!armed() || njobs == 1) { for (int j = 0; j < njobs; ++j) fn(arg, j); return; }
        fn_.store(fn, std::memory_order_relaxed); arg_.store(arg, std::memory_order_relaxed); njobs_.store((unsigned)njobs, std::memory_order_relaxed);
        const uint64_t tk = (uint64_t)(++ticket_) << 32;
        done_.store(tk, std::memory_order_relaxed);
        claim_.store(tk, std::memory_order_release);          // publishes fn_ / arg_ / njobs_ with the ticket
        work(ticket_);
        while (done_.load(std::memory_order_acquire) != (tk | (unsigned)njobs)) __builtin_ia32_pause();
        // close the batch: a late helper that still holds an old value of the claim word must fail its exchange once the next
        // batch's description (fn_ / arg_ / njobs_) is being written
        claim_.store(tk | 0xffffffffu, std::memory_order_release);
    }

private:
    // claim and run jobs of ticket `tk` until none is left (or a newer ticket shows up: then this thread is late, not needed)
    void work(unsigned tk) {
        uint64_t v = claim_.load(std::memory_order_acquire);
        for (;;) {
            if ((unsigned)(v >> 32) != tk) return;
            const unsigned j = (unsigned)v;
            // the batch's description is read through relaxed atomics (a late helper may look at it while run() writes the next one: that
            // read is then discarded, because its exchange on the closed claim word fails) and used only after a successful claim
            const unsigned nj = njobs_.load(std::memory_order_relaxed);
            void (*const fn)(void*, int) = fn_.load(std::memory_order_relaxed);
            void* const arg = arg_.load(std::memory_order_relaxed);
            if (j >= nj) return;                               // (njobs_ belongs to ticket tk: read after the acquire that showed tk)
            if (claim_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel, std::memory_order_acquire)) {
                fn(arg, (int)j);
                done_.fetch_add(1, std::memory_order_release);
                v = claim_.load(std::memory_order_acquire);
            }
        }
    }
    void loop() {
        unsigned seen = 0;
        for (;;) {
            if (armed_.load(std::memory_order_acquire) <= 0) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return quit_ || armed_.load(std::memory_order_acquire) > 0; });
                if (quit_) return;
                continue;
            }
            const unsigned tk = (unsigned)(claim_.load(std::memory_order_acquire) >> 32);
            if (tk != seen) { seen = tk; work(tk); }
            else __builtin_ia32_pause();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::atomic<int> armed_{0};
    bool quit_ = false;
    std::atomic<void (*)(void*, int)> fn_{nullptr};
    std::atomic<void*> arg_{nullptr};
    std::atomic<unsigned> njobs_{0};
    unsigned ticket_ = 0;
    int near_cpu_ = -1;
    alignas(64) std::atomic<uint64_t> claim_{0};      // (ticket << 32) | next unclaimed job
    alignas(64) std::atomic<uint64_t> done_{0};       // (ticket << 32) | finished jobs
    std::vector<std::thread> th_;                     // last member: started when everything above exists
};
using SpinHelper = SpinTeam;      // (the two-way version's name)

// ---- supernodal LDL^T -------------------------------------------------------------------------------------------
// Same factorisation, organised by supernodes (runs of columns with identical structure below the diagonal, stored as
// dense column-major panels) and computed left-looking with dense kernels.  The Galerkin coarsest operator has ~21
// entries per row, so nnz(L)/n is 100-150 and the elimination spends its time in dense-ish blocks: the simplicial code
// above runs at ~2 GFLOP/s (one scattered multiply-add per entry), the panels at several times that, and the
// back-substitution of a V-cycle streams values only (indices once per supernode) with contiguous inner loops.
// The inner kernels are compiled twice (baseline x86-64 and AVX2+FMA) and picked at run time.
class SupernodalLDLT {
public:
    int n = 0;
    bool ok = false;
    std::vector<int> perm;              // new -> old

    double phase_ms[3] = {0, 0, 0};     // ordering / symbolic / numeric of the last factor() (measurement aid)
    bool factor(const Compressed& A, bool reuse_perm = false) {
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
        ok = false;
        phase_ms[0] = phase_ms[1] = 0.0;
        if (!(reuse_perm && n == A.n_outer && (int)perm.size() == n && symbolic_ready_)) {
            auto t0 = clk::now();
            SparseLDLT ord;
            ord.compute_ordering(A);
            perm = ord.perm;
            n = A.n_outer;
            phase_ms[0] = ms(t0); t0 = clk::now();
            structure_error_ = false;
            symbolic(A);
            phase_ms[1] = ms(t0);
            if (structure_error_) return false;
        }
        auto t1 = clk::now();
        numeric(A);
        phase_ms[2] = ms(t1);
        return ok;
    }

    long factor_nnz() const { return nnz_l_; }
    static int numeric_threads() {
        // (a SpinTeam takes at most 15 helpers: more than 16 threads would make numeric() rebuild its team on every call)
        const int v = EnvSwitches::get().ldlt_threads;
        return std::min(16, v > 0 ? v : std::max(1, std::min(hw_threads(), 8)));
    }

    // Read the factor once (one word per cache line): after a factorisation the panels sit in the caches of whichever core
    // computed them, and the first back-substitutions on the solving thread pay for pulling them over (~250 us instead of ~65 us
    // at n = 1929 on a 256-core host).  The engine calls this while the device runs the way down of the first cycle.
    double warm() const {
        double acc = 0.0;
        for (size_t i = 0; i < pan_.size(); i += 8) acc += pan_[i];
        long idx = 0;
        for (size_t i = 0; i < rows_.size(); i += 16) idx += rows_[i];
        return acc + (double)idx;
    }

    // scratch of one single-column solve: per part of the elimination tree (+ the top) a gather buffer, per part an accumulator of
    // length n (zero on entry, zero again on exit)
    size_t scratch_doubles() const { return (size_t)groups_ * ((size_t)max_rows_ + 1 + (size_t)n); }

    void solve(const double* b, double* x, double* work) const {
        std::vector<double> t(scratch_doubles(), 0.0);       // own scratch: callable concurrently (the dense-inverse build does)
        solve_column(b, x, work, t.data(), nullptr);
    }

    // d right-hand sides (columns b + c*ldb -> x + c*ldx); work: n * d doubles.  Every column is an independent
    // single-column solve (so columns cannot interact and a column's result does not depend on d); with more than one
    // column they run concurrently on the worker pool -- the factor is read-only and shared.  helper (optional): a team of
    // spinning threads that shares the parts of the elimination tree of a single-column solve with the caller (same arithmetic
    // with or without it, whatever its size).
    void solve_multi(const double* b, size_t ldb, double* x, size_t ldx, int d, double* work, SpinTeam* helper = nullptr) const {
        const size_t nt = scratch_doubles();
        if (scratch_.size() < nt * (size_t)d) scratch_.resize(nt * (size_t)d, 0.0);
        if (d == 1) { solve_column(b, x, work, scratch_.data(), helper); return; }
        // several columns: one job per column.  With an armed team (inside a V-cycle solve) the columns go to its spinning threads -- a
        // hand-over costs a cache line, where waking worker-pool threads costs tens of microseconds per solve (d = 3 at n = 1929:
        // 125 us per cycle through the pool); otherwise through the worker pool.  Each column is the one-thread solve either way.
        struct ColJob { const SupernodalLDLT* self; const double* b; size_t ldb; double* x; size_t ldx; double* work; double* scratch; size_t nt; };
        ColJob job{this, b, ldb, x, ldx, work, scratch_.data(), nt};
        auto run = [](void* p, int c) {
            const ColJob* j = (const ColJob*)p;
            j->self->solve_column(j->b + (size_t)c * j->ldb, j->x + (size_t)c * j->ldx, j->work + (size_t)c * j->self->n, j->scratch + j->nt * (size_t)c, nullptr);
        };
        if (helper && helper->armed() && d <= kMaxCols) { solve_columns_staged(b, ldb, x, ldx, d, work, helper); return; }
        if (helper && helper->armed()) { helper->run(run, &job, d); return; }
        parallel_ranges(d, d, [&](int c0, int c1, int) { for (int c = c0; c < c1; ++c) run(&job, c); }, 2);
    }

    // d <= kMaxCols columns through the stages of the elimination tree TOGETHER (round 4): a stage's jobs are (group, column) pairs, so the team
    // has d times as many independent jobs per stage -- above all in the top of the tree, where a stage holds one or two chains and a
    // single column keeps one thread busy (d = 3 at n = 1 929: ~100 us per cycle with one thread per column, see the bench line).  Every
    // (group, column) job is the single-column arithmetic on that column's own buffers: results bitwise those of solve_column.
    static constexpr int kMaxCols = 4;
    struct MultiJob { const SupernodalLDLT* self; double* y[kMaxCols]; double* t[kMaxCols]; bool forward; const int* groups; int ngroups; int ncols; };
    static void run_part_multi(void* p, int idx) {
        MultiJob* j = (MultiJob*)p;
        const int c = idx / j->ngroups, q = idx - c * j->ngroups;
        PartJob one{j->self, j->y[c], j->t[c], j->forward, j->groups};
        run_part(&one, q);
    }
    // the PARTS (stage 0) of a multi-column solve: one job per part taking all the columns through every supernode of the part, column by
    // column -- the single-column arithmetic on each column's own buffers (same bits), but the panel is fetched once instead of once per
    // column (the parts are where the team is saturated: 8 parts x 3 columns on 8 threads; the top keeps one job per (chain, column))
    static void run_part_cols(void* p, int q) {
        MultiJob* j = (MultiJob*)p;
        const SupernodalLDLT* S = j->self;
        const int g = j->groups[q], d = j->ncols;
        const bool avx = has_avx2();
        const std::vector<int>& list = S->part_sn_[(size_t)g];
        if (j->forward) {
            for (int s : list) {
                const int f = S->sn_first_[s], w = S->sn_first_[s + 1] - f;
                const int* R = S->rows_.data() + S->rows_ptr_[s];
                const int r = S->rows_ptr_[s + 1] - S->rows_ptr_[s];
                const double* P = S->pan_.data() + S->pan_ptr_[s];
                const int k = S->own_rows_[s];
                for (int c = 0; c < d; ++c) {
                    double* y = j->y[c];
                    double* t = S->group_buffer(j->t[c], g);
                    double* acc = S->group_acc(j->t[c], g);
                    if (avx) sn_forward1_avx2(P, w + r, w, r, y + f, t); else sn_forward1_base(P, w + r, w, r, y + f, t);
                    for (int i = 0; i < k; ++i) y[R[i]] -= t[i];
                    for (int i = k; i < r; ++i) acc[R[i]] += t[i];
                }
            }
        } else {
            for (size_t qq = list.size(); qq-- > 0;) {
                const int s = list[qq];
                const int f = S->sn_first_[s], w = S->sn_first_[s + 1] - f;
                const int* R = S->rows_.data() + S->rows_ptr_[s];
                const int r = S->rows_ptr_[s + 1] - S->rows_ptr_[s];
                const double* P = S->pan_.data() + S->pan_ptr_[s];
                for (int c = 0; c < d; ++c) {
                    double* y = j->y[c];
                    double* t = S->group_buffer(j->t[c], g);
                    for (int i = 0; i < r; ++i) t[i] = y[R[i]];
                    if (avx) sn_backward1_avx2(P, w + r, w, r, y + f, t); else sn_backward1_base(P, w + r, w, r, y + f, t);
                }
            }
        }
    }
    void solve_columns_staged(const double* b, size_t ldb, double* x, size_t ldx, int d, double* work, SpinTeam* team) const {
        const int S = (int)stage_groups_.size();
        const size_t nt = scratch_doubles();
        MultiJob job{this, {}, {}, true, nullptr, 0, d};
        for (int c = 0; c < d; ++c) {
            job.y[c] = work + (size_t)c * n; job.t[c] = scratch_.data() + nt * (size_t)c;
            const double* bc = b + (size_t)c * ldb;
            for (int i = 0; i < n; ++i) job.y[c][i] = bc[perm[i]];
        }
        for (int st = 0; st < S; ++st) {
            job.groups = stage_groups_[st].data(); job.ngroups = (int)stage_groups_[st].size(); job.forward = true;
            if (st == 0) team->run(run_part_cols, &job, job.ngroups);       // (stage 0 = the parts: nothing to take in from earlier stages; one job per part, not per (part, column): measured in round 4)
            else team->run(run_part_multi, &job, job.ngroups * d);
        }
        for (int c = 0; c < d; ++c) for (int j = 0; j < n; ++j) job.y[c][j] /= D_[j];
        for (int st = S - 1; st >= 0; --st) {
            job.groups = stage_groups_[st].data(); job.ngroups = (int)stage_groups_[st].size(); job.forward = false;
            if (st == 0) team->run(run_part_cols, &job, job.ngroups);
            else team->run(run_part_multi, &job, job.ngroups * d);
        }
        for (int c = 0; c < d; ++c) {
            double* xc = x + (size_t)c * ldx;
            for (int i = 0; i < n; ++i) xc[perm[i]] = job.y[c][i];
        }
    }

    // share of the factor (panel entries) in the lightest / the heaviest part of the elimination tree / in the part above them
    void split_report(long out[3]) const {
        out[0] = out[1] = split_work_.empty() ? 0 : split_work_[0];
        for (int t = 0; t < parts_; ++t) { out[0] = std::min(out[0], split_work_[t]); out[1] = std::max(out[1], split_work_[t]); }
        out[2] = 0;
        for (int g = parts_; g < groups_ && g < (int)split_work_.size(); ++g) out[2] += split_work_[g];
    }
    int parts() const { return parts_; }
    // measurement aid: microseconds of the four phases of one single-column solve (parts down, top down, top up, parts up) and
    // the number of top supernodes
    void profile(const double* b, double* y, SpinTeam* team, int reps, double out[6]) const {
        using clk = std::chrono::steady_clock;
        auto us = [](clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
        std::vector<double> tt(scratch_doubles(), 0.0);
        double* t = tt.data();
        const int S = (int)stage_groups_.size();
        for (int k = 0; k < 6; ++k) out[k] = 0.0;
        for (int rep = 0; rep < reps; ++rep) {
            for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
            auto t0 = clk::now();
            run_stage(0, y, t, true, team);
            out[0] += us(t0); t0 = clk::now();
            for (int st = 1; st < S; ++st) run_stage(st, y, t, true, team);
            out[1] += us(t0); t0 = clk::now();
            for (int j = 0; j < n; ++j) y[j] /= D_[j];
            for (int st = S - 1; st >= 1; --st) run_stage(st, y, t, false, team);
            out[2] += us(t0); t0 = clk::now();
            run_stage(0, y, t, false, team);
            out[3] += us(t0);
        }
        for (int k = 0; k < 4; ++k) out[k] /= reps;
        size_t top_sn = 0;
        for (int g = parts_; g < groups_; ++g) top_sn += part_sn_[g].size();
        out[4] = (double)top_sn;
        out[5] = (double)(groups_ - parts_);
    }

    // ---- the factor as the DEVICE reads it (engine.hip: the dense inverse of the coarsest operator is built on the GPU, setup_kernels.hip.hpp::
    // coarse_inverse_tiles).  Supernodes are cut into CHUNKS of <= kChunk columns; a chunk's rows below its own columns are the later columns of
    // its supernode followed by the supernode's row list, its values row-major, kChunk per row (zero padded), its diagonal block a padded
    // kChunk x kChunk strictly lower triangle.  fdesc: first column of the subtree below the chunk's last column (the numbering is a postorder, a
    // subtree is a run of columns): L^-1 e_c is nonzero only in chunks whose subtree holds c.  Levels of the way up: a chunk comes after every
    // chunk that owns one of its rows (its ancestors); level 0 = the roots.  pattern: lists that depend on the structure only.
    static constexpr int kChunk = 8;
    struct DeviceFactor {
        int n = 0, nq = 0, nlev = 0;
        std::vector<int> q_col0, q_w, q_rptr, q_fdesc, rows, lev_ptr, lev_q, perm, first_q;      // first_q[c]: the chunk that holds column c
        // chunks of every level with at least `big_rows` rows below them (they come first in lev_q: sorted by rows, descending)
        std::vector<int> big_per_level(int big_rows) const {
            std::vector<int> out((size_t)nlev, 0);
            for (int v = 0; v < nlev; ++v)
                for (int k = lev_ptr[(size_t)v]; k < lev_ptr[(size_t)v + 1] && q_rptr[(size_t)lev_q[(size_t)k] + 1] - q_rptr[(size_t)lev_q[(size_t)k]] >= big_rows; ++k) ++out[(size_t)v];
            return out;
        }
        // the chunks a tile of `width` columns visits on the way down: the union of its columns' paths to the root, ascending
        void tile_paths(int width, std::vector<int>& tile_ptr, std::vector<int>& tile_q) const {
            const int nt = (n + width - 1) / width;
            tile_ptr.assign((size_t)nt + 1, 0); tile_q.clear();
            std::vector<int> mark((size_t)nq, -1);
            for (int t = 0; t < nt; ++t) {
                const size_t at = tile_q.size();
                for (int c = t * width; c < std::min(n, (t + 1) * width); ++c)
                    for (int q = first_q[(size_t)c]; q >= 0 && mark[(size_t)q] != t;) {
                        mark[(size_t)q] = t; tile_q.push_back(q);
                        q = q_rptr[q + 1] > q_rptr[q] ? first_q[(size_t)rows[(size_t)q_rptr[q]]] : -1;
                    }
                std::sort(tile_q.begin() + (long)at, tile_q.end());
                tile_ptr[(size_t)t + 1] = (int)tile_q.size();
            }
        }
        std::vector<double> vals, tri, dinv;
    };
    // Column c (factor numbering) of A^-1 in rows >= c, by the algorithm of the device kernel on the exported layout -- way down in push form along the
    // column's path to the root, way up in pull form level by level -- on ONE column: the host's check of the layout and of the schedule
    // (gmg_host_ldlt_probe; tests/test_host.py).  x: n doubles, rows < c are left zero.
    static void emulate_device_column(const DeviceFactor& E, int c, double* x) {
        const int n = E.n;
        std::fill(x, x + n, 0.0);
        x[c] = 1.0;
        for (int q = E.first_q[(size_t)c]; q >= 0;) {                                       // way down: the chunks on the path, ascending
            const int col0 = E.q_col0[(size_t)q], w = E.q_w[(size_t)q];
            double y[kChunk];
            for (int jj = 0; jj < kChunk; ++jj) y[jj] = jj < w ? x[col0 + jj] : 0.0;
            const double* T = E.tri.data() + (size_t)q * kChunk * kChunk;
            for (int jj = 0; jj < kChunk; ++jj) for (int ii = jj + 1; ii < kChunk; ++ii) y[ii] -= T[ii * kChunk + jj] * y[jj];
            for (int jj = 1; jj < w; ++jj) x[col0 + jj] = y[jj];
            for (int i = E.q_rptr[(size_t)q]; i < E.q_rptr[(size_t)q + 1]; ++i) {
                double acc = 0.0;
                for (int jj = 0; jj < kChunk; ++jj) acc += E.vals[(size_t)i * kChunk + jj] * y[jj];
                x[E.rows[(size_t)i]] -= acc;
            }
            q = E.q_rptr[(size_t)q + 1] > E.q_rptr[(size_t)q] ? E.first_q[(size_t)E.rows[(size_t)E.q_rptr[(size_t)q]]] : -1;
        }
        for (int v = 0; v < E.nlev; ++v)                                                    // way up: level by level, every chunk that reaches row c or beyond
            for (int k = E.lev_ptr[(size_t)v]; k < E.lev_ptr[(size_t)v + 1]; ++k) {
                const int q = E.lev_q[(size_t)k], col0 = E.q_col0[(size_t)q], w = E.q_w[(size_t)q];
                if (col0 + w - 1 < c) continue;
                double acc[kChunk];
                for (int jj = 0; jj < kChunk; ++jj) acc[jj] = 0.0;
                for (int i = E.q_rptr[(size_t)q]; i < E.q_rptr[(size_t)q + 1]; ++i) {
                    const double xi = x[E.rows[(size_t)i]];
                    for (int jj = 0; jj < kChunk; ++jj) acc[jj] -= E.vals[(size_t)i * kChunk + jj] * xi;
                }
                for (int jj = 0; jj < kChunk; ++jj) acc[jj] += jj < w ? x[col0 + jj] * E.dinv[(size_t)col0 + jj] : 0.0;
                const double* T = E.tri.data() + (size_t)q * kChunk * kChunk;
                for (int jj = kChunk - 2; jj >= 0; --jj) for (int ii = jj + 1; ii < kChunk; ++ii) acc[jj] -= T[ii * kChunk + jj] * acc[ii];
                for (int jj = 0; jj < w; ++jj) x[col0 + jj] = acc[jj];
            }
    }
    void export_device_factor(DeviceFactor& E) const {
        E.n = n;
        E.perm = perm;
        E.q_col0.clear(); E.q_w.clear(); E.q_rptr.assign(1, 0); E.rows.clear();
        E.first_q.assign((size_t)n, 0);
        size_t n_rows = 0;
        for (int s = 0; s < ns_; ++s) {
            const int f = sn_first_[s], w = sn_first_[s + 1] - f, r = rows_ptr_[s + 1] - rows_ptr_[s];
            for (int c0 = 0; c0 < w; c0 += kChunk) { const int wc = std::min(kChunk, w - c0); n_rows += (size_t)(w - c0 - wc) + r; }
        }
        E.rows.reserve(n_rows);
        for (int s = 0; s < ns_; ++s) {
            const int f = sn_first_[s], w = sn_first_[s + 1] - f, r = rows_ptr_[s + 1] - rows_ptr_[s];
            const int* R = rows_.data() + rows_ptr_[s];
            for (int c0 = 0; c0 < w; c0 += kChunk) {
                const int wc = std::min(kChunk, w - c0), q = (int)E.q_col0.size();
                E.q_col0.push_back(f + c0); E.q_w.push_back(wc);
                for (int j = 0; j < wc; ++j) E.first_q[(size_t)f + c0 + j] = q;
                for (int i = c0 + wc; i < w; ++i) E.rows.push_back(f + i);
                for (int i = 0; i < r; ++i) E.rows.push_back(R[i]);
                E.q_rptr.push_back((int)E.rows.size());
            }
        }
        E.nq = (int)E.q_col0.size();
        // values
        E.vals.assign(E.rows.size() * kChunk, 0.0);
        E.tri.assign((size_t)E.nq * kChunk * kChunk, 0.0);
        E.dinv.resize((size_t)n);
        for (int j = 0; j < n; ++j) E.dinv[j] = 1.0 / D_[j];
        {
            int q = 0;
            for (int s = 0; s < ns_; ++s) {
                const int f = sn_first_[s], w = sn_first_[s + 1] - f, r = rows_ptr_[s + 1] - rows_ptr_[s], ld = w + r;
                const double* P = pan_.data() + pan_ptr_[s];
                for (int c0 = 0; c0 < w; c0 += kChunk, ++q) {
                    const int wc = std::min(kChunk, w - c0);
                    double* T = E.tri.data() + (size_t)q * kChunk * kChunk;
                    for (int jj = 0; jj < wc; ++jj)
                        for (int ii = jj + 1; ii < wc; ++ii) T[ii * kChunk + jj] = P[(size_t)(c0 + jj) * ld + c0 + ii];
                    double* V = E.vals.data() + (size_t)E.q_rptr[q] * kChunk;
                    const int below = ld - (c0 + wc);                      // rows c0 + wc .. ld - 1 of the panel, in order
                    for (int jj = 0; jj < wc; ++jj) {
                        const double* col = P + (size_t)(c0 + jj) * ld + c0 + wc;
                        for (int i = 0; i < below; ++i) V[(size_t)i * kChunk + jj] = col[i];
                    }
                }
            }
        }
        // subtrees and levels (structure only)
        std::vector<int> fd((size_t)n);
        for (int j = 0; j < n; ++j) fd[j] = j;
        E.q_fdesc.assign((size_t)E.nq, 0);
        {   // parent of a column = its first row below the diagonal: the next column of its supernode, or the supernode's first row
            for (int s = 0; s < ns_; ++s) {
                const int f = sn_first_[s], l = sn_first_[s + 1] - 1;
                for (int j = f; j < l; ++j) fd[j + 1] = std::min(fd[j + 1], fd[j]);
                if (rows_ptr_[s + 1] > rows_ptr_[s]) { const int p = rows_[rows_ptr_[s]]; fd[p] = std::min(fd[p], fd[l]); }
            }
        }
        std::vector<int> lvl((size_t)E.nq, 0);
        int deepest = 0;
        for (int q = E.nq - 1; q >= 0; --q) {
            E.q_fdesc[q] = fd[(size_t)E.q_col0[q] + E.q_w[q] - 1];
            int m = -1;
            for (int i = E.q_rptr[q]; i < E.q_rptr[q + 1]; ++i) m = std::max(m, lvl[E.first_q[(size_t)E.rows[i]]]);
            lvl[q] = m + 1;
            deepest = std::max(deepest, lvl[q]);
        }
        E.nlev = E.nq ? deepest + 1 : 0;
        E.lev_ptr.assign((size_t)E.nlev + 1, 0);
        for (int q = 0; q < E.nq; ++q) E.lev_ptr[(size_t)lvl[q] + 1]++;
        for (int v = 0; v < E.nlev; ++v) E.lev_ptr[v + 1] += E.lev_ptr[v];
        E.lev_q.resize((size_t)E.nq);
        {   // within a level: the chunks with the most rows first (the waves of a tile take them round robin)
            std::vector<int> fill(E.lev_ptr.begin(), E.lev_ptr.end() - (E.nlev ? 1 : 0));
            for (int q = 0; q < E.nq; ++q) E.lev_q[(size_t)fill[lvl[q]]++] = q;
            for (int v = 0; v < E.nlev; ++v)
                std::stable_sort(E.lev_q.begin() + E.lev_ptr[v], E.lev_q.begin() + E.lev_ptr[v + 1],
                                 [&](int a, int b2) { return E.q_rptr[a + 1] - E.q_rptr[a] > E.q_rptr[b2 + 1] - E.q_rptr[b2]; });
        }
    }

private:
    static constexpr int kMaxWidth = 48;        // columns per supernode (panel stays in L1/L2)
    bool symbolic_ready_ = false;
    bool structure_error_ = false;              // symbolic(): a supernode's row list did not match its column count (never seen)
    long nnz_l_ = 0;
    int max_rows_ = 0;                           // longest below-diagonal row structure of a supernode
    // split of the elimination tree for the back-substitution (plan_split): parts_ sets of disjoint subtrees, and the part above
    // them ("top": their common ancestors), supernodes ascending in each.  parts_ follows from the factor alone (never from the
    // machine), so the arithmetic -- which subtree's contributions are subtracted in which order -- is the same everywhere
    // The top is a tree of CHAINS (a separator = a run of supernodes each with one child in the top); chains at the same depth below the
    // root chain are independent of each other exactly like the parts are, so the way down runs: the parts, then the deepest chains, ...,
    // then the root chain -- one hand-over per stage, what a group sends to rows above itself goes through its accumulator -- and the
    // way up the reverse.  At n = 6 608 (8 parts, 7 chains) the longest path through the top is 3 chains of its 7.
    int parts_ = 2;
    int groups_ = 3;                             // parts + chains of the top
    std::vector<std::vector<int>> part_sn_;      // supernodes of every group, ascending: [0 .. parts_) the parts, [parts_ .. groups_) the top's chains
    std::vector<std::vector<int>> stage_groups_; // groups that run side by side, in the order of the way down: [0] = the parts, ..., last = root chain(s)
    std::vector<std::vector<int>> group_cols_;   // columns of a chain
    std::vector<std::vector<int>> group_feeds_;  // groups of earlier stages (whose accumulators a chain takes in), in the order they are subtracted
    std::vector<int> own_rows_;                  // per supernode: leading rows of its structure that lie inside its own group
    // numeric factorisation: for every supernode the descendants that update it, ascending, with the position in the descendant's row
    // list where the rows inside this supernode's columns begin (symbolic); the supernodes of the top, ascending (plan_split)
    std::vector<int> upd_ptr_, upd_sn_, upd_pos_;
    std::vector<int> top_sn_;
    std::shared_ptr<SpinTeam> fteam_;            // the numeric factorisation's team (shared by copies of the factor: only one of them factors at a time)
    std::vector<long> split_work_;               // panel entries per group
    mutable std::vector<double> scratch_;        // gathered right-hand-side rows of one supernode (a handle is not thread-safe)
    std::vector<int> inv_;                       // old -> new
    std::vector<int> Cp_, Ci_;                   // upper triangle of P A P^T by columns (pattern)
    int ns_ = 0;
    std::vector<int> sn_first_;                  // ns + 1: first column of each supernode
    std::vector<int> sn_of_;                     // column -> supernode
    std::vector<int> rows_ptr_, rows_;           // below-diagonal row structure of each supernode (sorted)
    std::vector<size_t> pan_ptr_;                // offset of each panel in pan_
    // panels: (w + r) x w column-major, ld = w + r.  Storage from calloc: a new block is zero without being touched, so the first
    // factorisation faults its pages in once, while it fills them; a re-factorisation zeroes it on all threads first (a std::vector
    // touched the 8 MB of the n = 6 608 factor twice on one thread before any use: 4.5 of the 12 ms of the symbolic phase)
    struct PanelStore {
        double* p = nullptr; size_t n = 0;
        PanelStore() = default;
        PanelStore(const PanelStore& o) { *this = o; }
        PanelStore& operator=(const PanelStore& o) { if (this != &o) { resize_uninitialised(o.n); if (n) std::memcpy(p, o.p, n * sizeof(double)); fresh = false; } return *this; }
        ~PanelStore() { std::free(p); }
        bool fresh = false;      // just allocated with calloc: all zero without having been touched (big blocks come straight from the kernel)
        void resize_uninitialised(size_t m) { if (m != n) { std::free(p); p = m ? (double*)std::calloc(m, sizeof(double)) : nullptr; if (m && !p) throw std::bad_alloc(); n = m; fresh = true; } }
        double* data() { return p; } const double* data() const { return p; }
        size_t size() const { return n; }
        double operator[](size_t i) const { return p[i]; }
    };
    PanelStore pan_;
    std::vector<double> D_;

    // upper triangle of P A P^T by columns (row indices only) + elimination tree + column counts of L
    void upper_and_etree(const Compressed& A, std::vector<int>& parent, std::vector<int>& lnz, bool counts = true) {
        inv_.assign(n, 0);
        for (int i = 0; i < n; ++i) inv_[perm[i]] = i;
        Cp_.assign(n + 1, 0);
        for (int k = 0; k < n; ++k) {
            const int old = perm[k];
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) if (inv_[A.idx[p]] <= k) Cp_[k + 1]++;
        }
        for (int k = 0; k < n; ++k) Cp_[k + 1] += Cp_[k];
        Ci_.resize(Cp_[n]);
        for (int k = 0; k < n; ++k) {
            int q = Cp_[k];
            const int old = perm[k];
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) { const int i = inv_[A.idx[p]]; if (i <= k) Ci_[q++] = i; }
        }
        parent.assign(n, -1); lnz.assign(n, 0);
        if (!counts) {
            // the tree alone (what the postorder needs): ancestors with path compression, O(nnz(A) alpha) instead of the O(nnz(L)) walk
            std::vector<int> anc(n, -1);
            for (int k = 0; k < n; ++k)
                for (int p = Cp_[k]; p < Cp_[k + 1]; ++p) {
                    int i = Ci_[p];
                    while (i < k) {
                        const int next = anc[i];
                        anc[i] = k;
                        if (next < 0) { parent[i] = k; break; }
                        if (next == k) break;
                        i = next;
                    }
                }
            return;
        }
        {   // the tree as above ...
            std::vector<int> anc(n, -1);
            for (int k = 0; k < n; ++k)
                for (int p = Cp_[k]; p < Cp_[k + 1]; ++p) {
                    int i = Ci_[p];
                    while (i < k) {
                        const int next = anc[i];
                        anc[i] = k;
                        if (next < 0) { parent[i] = k; break; }
                        if (next == k) break;
                        i = next;
                    }
                }
        }
        // ... and the column counts of L without visiting its entries (Gilbert, Ng, Peyton 1994), valid because the numbering is a
        // postorder of the tree (every subtree is an interval [first[j], j]).  Row i of L is the subtree T_i spanned by the nonzeros of
        // row i of A below the diagonal; column j's count is the number of T_i that contain j.  Give +1 to every leaf of T_i, -1 to the
        // lowest common ancestor of consecutive leaves and -1 to parent(i): the weights inside subtree(j) then sum to 1 exactly when j
        // is in T_i.  Walking the columns in order, j is a leaf of T_i iff the previous nonzero of row i lies before first[j], and the
        // common ancestor with the previous leaf is the root of its set in a union-find in which finished columns hang below their
        // parents.  O(nnz(A) alpha(n)) instead of the O(nnz(L)) row-subtree walk (checked against that walk when it replaced it, round 3).
        std::vector<int> first(n), delta(n, 0), prevnz(n, -1), prevleaf(n, -1), uf(n);
        for (int j = 0; j < n; ++j) { first[j] = j; uf[j] = j; }
        for (int j = 0; j < n; ++j) if (parent[j] >= 0 && first[j] < first[parent[j]]) first[parent[j]] = first[j];
        auto find = [&](int v) { int r = v; while (uf[r] != r) r = uf[r]; while (uf[v] != r) { const int nx = uf[v]; uf[v] = r; v = nx; } return r; };
        for (int j = 0; j < n; ++j) {
            const int old = perm[j];
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) {
                const int i = inv_[A.idx[p]];
                if (i <= j) continue;
                if (prevnz[i] < first[j]) {                       // j is a leaf of T_i
                    delta[j]++;
                    if (prevleaf[i] >= 0) delta[find(prevleaf[i])]--;
                    prevleaf[i] = j;
                }
                prevnz[i] = j;
            }
            if (parent[j] >= 0) uf[j] = parent[j];
        }
        for (int i = 0; i < n; ++i) {
            if (prevnz[i] < 0) delta[i]++;                        // no nonzero below the diagonal in row i: T_i = {i}
            if (parent[i] >= 0) delta[parent[i]]--;
        }
        for (int j = 0; j < n; ++j) if (parent[j] >= 0) delta[parent[j]] += delta[j];
        for (int j = 0; j < n; ++j) lnz[j] = delta[j] - 1;
    }

    void symbolic(const Compressed& A) {
        std::vector<int> parent, lnz;
        upper_and_etree(A, parent, lnz, false);
        {   // postorder the elimination tree (same fill; makes every subtree, hence every supernode, a run of columns)
            std::vector<int> head(n, -1), nxt(n, -1), post, stack;
            for (int j = n - 1; j >= 0; --j) if (parent[j] >= 0) { nxt[j] = head[parent[j]]; head[parent[j]] = j; }
            post.reserve(n);
            for (int root = 0; root < n; ++root) {
                if (parent[root] >= 0) continue;
                stack.push_back(root);
                while (!stack.empty()) {
                    const int v = stack.back();
                    const int c = head[v];
                    if (c >= 0) { head[v] = nxt[c]; stack.push_back(c); }
                    else { post.push_back(v); stack.pop_back(); }
                }
            }
            std::vector<int> p2(n);
            for (int k = 0; k < n; ++k) p2[k] = perm[post[k]];
            perm.swap(p2);
            upper_and_etree(A, parent, lnz);
        }
        long nnz_l = 0;
        for (int k = 0; k < n; ++k) nnz_l += lnz[k];
        // fundamental supernodes: column j + 1 joins j's supernode when it is j's parent and has the same structure
        std::vector<int> first;
        int start = 0;
        for (int j = 0; j < n; ++j) {
            const bool last = j + 1 == n;
            const bool join = !last && parent[j] == j + 1 && lnz[j] == lnz[j + 1] + 1 && (j + 1 - start) < kMaxWidth;
            if (!join) { first.push_back(start); start = j + 1; }
        }
        first.push_back(n);
        // relaxed amalgamation: a supernode is merged into the one that follows it when that one starts at its parent column
        // and the explicit zeros this stores are few -- wider panels, fewer tiny ones (most fundamental supernodes are 1 wide)
        sn_first_.clear();
        {
            const int nf = (int)first.size() - 1;
            int cur_first = first[0];
            long cur_zeros = 0;
            for (int q = 0; q < nf; ++q) {
                const int lastc = first[q + 1] - 1;                  // last column of the (possibly merged) current supernode
                bool merge = false;
                if (q + 1 < nf && parent[lastc] == first[q + 1]) {
                    const int wp = first[q + 2] - first[q + 1];
                    const int lp = first[q + 2] - 1;
                    const long wc = first[q + 1] - cur_first;
                    const long extra = wc * ((long)(wp + lnz[lp]) - lnz[lastc]);       // zeros added to the child's columns
                    const long merged = (wc + wp) * (long)(wp + lnz[lp]) + wc * (wc - 1) / 2;
                    merge = wc + wp <= kMaxWidth && (cur_zeros + extra) * 4 <= merged;
                    if (merge) cur_zeros += extra;
                }
                if (!merge) { sn_first_.push_back(cur_first); cur_first = first[q + 1]; cur_zeros = 0; }
            }
        }
        ns_ = (int)sn_first_.size();
        sn_first_.push_back(n);
        sn_of_.assign(n, 0);
        rows_ptr_.assign(ns_ + 1, 0);
        for (int s = 0; s < ns_; ++s) {
            for (int j = sn_first_[s]; j < sn_first_[s + 1]; ++j) sn_of_[j] = s;
            rows_ptr_[s + 1] = rows_ptr_[s] + lnz[sn_first_[s + 1] - 1];
        }
        rows_.resize(rows_ptr_[ns_]);
        pan_ptr_.assign(ns_ + 1, 0);
        nnz_l_ = nnz_l;
        // Row structure of every supernode = structure of its last column l: the entries of A below l in the supernode's columns and
        // what its children hand up (a child of a column of the supernode that lies outside it is the LAST column of its own supernode,
        // so its structure is that supernode's row list).  Every list is read once by its parent: O(sum of the lists) instead of the
        // O(nnz(L)) column-by-column fill this replaces; the count must come out as lnz[l].
        {
            std::vector<int> kid_head(ns_, -1), kid_next(ns_, -1), mark(n, -1);
            for (int s = ns_ - 1; s >= 0; --s) {
                const int l = sn_first_[s + 1] - 1;
                if (parent[l] >= 0) { const int t = sn_of_[parent[l]]; kid_next[s] = kid_head[t]; kid_head[t] = s; }
            }
            for (int s = 0; s < ns_; ++s) {
                const int f = sn_first_[s], l = sn_first_[s + 1] - 1, w = l - f + 1;
                int* out = rows_.data() + rows_ptr_[s];
                const int cap = rows_ptr_[s + 1] - rows_ptr_[s];
                int cnt = 0;
                bool over = false;
                auto add = [&](int i) { if (i > l && mark[i] != s) { mark[i] = s; if (cnt < cap) out[cnt++] = i; else over = true; } };
                for (int j = f; j <= l; ++j) {
                    const int old = perm[j];
                    for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) add(inv_[A.idx[p]]);
                }
                for (int c = kid_head[s]; c >= 0; c = kid_next[c])
                    for (int q = rows_ptr_[c]; q < rows_ptr_[c + 1]; ++q) add(rows_[q]);
                if (over || cnt != cap) { symbolic_ready_ = false; ok = false; structure_error_ = true; return; }
                std::sort(out, out + cnt);
                pan_ptr_[s + 1] = pan_ptr_[s] + (size_t)(w + lnz[l]) * w;
            }
        }
        max_rows_ = 0;
        for (int q = 0; q < ns_; ++q) max_rows_ = std::max(max_rows_, rows_ptr_[q + 1] - rows_ptr_[q]);
        pan_.resize_uninitialised(pan_ptr_[ns_]);
        D_.assign(n, 0.0);
        {   // update lists: a descendant's sorted row list visits its ancestors' supernodes one after the other
            upd_ptr_.assign((size_t)ns_ + 1, 0);
            for (int pass = 0; pass < 2; ++pass) {
                std::vector<int> fill(pass ? upd_ptr_ : std::vector<int>());
                for (int dd = 0; dd < ns_; ++dd) {
                    const int* Rd = rows_.data() + rows_ptr_[dd];
                    const int rd = rows_ptr_[dd + 1] - rows_ptr_[dd];
                    for (int p = 0; p < rd;) {
                        const int t = sn_of_[Rd[p]];
                        if (pass) { upd_sn_[fill[t]] = dd; upd_pos_[fill[t]] = p; ++fill[t]; } else upd_ptr_[t + 1]++;
                        const int end = sn_first_[t + 1];
                        while (p < rd && Rd[p] < end) ++p;
                    }
                }
                if (!pass) {
                    for (int t = 0; t < ns_; ++t) upd_ptr_[t + 1] += upd_ptr_[t];
                    upd_sn_.assign((size_t)upd_ptr_[ns_], 0); upd_pos_.assign((size_t)upd_ptr_[ns_], 0);
                }
            }
        }
        plan_split();
        symbolic_ready_ = true;
    }

    // Parts of the elimination tree that can be back-substituted independently: sets of disjoint subtrees.  Start from the roots;
    // while one subtree holds more than 1.1 / parts_ of what is left, move its root to the "top" part and consider its children
    // instead; then deal the subtrees to parts_ bins, largest first.  (The etree is postordered: a subtree is a run of supernodes,
    // descendants first.)  parts_: one per ~kPartWork panel entries, 2 .. kMaxParts.
    static constexpr long kPartWork = 60000;
    static constexpr int kMaxParts = 8;
    void plan_split() {
        std::vector<int> parent(ns_, -1);
        std::vector<long> work(ns_, 0);
        std::vector<std::vector<int>> kids(ns_);
        long all = 0;
        for (int s = 0; s < ns_; ++s) {
            const int r = rows_ptr_[s + 1] - rows_ptr_[s], w = sn_first_[s + 1] - sn_first_[s];
            work[s] = (long)(w + r) * w;
            all += work[s];
            if (r > 0) parent[s] = sn_of_[rows_[rows_ptr_[s]]];
        }
        parts_ = (int)std::max<long>(2, std::min<long>(kMaxParts, (all + kPartWork / 2) / kPartWork));
        std::vector<long> sub(work);
        for (int s = 0; s < ns_; ++s) if (parent[s] >= 0) { sub[parent[s]] += sub[s]; kids[parent[s]].push_back(s); }
        std::vector<int> cand;
        for (int s = 0; s < ns_; ++s) if (parent[s] < 0) cand.push_back(s);
        std::vector<char> in_top(ns_, 0);
        for (;;) {
            long total = 0;
            int big = -1;
            for (int c : cand) { total += sub[c]; if (big < 0 || sub[c] > sub[big]) big = c; }
            if (big < 0 || sub[big] * 100 * parts_ <= total * 110 || kids[big].empty()) break;
            in_top[big] = 1;
            cand.erase(std::find(cand.begin(), cand.end(), big));
            cand.insert(cand.end(), kids[big].begin(), kids[big].end());
        }
        std::sort(cand.begin(), cand.end(), [&](int a, int b2) { return sub[a] != sub[b2] ? sub[a] > sub[b2] : a < b2; });
        std::vector<int> half(ns_, -1);              // supernode -> part, -1 = top
        std::vector<long> load(parts_, 0);
        std::vector<int> root_half(ns_, -1);
        for (int c : cand) {
            int t = 0;
            for (int q = 1; q < parts_; ++q) if (load[q] < load[t]) t = q;
            root_half[c] = t; load[t] += sub[c];
        }
        for (int s = ns_ - 1; s >= 0; --s) {          // parents before children
            if (in_top[s]) continue;
            half[s] = root_half[s] >= 0 ? root_half[s] : half[parent[s]];
        }
        // chains of the top: a top supernode continues its parent's chain when it is the parent's only child in the top, else it
        // starts a chain one level deeper (parents before children: descending order)
        std::vector<int> top_kids(ns_, 0), chain_depth;
        for (int s = 0; s < ns_; ++s) if (in_top[s] && parent[s] >= 0 && in_top[parent[s]]) top_kids[parent[s]]++;
        for (int s = ns_ - 1; s >= 0; --s) {
            if (!in_top[s]) continue;
            const int p = parent[s];
            if (p >= 0 && in_top[p] && top_kids[p] == 1) { half[s] = half[p]; continue; }
            half[s] = parts_ + (int)chain_depth.size();
            chain_depth.push_back((p >= 0 && in_top[p]) ? chain_depth[half[p] - parts_] + 1 : 0);
        }
        groups_ = parts_ + (int)chain_depth.size();
        int deepest = -1;
        for (int d : chain_depth) deepest = std::max(deepest, d);
        stage_groups_.assign((size_t)deepest + 2, std::vector<int>());
        for (int q = 0; q < parts_; ++q) stage_groups_[0].push_back(q);
        for (int g = parts_; g < groups_; ++g) stage_groups_[(size_t)(deepest - chain_depth[g - parts_]) + 1].push_back(g);      // deepest chains first
        for (auto& st : stage_groups_) std::sort(st.begin(), st.end());
        part_sn_.assign((size_t)groups_, std::vector<int>());
        group_cols_.assign((size_t)groups_, std::vector<int>());
        group_feeds_.assign((size_t)groups_, std::vector<int>());
        for (size_t st = 1; st < stage_groups_.size(); ++st)
            for (int g : stage_groups_[st])
                for (size_t e = 0; e < st; ++e) group_feeds_[g].insert(group_feeds_[g].end(), stage_groups_[e].begin(), stage_groups_[e].end());
        scratch_.clear();                            // (its layout follows max_rows_ / n / groups_: the accumulators must start from zero)
        split_work_.assign((size_t)groups_, 0);
        own_rows_.assign(ns_, 0);
        top_sn_.clear();
        for (int s = 0; s < ns_; ++s) {
            const int t = half[s];
            if (t >= parts_) top_sn_.push_back(s);
            part_sn_[t].push_back(s);
            split_work_[t] += work[s];
            const int* R = rows_.data() + rows_ptr_[s];
            const int r = rows_ptr_[s + 1] - rows_ptr_[s];
            int k = 0;
            while (k < r && half[sn_of_[R[k]]] == t) ++k;     // ancestors inside the group come first (ascending rows = the path to the root)
            own_rows_[s] = k;
            if (t >= parts_) for (int j = sn_first_[s]; j < sn_first_[s + 1]; ++j) group_cols_[t].push_back(j);
        }
    }

    // dense kernels, compiled for the baseline ISA and for AVX2+FMA
#define GMG_LDLT_KERNELS(SUFFIX, ATTR)                                                                                          \
    /* in-place LDL^T of the (m x w) panel P (ld): unit lower factor below the diagonal, d[] the pivots; false on a zero pivot */  \
    ATTR static bool panel_factor##SUFFIX(double* P, int ld, int m, int w, double* d) {                                        \
        for (int j = 0; j < w; ++j) {                                                                                            \
            double* cj = P + (size_t)j * ld;                                                                                     \
            const double dj = cj[j];                                                                                             \
            if (dj == 0.0 || !std::isfinite(dj)) return false;                                                                   \
            d[j] = dj;                                                                                                           \
            const double inv = 1.0 / dj;                                                                                         \
            for (int k = j + 1; k < w; ++k) {                                                                                    \
                double* ck = P + (size_t)k * ld;                                                                                 \
                const double f = cj[k] * inv;                                                                                    \
                for (int i = k; i < m; ++i) ck[i] -= cj[i] * f;                                                                  \
            }                                                                                                                    \
            for (int i = j + 1; i < m; ++i) cj[i] *= inv;                                                                        \
        }                                                                                                                        \
        return true;                                                                                                             \
    }                                                                                                                            \
    /* U (k x k1, ld k, lower trapezoid: rows i >= j of column j) = Lk (k x w, ld) diag(d) Lk[0..k1)^T */                          \
    ATTR static void panel_update##SUFFIX(const double* Lk, int ld, int k, int k1, int w, const double* d, double* U) {         \
        for (int j = 0; j < k1; ++j) {                                                                                           \
            double* uj = U + (size_t)j * k;                                                                                      \
            for (int i = j; i < k; ++i) uj[i] = 0.0;                                                                             \
            for (int c = 0; c < w; ++c) {                                                                                        \
                const double* lc = Lk + (size_t)c * ld;                                                                          \
                const double f = lc[j] * d[c];                                                                                   \
                for (int i = j; i < k; ++i) uj[i] += lc[i] * f;                                                                  \
            }                                                                                                                    \
        }                                                                                                                        \
    }                                                                                                                            \
    /* dot product with four independent partial sums (the explicit re-association lets it vectorise) */                        \
    ATTR static double dot4##SUFFIX(const double* a, const double* b, int m) {                                                  \
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;                                                                           \
        int i = 0;                                                                                                               \
        for (; i + 4 <= m; i += 4) { s0 += a[i] * b[i]; s1 += a[i + 1] * b[i + 1]; s2 += a[i + 2] * b[i + 2]; s3 += a[i + 3] * b[i + 3]; } \
        for (; i < m; ++i) s0 += a[i] * b[i];                                                                                    \
        return (s0 + s1) + (s2 + s3);                                                                                            \
    }                                                                                                                            \
    /* one right-hand side: forward step of a supernode, ys <- L_d^-1 ys (unit lower, w x w), t = L_b ys (r), in two pieces so that   \
       the rows of a big panel can be shared between threads.  The rectangular part takes EIGHT / FOUR columns of the panel per     \
       pass over t (one load and one store of t[i] per group instead of per column); every t[i] still accumulates its columns one   \
       after the other in column order -- the results are those of the one-column-per-pass loop, bit for bit, and a row's value     \
       does not depend on which range of rows it is computed with */                                                               \
    ATTR static void sn_forward_tri##SUFFIX(const double* P, int ld, int w, double* ys) {                                       \
        for (int j = 0; j < w; ++j) {                                                                                            \
            const double* cj = P + (size_t)j * ld;                                                                               \
            const double yj = ys[j];                                                                                             \
            for (int i = j + 1; i < w; ++i) ys[i] -= cj[i] * yj;                                                                 \
        }                                                                                                                        \
    }                                                                                                                            \
    /* rows [r0, r1) of t = L_b ys */                                                                                              \
    ATTR static void sn_forward_rect##SUFFIX(const double* P, int ld, int w, int r0, int r1, const double* ys, double* t) {     \
        int j = 0;                                                                                                               \
        if (w >= 8) {                                                                                                            \
            const double *c0 = P + w, *c1 = c0 + ld, *c2 = c1 + ld, *c3 = c2 + ld, *c4 = c3 + ld, *c5 = c4 + ld, *c6 = c5 + ld, *c7 = c6 + ld; \
            const double y0 = ys[0], y1 = ys[1], y2 = ys[2], y3 = ys[3], y4 = ys[4], y5 = ys[5], y6 = ys[6], y7 = ys[7];        \
            for (int i = r0; i < r1; ++i)                                                                                        \
                t[i] = (((((((0.0 + c0[i] * y0) + c1[i] * y1) + c2[i] * y2) + c3[i] * y3) + c4[i] * y4) + c5[i] * y5) + c6[i] * y6) + c7[i] * y7; \
            j = 8;                                                                                                               \
        } else if (w >= 4) {                                                                                                     \
            const double *c0 = P + w, *c1 = c0 + ld, *c2 = c1 + ld, *c3 = c2 + ld;                                               \
            const double y0 = ys[0], y1 = ys[1], y2 = ys[2], y3 = ys[3];                                                         \
            for (int i = r0; i < r1; ++i) t[i] = (((0.0 + c0[i] * y0) + c1[i] * y1) + c2[i] * y2) + c3[i] * y3;                    \
            j = 4;                                                                                                               \
        } else {                                                                                                                 \
            for (int i = r0; i < r1; ++i) t[i] = 0.0;                                                                            \
        }                                                                                                                        \
        for (; j + 8 <= w; j += 8) {                                                                                             \
            const double *c0 = P + (size_t)j * ld + w, *c1 = c0 + ld, *c2 = c1 + ld, *c3 = c2 + ld, *c4 = c3 + ld, *c5 = c4 + ld, *c6 = c5 + ld, *c7 = c6 + ld; \
            const double y0 = ys[j], y1 = ys[j + 1], y2 = ys[j + 2], y3 = ys[j + 3], y4 = ys[j + 4], y5 = ys[j + 5], y6 = ys[j + 6], y7 = ys[j + 7]; \
            for (int i = r0; i < r1; ++i)                                                                                        \
                t[i] = (((((((t[i] + c0[i] * y0) + c1[i] * y1) + c2[i] * y2) + c3[i] * y3) + c4[i] * y4) + c5[i] * y5) + c6[i] * y6) + c7[i] * y7; \
        }                                                                                                                        \
        for (; j + 4 <= w; j += 4) {                                                                                             \
            const double *c0 = P + (size_t)j * ld + w, *c1 = c0 + ld, *c2 = c1 + ld, *c3 = c2 + ld;                              \
            const double y0 = ys[j], y1 = ys[j + 1], y2 = ys[j + 2], y3 = ys[j + 3];                                             \
            for (int i = r0; i < r1; ++i) t[i] = (((t[i] + c0[i] * y0) + c1[i] * y1) + c2[i] * y2) + c3[i] * y3;                   \
        }                                                                                                                        \
        for (; j < w; ++j) {                                                                                                     \
            const double* cb = P + (size_t)j * ld + w;                                                                           \
            const double yj = ys[j];                                                                                             \
            for (int i = r0; i < r1; ++i) t[i] += cb[i] * yj;                                                                    \
        }                                                                                                                        \
    }                                                                                                                            \
    ATTR static void sn_forward1##SUFFIX(const double* P, int ld, int w, int r, double* ys, double* t) {                        \
        sn_forward_tri##SUFFIX(P, ld, w, ys);                                                                                    \
        sn_forward_rect##SUFFIX(P, ld, w, 0, r, ys, t);                                                                          \
    }                                                                                                                            \
    /* ... and its backward step: ys <- L_d^-T (ys - L_b^T t), again in two pieces.  bt[j] = column j of L_b . t for the columns    \
       [j0, j1) (j0 a multiple of 4): the r-long dot products of four columns share one pass over t; each keeps the four           \
       interleaved partial sums of dot4 (same association, same bits whatever the column range) */                                  \
    ATTR static void sn_backward_rect##SUFFIX(const double* P, int ld, int w, int r, int j0, int j1, const double* t, double* bt) { \
        int j = j0;                                                                                                              \
        for (; j + 4 <= j1; j += 4) {                                                                                            \
            const double *c0 = P + (size_t)j * ld + w, *c1 = c0 + ld, *c2 = c1 + ld, *c3 = c2 + ld;                              \
            double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, a3[4] = {0, 0, 0, 0};                        \
            int i = 0;                                                                                                           \
            for (; i + 4 <= r; i += 4)                                                                                           \
                for (int k = 0; k < 4; ++k) {                                                                                    \
                    const double tv = t[i + k];                                                                                  \
                    a0[k] += c0[i + k] * tv; a1[k] += c1[i + k] * tv; a2[k] += c2[i + k] * tv; a3[k] += c3[i + k] * tv;          \
                }                                                                                                                \
            for (; i < r; ++i) { const double tv = t[i]; a0[0] += c0[i] * tv; a1[0] += c1[i] * tv; a2[0] += c2[i] * tv; a3[0] += c3[i] * tv; } \
            bt[j] = (a0[0] + a0[1]) + (a0[2] + a0[3]); bt[j + 1] = (a1[0] + a1[1]) + (a1[2] + a1[3]);                            \
            bt[j + 2] = (a2[0] + a2[1]) + (a2[2] + a2[3]); bt[j + 3] = (a3[0] + a3[1]) + (a3[2] + a3[3]);                        \
        }                                                                                                                        \
        for (; j < j1; ++j) bt[j] = dot4##SUFFIX(P + (size_t)j * ld + w, t, r);                                                  \
    }                                                                                                                            \
    ATTR static void sn_backward_tri##SUFFIX(const double* P, int ld, int w, double* ys, const double* bt) {                    \
        for (int j = w - 1; j >= 0; --j) {                                                                                       \
            const double* cj = P + (size_t)j * ld;                                                                               \
            ys[j] -= bt[j] + dot4##SUFFIX(cj + j + 1, ys + j + 1, w - 1 - j);                                                    \
        }                                                                                                                        \
    }                                                                                                                            \
    ATTR static void sn_backward1##SUFFIX(const double* P, int ld, int w, int r, double* ys, const double* t) {                 \
        double bt[kMaxWidth];                                                                                                    \
        sn_backward_rect##SUFFIX(P, ld, w, r, 0, w, t, bt);                                                                      \
        sn_backward_tri##SUFFIX(P, ld, w, ys, bt);                                                                               \
    }
    GMG_LDLT_KERNELS(_base, )
    // (this header is host-only code, but engine.hip is also parsed by hipcc's device pass, which knows no x86 features)
#if defined(__HIP_DEVICE_COMPILE__)
    GMG_LDLT_KERNELS(_avx2, )
    GMG_LDLT_KERNELS(_avx512, )
    static bool has_avx2() { return false; }
    static bool has_avx512() { return false; }
#else
    GMG_LDLT_KERNELS(_avx2, __attribute__((target("avx2,fma"))))
    GMG_LDLT_KERNELS(_avx512, __attribute__((target("avx512f,avx512dq,avx2,fma"))))
    static bool has_avx2() {
        static const bool v = [] { __builtin_cpu_init(); return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"); }();
        return v;
    }
    static bool has_avx512() {
        static const bool v = [] { __builtin_cpu_init(); return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }();
        return v;
    }
#endif
#undef GMG_LDLT_KERNELS

    // panel_update as a register-blocked product (AVX2 + FMA): 8 rows x 4 columns of U per pass over the w columns of the descendant's
    // panel, instead of one axpy per (column of U, column of the panel) -- the panel is read k1 / 4 times instead of k1 times and U is
    // written once.  Every entry of U is still ONE accumulator summed over the panel's columns in order with one fused multiply-add
    // each, the arithmetic the axpy version compiles to.  Rows above a column's diagonal inside a 4-column block are computed too
    // (into unused storage of U).
#if !defined(__HIP_DEVICE_COMPILE__)
    __attribute__((target("avx2,fma"))) static void panel_update_blocked(const double* Lk, int ld, int k, int k1, int w, const double* d, double* U, double* F) {
        for (int c = 0; c < w; ++c) for (int j = 0; j < k1; ++j) F[(size_t)c * k1 + j] = Lk[(size_t)c * ld + j] * d[c];
        for (int j0 = 0; j0 < k1; j0 += 4) {
            const int jn = std::min(4, k1 - j0);
            int i0 = j0;
            for (; i0 + 8 <= k; i0 += 8) {
                __m256d a00 = _mm256_setzero_pd(), a01 = a00, a10 = a00, a11 = a00, a20 = a00, a21 = a00, a30 = a00, a31 = a00;
                const double* lp = Lk + i0;
                const double* fp = F + j0;
                if (jn == 4) {
                    for (int c = 0; c < w; ++c, lp += ld, fp += k1) {
                        const __m256d x0 = _mm256_loadu_pd(lp), x1 = _mm256_loadu_pd(lp + 4);
                        const __m256d b0 = _mm256_broadcast_sd(fp), b1 = _mm256_broadcast_sd(fp + 1), b2 = _mm256_broadcast_sd(fp + 2), b3 = _mm256_broadcast_sd(fp + 3);
                        a00 = _mm256_fmadd_pd(x0, b0, a00); a01 = _mm256_fmadd_pd(x1, b0, a01);
                        a10 = _mm256_fmadd_pd(x0, b1, a10); a11 = _mm256_fmadd_pd(x1, b1, a11);
                        a20 = _mm256_fmadd_pd(x0, b2, a20); a21 = _mm256_fmadd_pd(x1, b2, a21);
                        a30 = _mm256_fmadd_pd(x0, b3, a30); a31 = _mm256_fmadd_pd(x1, b3, a31);
                    }
                    double* u = U + (size_t)j0 * k + i0;
                    _mm256_storeu_pd(u, a00); _mm256_storeu_pd(u + 4, a01);
                    _mm256_storeu_pd(u + k, a10); _mm256_storeu_pd(u + k + 4, a11);
                    _mm256_storeu_pd(u + 2 * (size_t)k, a20); _mm256_storeu_pd(u + 2 * (size_t)k + 4, a21);
                    _mm256_storeu_pd(u + 3 * (size_t)k, a30); _mm256_storeu_pd(u + 3 * (size_t)k + 4, a31);
                } else {
                    for (int jj = 0; jj < jn; ++jj) {
                        __m256d s0 = _mm256_setzero_pd(), s1 = s0;
                        const double* l2 = Lk + i0;
                        for (int c = 0; c < w; ++c, l2 += ld) {
                            const __m256d b = _mm256_broadcast_sd(F + (size_t)c * k1 + j0 + jj);
                            s0 = _mm256_fmadd_pd(_mm256_loadu_pd(l2), b, s0); s1 = _mm256_fmadd_pd(_mm256_loadu_pd(l2 + 4), b, s1);
                        }
                        _mm256_storeu_pd(U + (size_t)(j0 + jj) * k + i0, s0); _mm256_storeu_pd(U + (size_t)(j0 + jj) * k + i0 + 4, s1);
                    }
                }
            }
            for (; i0 < k; ++i0)                                   // the last rows, one at a time (same accumulation)
                for (int jj = 0; jj < jn; ++jj) {
                    double acc = 0.0;
                    for (int c = 0; c < w; ++c) acc = __builtin_fma(Lk[(size_t)c * ld + i0], F[(size_t)c * k1 + j0 + jj], acc);
                    U[(size_t)(j0 + jj) * k + i0] = acc;
                }
        }
    }
#else
    static void panel_update_blocked(const double*, int, int, int, int, const double*, double*, double*) {}
#endif

    // forward substitution over the supernodes of one part, ascending.  Rows of the part itself are updated in place; rows above
    // it (the top part, shared with the other half) go to this half's accumulator.
    void forward_part(int part, double* y, double* t, double* acc) const {
        const bool avx = has_avx2();
        for (int s : part_sn_[part]) {
            const int f = sn_first_[s], w = sn_first_[s + 1] - f;
            const int* R = rows_.data() + rows_ptr_[s];
            const int r = rows_ptr_[s + 1] - rows_ptr_[s];
            const double* P = pan_.data() + pan_ptr_[s];
            if (avx) sn_forward1_avx2(P, w + r, w, r, y + f, t); else sn_forward1_base(P, w + r, w, r, y + f, t);
            const int k = own_rows_[s];
            for (int i = 0; i < k; ++i) y[R[i]] -= t[i];
            for (int i = k; i < r; ++i) acc[R[i]] += t[i];
        }
    }
    void backward_part(int part, double* y, double* t) const {
        const bool avx = has_avx2();
        const std::vector<int>& list = part_sn_[part];
        for (size_t q = list.size(); q-- > 0;) {
            const int s = list[q];
            const int f = sn_first_[s], w = sn_first_[s + 1] - f;
            const int* R = rows_.data() + rows_ptr_[s];
            const int r = rows_ptr_[s + 1] - rows_ptr_[s];
            const double* P = pan_.data() + pan_ptr_[s];
            for (int i = 0; i < r; ++i) t[i] = y[R[i]];
            if (avx) sn_backward1_avx2(P, w + r, w, r, y + f, t); else sn_backward1_base(P, w + r, w, r, y + f, t);
        }
    }

    struct PartJob { const SupernodalLDLT* self; double *y, *t; bool forward; const int* groups; };
    double* group_buffer(double* t, int g) const { return t + ((size_t)max_rows_ + 1) * (size_t)g; }
    double* group_acc(double* t, int g) const { return t + ((size_t)max_rows_ + 1) * (size_t)groups_ + (size_t)n * (size_t)g; }
    static void run_part(void* p, int idx) {
        PartJob* j = (PartJob*)p;
        const SupernodalLDLT* S = j->self;
        const int g = j->groups[idx];
        if (!j->forward) { S->backward_part(g, j->y, S->group_buffer(j->t, g)); return; }
        // a chain first takes in what the groups of the earlier stages sent to its columns, in a fixed order (and leaves their
        // accumulators zero)
        for (int c : S->group_cols_[g]) {
            double v = j->y[c];
            for (int f : S->group_feeds_[g]) { double* a = S->group_acc(j->t, f); v -= a[c]; a[c] = 0.0; }
            j->y[c] = v;
        }
        S->forward_part(g, j->y, S->group_buffer(j->t, g), S->group_acc(j->t, g));
    }
    void run_stage(int st, double* y, double* t, bool forward, SpinTeam* team) const {
        const std::vector<int>& gs = stage_groups_[st];
        PartJob job{this, y, t, forward, gs.data()};
        if (team) team->run(run_part, &job, (int)gs.size()); else for (int q = 0; q < (int)gs.size(); ++q) run_part(&job, q);
    }
    // One column.  L z = y runs stage by stage from the leaves of the elimination tree to its root (the parts, then the chains of the
    // top by depth): the groups of a stage are independent -- a column's structure lies on its path to the root -- and what they send
    // to rows above themselves is accumulated per group and subtracted in a fixed order by the chain that owns the row; L^T x = z the
    // other way round.  The arithmetic does not depend on whether / how many threads of `team` share the groups with this one.
    // t: scratch_doubles() doubles, accumulators zero on entry and on exit.
    // (Sharing the rows / columns of single panels with the team was built and measured on the GPU box's EPYC 9575F: 41 + 59 us of
    // the then sequential top became 107 + 113 us -- a hand-over per panel costs more than the 3 us a panel takes.)
    void solve_column(const double* b, double* x, double* y, double* t, SpinTeam* team) const {
        const int S = (int)stage_groups_.size();
        for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
        for (int st = 0; st < S; ++st) run_stage(st, y, t, true, team);
        for (int j = 0; j < n; ++j) y[j] /= D_[j];
        for (int st = S - 1; st >= 0; --st) run_stage(st, y, t, false, team);
        for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
    }

    // Numeric factorisation, left-looking by supernodes: assemble the panel from A, subtract the updates of the descendants that reach
    // it -- in ASCENDING order of the descendants (upd_*: fixed by the symbolic phase, so the sums do not depend on who computes them) --,
    // factor the panel.  Threaded along the split of the elimination tree the back-substitution uses (plan_split): the parts are sets of
    // whole subtrees, independent of each other -- one job each --; the supernodes of the top (separators: every one of them is updated by
    // descendants from all over the tree, half of the panel entries and most of the arithmetic at n = 6 005) follow one after the other,
    // each with its update phase in two parallel steps: the dense products of its descendants side by side, then their subtraction from
    // the panel by column ranges (a column receives its updates in the same order whichever range it falls into, and every entry of a
    // product is one accumulator over the descendant's columns: the factor is bitwise the same for any number of threads --
    // tests/test_host.py).  The demos' call (a new tau per frame, 36 k vertices, coarsest level of 6 005 unknowns) spent most of its
    // set-up here: 7.5 ms on one thread, 3.0 ms on eight.
    struct NumericWork { std::vector<int> relpos; std::vector<double> U, F; };
    // updates of supernode s restricted to its columns [c0, c1) (relative), descendant after descendant (the parts: the whole supernode at once)
    void apply_updates(int s, int c0, int c1, const std::vector<int>& relpos, NumericWork& wk, bool avx) {
        const int f = sn_first_[s], w = sn_first_[s + 1] - f;
        const int ld = w + (rows_ptr_[s + 1] - rows_ptr_[s]);
        double* P = pan_.data() + pan_ptr_[s];
        const int col_lo = f + c0, col_hi = f + c1;
        for (int q = upd_ptr_[s]; q < upd_ptr_[s + 1]; ++q) {
            const int dd = upd_sn_[q];
            const int fd = sn_first_[dd], wd = sn_first_[dd + 1] - fd;
            const int* Rd = rows_.data() + rows_ptr_[dd];
            const int rd = rows_ptr_[dd + 1] - rows_ptr_[dd];
            const int ldd = wd + rd;
            int p0 = upd_pos_[q];
            while (p0 < rd && Rd[p0] < col_lo) ++p0;                    // (rows inside this supernode's columns: at most w of them)
            int k1 = 0;
            while (p0 + k1 < rd && Rd[p0 + k1] < col_hi) ++k1;
            if (k1 == 0) continue;
            const int k = rd - p0;
            const double* Lk = pan_.data() + pan_ptr_[dd] + wd + p0;
            wk.U.resize((size_t)k * k1);
            if (avx) { wk.F.resize((size_t)wd * k1); panel_update_blocked(Lk, ldd, k, k1, wd, D_.data() + fd, wk.U.data(), wk.F.data()); }
            else panel_update_base(Lk, ldd, k, k1, wd, D_.data() + fd, wk.U.data());
            for (int j = 0; j < k1; ++j) {
                double* tc = P + (size_t)(Rd[p0 + j] - f) * ld;
                const double* uj = wk.U.data() + (size_t)j * k;
                for (int i = j; i < k; ++i) tc[relpos[Rd[p0 + i]]] -= uj[i];
            }
        }
    }
    void set_relpos(int s, std::vector<int>& relpos) const {
        const int f = sn_first_[s], w = sn_first_[s + 1] - f;
        const int* R = rows_.data() + rows_ptr_[s];
        const int r = rows_ptr_[s + 1] - rows_ptr_[s];
        for (int j = 0; j < w; ++j) relpos[f + j] = j;
        for (int i = 0; i < r; ++i) relpos[R[i]] = w + i;
    }
    bool factor_panel(int s, bool avx) {
        const int f = sn_first_[s], w = sn_first_[s + 1] - f;
        const int ld = w + (rows_ptr_[s + 1] - rows_ptr_[s]);
        double* P = pan_.data() + pan_ptr_[s];
        return avx ? panel_factor_avx2(P, ld, ld, w, D_.data() + f) : panel_factor_base(P, ld, ld, w, D_.data() + f);
    }
    void numeric(const Compressed& A) {
        if (pan_.fresh) pan_.fresh = false;       // first factorisation into this storage: zero already, pages not yet touched
        else {
            const size_t chunk = (size_t)1 << 16, nch = (pan_.size() + chunk - 1) / chunk;
            parallel_ranges((int)nch, hw_threads(), [&](int lo, int hi, int) {
                const size_t a = (size_t)lo * chunk, e = std::min(pan_.size(), (size_t)hi * chunk);
                if (a < e) std::memset(pan_.data() + a, 0, (e - a) * sizeof(double));
            }, 8);
        }
        const bool avx = has_avx2();
        const int T = numeric_threads();
        std::atomic<int> failed{0};
        const bool trace = EnvSwitches::get().trace_ldlt;
        auto tr0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) { if (!trace) return; auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "[gmg ldlt] numeric %-8s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tr0).count()); tr0 = now; };
        lap("zero");
        // a team of spinning threads for the few milliseconds this takes (hand-overs of ~0.2 us: an update phase of a top supernode is
        // ~0.2 ms of work in all, the worker pool's condition variable would eat it); created on first use, asleep when not armed
        if (T > 1 && (!fteam_ || fteam_->helpers() != T - 1)) fteam_ = std::make_shared<SpinTeam>(T - 1);
        SpinTeam* team = T > 1 ? fteam_.get() : nullptr;
        struct Armed { SpinTeam* t; explicit Armed(SpinTeam* q) : t(q) { if (t) t->arm(); } ~Armed() { if (t) t->disarm(); } } armed(team);
        auto run_jobs = [&](void (*fn)(void*, int), void* arg, int njobs) { if (team) team->run(fn, arg, njobs); else for (int j = 0; j < njobs; ++j) fn(arg, j); };
        // ---- the parts: whole subtrees, one job per part
        struct PartsCtx { SupernodalLDLT* self; const Compressed* A; std::atomic<int>* failed; bool avx; } pc{this, &A, &failed, avx};
        run_jobs([](void* p, int part) {
            PartsCtx& c = *(PartsCtx*)p;
            SupernodalLDLT& L = *c.self;
            try {                                               // (a job may run on a team thread: nothing may leave it -- the work buffers allocate)
                NumericWork wk;
                wk.relpos.assign((size_t)L.n, -1);
                for (int s : L.part_sn_[(size_t)part]) {
                    if (c.failed->load(std::memory_order_relaxed)) return;
                    const int f = L.sn_first_[s], w = L.sn_first_[s + 1] - f;
                    L.set_relpos(s, wk.relpos);
                    L.s_assemble(*c.A, s, f, w, L.pan_.data() + L.pan_ptr_[s], w + (L.rows_ptr_[s + 1] - L.rows_ptr_[s]), wk.relpos);
                    L.apply_updates(s, 0, w, wk.relpos, wk, c.avx);
                    if (!L.factor_panel(s, c.avx)) { c.failed->store(1, std::memory_order_relaxed); return; }
                }
            } catch (...) { c.failed->store(2, std::memory_order_relaxed); }
        }, &pc, parts_);
        if (failed.load() == 2) throw std::bad_alloc();      // (on the calling thread: becomes the C-ABI's error code)
        if (failed.load()) return;
        lap("parts");
        // ---- the top: one supernode after the other.  Its updates in two parallel steps: the dense products U = L_k D L_k1^T of all its
        // descendants side by side (one job each, into one buffer), then their subtraction from the panel by column ranges, every range
        // walking the descendants in ascending order
        double t_fac = 0, t_prod = 0, t_sub = 0, w_sum = 0, r_sum = 0, u_sum = 0; int n_top = 0;
        std::vector<int> relpos((size_t)n, -1);
        struct Upd { int dd, p0, k, k1; size_t u, f; };
        std::vector<Upd> ups;
        std::vector<double> buf;
        struct TopCtx { SupernodalLDLT* self; int s, w, slices; const std::vector<int>* relpos; std::vector<Upd>* ups; double* buf; bool avx; } tc{this, 0, 0, 1, &relpos, &ups, nullptr, avx};
        for (int s : top_sn_) {
            const int f = sn_first_[s], w = sn_first_[s + 1] - f;
            set_relpos(s, relpos);
            s_assemble(A, s, f, w, pan_.data() + pan_ptr_[s], w + (rows_ptr_[s + 1] - rows_ptr_[s]), relpos);
            const int nup = upd_ptr_[s + 1] - upd_ptr_[s];
            ups.clear();
            size_t need = 0;
            for (int q = upd_ptr_[s]; q < upd_ptr_[s + 1]; ++q) {
                const int dd = upd_sn_[q], p0 = upd_pos_[q];
                const int* Rd = rows_.data() + rows_ptr_[dd];
                const int rd = rows_ptr_[dd + 1] - rows_ptr_[dd], wd = sn_first_[dd + 1] - sn_first_[dd];
                int k1 = 0;
                while (p0 + k1 < rd && Rd[p0 + k1] < f + w) ++k1;
                const int k = rd - p0;
                ups.push_back(Upd{dd, p0, k, k1, need, need + (size_t)k * k1});
                need += (size_t)k * k1 + (size_t)wd * k1;
            }
            if (buf.size() < need) buf.resize(need);
            tc.s = s; tc.w = w; tc.buf = buf.data();
            auto tq = std::chrono::steady_clock::now();
            run_jobs([](void* p, int q) {                       // products
                TopCtx& c = *(TopCtx*)p;
                SupernodalLDLT& L = *c.self;
                const Upd& u = (*c.ups)[(size_t)q];
                const int fd = L.sn_first_[u.dd], wd = L.sn_first_[u.dd + 1] - fd;
                const int ldd = wd + (L.rows_ptr_[u.dd + 1] - L.rows_ptr_[u.dd]);
                const double* Lk = L.pan_.data() + L.pan_ptr_[u.dd] + wd + u.p0;
                if (c.avx) panel_update_blocked(Lk, ldd, u.k, u.k1, wd, L.D_.data() + fd, c.buf + u.u, c.buf + u.f);
                else panel_update_base(Lk, ldd, u.k, u.k1, wd, L.D_.data() + fd, c.buf + u.u);
            }, &tc, nup);
            if (trace) { auto now = std::chrono::steady_clock::now(); t_prod += std::chrono::duration<double, std::milli>(now - tq).count(); tq = now; }
            const int slices = std::max(1, std::min(T, w / 4));
            tc.slices = slices;
            run_jobs([](void* p, int sl) {                      // subtraction, columns [c0, c1) of the panel
                TopCtx& c = *(TopCtx*)p;
                SupernodalLDLT& L = *c.self;
                const int f = L.sn_first_[c.s];
                const int ld = c.w + (L.rows_ptr_[c.s + 1] - L.rows_ptr_[c.s]);
                double* P = L.pan_.data() + L.pan_ptr_[c.s];
                const int col_lo = f + (int)((long)c.w * sl / c.slices), col_hi = f + (int)((long)c.w * (sl + 1) / c.slices);
                const int* relpos = c.relpos->data();
                for (const Upd& u : *c.ups) {
                    const int* Rd = L.rows_.data() + L.rows_ptr_[u.dd] + u.p0;
                    for (int j = 0; j < u.k1; ++j) {
                        if (Rd[j] < col_lo) continue;
                        if (Rd[j] >= col_hi) break;
                        double* tcol = P + (size_t)(Rd[j] - f) * ld;
                        const double* uj = c.buf + u.u + (size_t)j * u.k;
                        for (int i = j; i < u.k; ++i) tcol[relpos[Rd[i]]] -= uj[i];
                    }
                }
            }, &tc, slices);
            if (trace) t_sub += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tq).count();
            auto tf = std::chrono::steady_clock::now();
            if (!factor_panel(s, avx)) return;      // (its rows shared between the team in ranges: slower -- short inner loops, shared cache lines)
            if (trace) { t_fac += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count(); ++n_top; w_sum += w; r_sum += rows_ptr_[s + 1] - rows_ptr_[s]; u_sum += nup; }
        }
        if (trace) std::fprintf(stderr, "[gmg ldlt] top: %d supernodes, mean width %.1f, mean rows below %.1f, mean updates %.1f, products %.2f ms, subtraction %.2f ms, panel factor %.2f ms\n", n_top, w_sum / std::max(n_top, 1), r_sum / std::max(n_top, 1), u_sum / std::max(n_top, 1), t_prod, t_sub, t_fac);
        lap("top");
        ok = true;
    }

    // panel of supernode s <- the entries of P A P^T in its columns (lower triangle)
    void s_assemble(const Compressed& A, int s, int f, int w, double* P, int ld, const std::vector<int>& relpos) const {
        (void)s;
        // lower column c of the permuted matrix = {(i, c) : i >= c}; by symmetry these are the upper entries (c, i), which
        // sit in the upper COLUMN i.  Walk A's column perm[c] directly instead: its entries (perm^-1(row), c) with row' >= c.
        for (int j = 0; j < w; ++j) {
            const int c = f + j, old = perm[c];
            double* pc = P + (size_t)j * ld;
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) {
                const int i = inv_[A.idx[p]];
                if (i >= c) pc[relpos[i]] += A.val[p];
            }
        }
    }

};

}  // namespace gmg
