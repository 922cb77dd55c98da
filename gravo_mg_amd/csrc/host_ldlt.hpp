// host_ldlt.hpp -- sparse LDL^T for the coarsest level, kept on the host as the north star asks.
//
// Replaces the reference's `Eigen::SimplicialLDLT<Eigen::SparseMatrix<double>> coarsestSolver`
// (gravomg/include/gravomg/multigrid_solver.h:145; factor at gravomg/src/multigrid_solver.cpp:1401,
// back-substitution once per V-cycle at :1075).  Eigen is a third-party dependency that is not
// present; this is an independent implementation of the same mathematical object: a fill-reducing
// symmetric permutation (minimum degree on the explicit elimination graph -- the coarsest level has
// 1 000 ... ~8 000 unknowns) followed by an up-looking sparse LDL^T driven by the
// elimination tree.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iterator>
#include <vector>

#include "host_sparse.hpp"

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

    long factor_nnz() const { return (long)Lp.empty() ? 0 : Lp[n]; }

    static constexpr int kLeaf = 320;             // nested dissection stops at regions of this size
    static constexpr int kMinDegreeMax = 2500;    // up to here the exact minimum-degree ordering (bit-row elimination graph) gives ~6 % less fill

private:
    // Nested dissection with breadth-first level-set separators: O(E log n), fill O(n log n) on mesh-like graphs.
    // A region is split by the middle level of a BFS started at a pseudo-peripheral vertex (edges of a BFS only join
    // equal or adjacent levels, so a whole level separates); the two halves are ordered first, the separator last.
    void nested_dissection(const Compressed& A) {
        perm.clear(); perm.reserve(n);
        std::vector<int> region(n, 0), level(n, -1), queue;
        int next_region = 1;
        struct Task { std::vector<int> nodes; int id; };
        // explicit stack; a task's separator is emitted AFTER its two halves, so push a marker task holding it
        struct Item { std::vector<int> nodes; int id; bool emit; };
        std::vector<Item> stack;
        { std::vector<int> all(n); for (int i = 0; i < n; ++i) all[i] = i; stack.push_back({std::move(all), 0, false}); }
        std::vector<int> out_rev;      // built in reverse (separators first), reversed at the end
        out_rev.reserve(n);
        while (!stack.empty()) {
            Item it = std::move(stack.back());
            stack.pop_back();
            if (it.emit) { for (auto r = it.nodes.rbegin(); r != it.nodes.rend(); ++r) out_rev.push_back(*r); continue; }
            if ((int)it.nodes.size() <= kLeaf) {      // leaf region: exact minimum degree on its own subgraph
                std::vector<int> ord = leaf_min_degree(A, it.nodes, region, it.id);
                for (auto r = ord.rbegin(); r != ord.rend(); ++r) out_rev.push_back(*r);
                continue;
            }
            const int id = it.id;
            auto bfs = [&](int start) {      // levels inside region `id`; returns the visit order in `queue`
                queue.clear(); queue.push_back(start); level[start] = 0;
                for (size_t h = 0; h < queue.size(); ++h) {
                    int v = queue[h];
                    for (int p = A.ptr[v]; p < A.ptr[v + 1]; ++p) { int w = A.idx[p]; if (region[w] == id && level[w] < 0) { level[w] = level[v] + 1; queue.push_back(w); } }
                }
            };
            for (int v : it.nodes) level[v] = -1;
            bfs(it.nodes[0]);
            if (queue.size() < it.nodes.size()) {
                // disconnected region: peel this component off and handle both parts independently
                std::vector<int> comp(queue), rest;
                const int idc = next_region++, idr = next_region++;
                for (int v : comp) region[v] = idc;
                for (int v : it.nodes) if (level[v] < 0) { rest.push_back(v); region[v] = idr; }
                stack.push_back({std::move(rest), idr, false});
                stack.push_back({std::move(comp), idc, false});
                continue;
            }
            const int far = queue.back();
            for (int v : it.nodes) level[v] = -1;
            bfs(far);
            const int depth = level[queue.back()];
            if (depth < 2) { for (auto r = it.nodes.rbegin(); r != it.nodes.rend(); ++r) out_rev.push_back(*r); continue; }   // clique-like: no separator
            // middle level by vertex count
            std::vector<int> cnt(depth + 1, 0);
            for (int v : queue) cnt[level[v]]++;
            int mid = 1, acc = cnt[0];
            while (mid < depth - 1 && acc + cnt[mid] < (int)queue.size() / 2) { acc += cnt[mid]; ++mid; }
            std::vector<int> a, b, sep;
            const int ida = next_region++, idb = next_region++;
            for (int v : queue) {
                if (level[v] < mid) { a.push_back(v); region[v] = ida; }
                else if (level[v] > mid) { b.push_back(v); region[v] = idb; }
                else { sep.push_back(v); region[v] = -1; }
            }
            // processing order (stack, reversed output): separator is emitted first into out_rev => eliminated last
            stack.push_back({std::move(a), ida, false});
            stack.push_back({std::move(b), idb, false});
            stack.push_back({std::move(sep), -1, true});
        }
        perm.assign(out_rev.rbegin(), out_rev.rend());
    }

    // Exact minimum degree restricted to the vertices of one region (ids local to `nodes`).
    static std::vector<int> leaf_min_degree(const Compressed& A, const std::vector<int>& nodes, const std::vector<int>& region, int id) {
        const int m = (int)nodes.size();
        std::vector<int> local(m);
        std::vector<std::pair<int, int>> key(m);
        for (int i = 0; i < m; ++i) key[i] = {nodes[i], i};
        std::sort(key.begin(), key.end());
        auto find = [&](int g) { auto it = std::lower_bound(key.begin(), key.end(), std::make_pair(g, -1)); return it->second; };
        std::vector<std::vector<int>> adj(m);
        for (int i = 0; i < m; ++i) {
            const int g = nodes[i];
            for (int p = A.ptr[g]; p < A.ptr[g + 1]; ++p) { int w = A.idx[p]; if (w != g && region[w] == id) adj[i].push_back(find(w)); }
            std::sort(adj[i].begin(), adj[i].end());
            adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
        }
        std::vector<char> done(m, 0);
        std::vector<int> out, merged;
        out.reserve(m);
        for (int step = 0; step < m; ++step) {
            int v = -1; size_t best = ~(size_t)0;
            for (int j = 0; j < m; ++j) if (!done[j] && adj[j].size() < best) { best = adj[j].size(); v = j; }
            done[v] = 1;
            out.push_back(nodes[v]);
            const std::vector<int>& nv = adj[v];
            for (int u : nv) {
                merged.clear();
                std::set_union(adj[u].begin(), adj[u].end(), nv.begin(), nv.end(), std::back_inserter(merged));
                std::vector<int>& au = adj[u];
                au.clear();
                for (int w : merged) if (w != u && w != v) au.push_back(w);
            }
            adj[v].clear();
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

}  // namespace gmg
