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
#include <cstring>
#include <thread>
#include <vector>

namespace gmg {

struct Compressed {
    int n_outer = 0;              // number of compressed vectors (columns for CSC, rows for CSR)
    int n_inner = 0;              // length of each compressed vector
    std::vector<int> ptr;         // n_outer + 1
    std::vector<int> idx;         // nnz, ascending inside each outer vector
    std::vector<double> val;      // nnz
    int nnz() const { return ptr.empty() ? 0 : ptr[n_outer]; }
    void assign(int nouter, int ninner, const int* p, const int* i, const double* v) {
        n_outer = nouter; n_inner = ninner;
        ptr.assign(p, p + nouter + 1);
        idx.assign(i, i + ptr[nouter]);
        val.assign(v, v + ptr[nouter]);
    }
};

inline int hw_threads() {
    unsigned t = std::thread::hardware_concurrency();
    if (t == 0) t = 1;
    return (int)std::min(t, 128u);
}

template <class F>
inline void parallel_ranges(int n, int nthreads, F&& f) {
    if (nthreads <= 1 || n < 4096) { f(0, n, 0); return; }
    std::vector<std::thread> pool;
    int chunk = (n + nthreads - 1) / nthreads;
    for (int t = 0; t < nthreads; ++t) {
        int lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        pool.emplace_back([&f, lo, hi, t] { f(lo, hi, t); });
    }
    for (auto& th : pool) th.join();
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
