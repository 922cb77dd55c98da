// host_plan.hpp -- turns the host hierarchy (A_k, U_k in compressed storage) into the device layout.
//
// Device layout (DESIGN.md "Data layout in HBM"):
//   * every level gets its own DEVICE NUMBERING: rows grouped by colour (multicolour Gauss-Seidel),
//     inside a colour kept in mesh order except for a stable sort by row length inside windows of
//     `sigma` rows (keeps SELL padding small without destroying locality), each colour class padded
//     to a multiple of `row_align` rows (64 = one wavefront; 64*P when P ranks split every colour);
//   * matrices are SELL-64: slices of 64 rows, one row per lane, entries stored j-major inside a
//     slice so that a wavefront's loads of col[]/val[] are contiguous (256 B / 512 B per j);
//   * A_k is stored WITHOUT its diagonal; the diagonal is a dense vector (the reference fetches it with
//     coeffRef(k,k), gravomg/src/multigrid_solver.cpp:1207);
//   * U_k (<= 3 entries per row, multigrid_solver.cpp:371-373,413-414) is stored twice: SELL of U for
//     x += U e, and SELL of U^T (rows sorted by length inside windows, output row in `row_of`) for
//     rc = U^T r.  U^T as CSR is exactly the reference's CSC storage of U.
// Everything here runs once per system on the host; the passes over rows are threaded (parallel_ranges) and the
// large arrays are left uninitialised until their single parallel fill.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <future>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include <stdlib.h>
#include <sys/mman.h>

#include "host_sparse.hpp"

namespace gmg {

constexpr int kSlice = 64;

// Plain array without value-initialisation (std::vector zero-fills: ~100 ms per 200 MB on one core).  Large arrays are
// 2 MiB-aligned and marked MADV_HUGEPAGE: the staging buffers of one 3 M-vertex level are ~0.5 GB of fresh memory, and
// with 4 KiB pages 128 threads first-touching them serialise on the page-fault path.
template <class T>
struct RawVec {
    T* p = nullptr;
    size_t n = 0;
    RawVec() {}
    RawVec(const RawVec&) = delete;
    RawVec& operator=(const RawVec&) = delete;
    RawVec(RawVec&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    RawVec& operator=(RawVec&& o) noexcept { if (this != &o) { std::free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~RawVec() { std::free(p); }
    void resize(size_t m) {
        std::free(p);
        p = nullptr; n = m;
        const size_t bytes = std::max<size_t>(m, 1) * sizeof(T);
        if (bytes >= (size_t)4 << 20) {
            void* q = nullptr;
            const size_t rounded = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            if (posix_memalign(&q, (size_t)2 << 20, rounded) == 0) {
                (void)madvise(q, rounded, MADV_HUGEPAGE);
                p = (T*)q;
                return;
            }
        }
        p = (T*)std::malloc(bytes);
    }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

struct LevelOrdering {
    int n = 0;                        // real unknowns
    int n_pad = 0;                    // device vector length (multiple of 64)
    int n_colors = 0;
    std::vector<int> color_begin;     // n_colors + 1, in device rows (multiples of row_align)
    std::vector<int> new2old;         // n_pad, -1 for padding rows
    std::vector<int> old2new;         // n
    // "blocked" levels (block-hybrid Gauss-Seidel, one launch per sweep): rows grouped into compact
    // blocks of <= block_rows rows, colour-sorted inside a block, each block padded to 64 rows.
    bool reordered = false;           // colour-major level whose rows follow a BFS patch order instead of the input order
    bool blocked = false;
    std::vector<int> blk_begin;       // n_blocks + 1, device rows (multiples of 64)
    std::vector<int> blk_ncolors;     // n_blocks
    std::vector<unsigned char> row_color;   // n_pad, colour of a row inside its block (padding rows: 0)
    int n_blocks() const { return blk_begin.empty() ? 0 : (int)blk_begin.size() - 1; }
};

struct SellHost {
    int n_rows_pad = 0;               // rows covered by slices (multiple of 64)
    int n_cols = 0;
    int n_slices = 0;
    int lpr = 1;                      // lanes per row: 1 (64 rows per slice) or 4 (16 rows per slice)
    int64_t nnz_real = 0;
    std::vector<int64_t> slice_ptr;   // n_slices + 1, element offsets (multiples of 64)
    RawVec<int> col;
    RawVec<double> val;
    std::vector<int> row_of;          // empty => slice row r is device row r; else output row (or -1)
    int64_t stored() const { return slice_ptr.empty() ? 0 : slice_ptr[n_slices]; }
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Greedy first-fit colouring in natural order (4 colours on a valence-6 mesh,
// 13-18 on Galerkin levels).  A is symmetric; self-loops ignored.
// Sparsity pattern by reference (the caller's arrays): what the colour-major ordering needs of a matrix.
struct PatternView {
    int n_outer;
    const int* ptr;
    const int* idx;
};

// Greedy first-fit colouring; vertices visited in `order` (natural order if empty).  Inherently sequential (a vertex
// sees the colours of the neighbours visited before it) and the longest host task of a set-up, so the loop works on
// one byte per vertex (3 MB instead of 12 at 3 M vertices: stays in cache) and finds the first free colour in a 64-bit
// mask; colours >= 64 (or > 254 colours in all) take the general path.  The result is the same either way.
template <class Mat>
inline int greedy_coloring_general(const Mat& A, std::vector<int>& color, const std::vector<int>& order) {
    const int n = A.n_outer;
    color.assign(n, -1);
    std::vector<int> forbid;
    int ncol = 0;
    for (int t = 0; t < n; ++t) {
        const int i = order.empty() ? t : order[t];
        forbid.assign((size_t)ncol + 1, 0);
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
            int j = A.idx[p];
            if (j != i && color[j] >= 0) forbid[color[j]] = 1;
        }
        int c = 0;
        while (c < ncol && forbid[c]) ++c;
        color[i] = c;
        if (c == ncol) ++ncol;
    }
    return ncol;
}

// One byte per vertex; returns -1 (c8 unusable) when more than 254 colours would be needed.
// idx_sorted: the indices of a row are ascending (canonical CSC / CSR).  Visiting in natural order, every neighbour with a larger index is
// still uncoloured and the walk over a row can stop at the first one: half the entries of a symmetric pattern, the same colours (round 4:
// 18 -> ~10 ms at 3 M vertices -- the sequential colouring is what a cold gmg_set_system otherwise waits for).
template <class Mat>
inline int greedy_coloring_bytes(const Mat& A, RawVec<unsigned char>& c8, const std::vector<int>& order, bool idx_sorted = false) {
    const int n = A.n_outer;
    constexpr unsigned char kNone = 255;
    c8.resize((size_t)std::max(n, 1));
    std::memset(c8.data(), kNone, (size_t)n);
    unsigned char* cc = c8.data();
    int ncol = 0;
    std::vector<unsigned char> forbid;
    for (int t = 0; t < n; ++t) {
        const int i = order.empty() ? t : order[t];
        uint64_t mask = 0;
        if (idx_sorted && order.empty()) {
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
                const int j = A.idx[p];
                if (j >= i) break;                         // (ascending indices: the rest of the row is not coloured yet)
                const unsigned c = cc[j];
                if (c < 64) mask |= (uint64_t)1 << c;
            }
        } else {
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
                const unsigned c = cc[A.idx[p]];           // the vertex itself is still kNone here
                if (c < 64) mask |= (uint64_t)1 << c;
            }
        }
        int c;
        if (~mask != 0) c = __builtin_ctzll(~mask);        // first free colour below 64 (== ncol when 0..ncol-1 are all taken)
        else {                                              // 0..63 all taken: first free colour >= 64
            forbid.assign((size_t)ncol + 1, 0);
            for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) { const unsigned cj = cc[A.idx[p]]; if (cj != kNone) forbid[cj] = 1; }
            c = 64;
            while (c < ncol && forbid[c]) ++c;
        }
        if (c >= 254) return -1;
        cc[i] = (unsigned char)c;
        if (c >= ncol) ncol = c + 1;
    }
    return ncol;
}

// The same colouring (natural order, ascending rows) started BEFORE the caller's arrays have been inspected -- a cold gmg_set_system waits for this loop
// longer than for anything else, and the inspection takes 6 of its first milliseconds at 3 M vertices (engine.hip: "colouring ahead of the verdict").
// Every pointer and index is checked against [0, nnz] / [0, n), so arrays that will fail the inspection cannot take the loop out of bounds; a result
// computed from rows that turn out not to be ascending is thrown away by the caller.  Returns the number of colours, -1 (64 colours are not enough:
// the general loop decides) or -2 (inconsistent arrays, or `stop` raised: no result).
inline int greedy_coloring_ahead(int n, const int* ptr, const int* idx, int64_t nnz, RawVec<unsigned char>& c8, const std::atomic<int>& stop) {
    constexpr unsigned char kNone = 255;
    c8.resize((size_t)std::max(n, 1));
    std::memset(c8.data(), kNone, (size_t)n);
    unsigned char* cc = c8.data();
    int ncol = 0;
    for (int i = 0; i < n; ++i) {
        if ((i & 8191) == 0 && stop.load(std::memory_order_relaxed)) return -2;
        const int64_t p0 = ptr[i], p1 = ptr[i + 1];
        if (p0 < 0 || p1 < p0 || p1 > nnz) return -2;
        uint64_t mask = 0;
        for (int64_t p = p0; p < p1; ++p) {
            const int j = idx[p];
            if ((unsigned)j >= (unsigned)n) return -2;
            if (j >= i) break;                             // (ascending indices: the rest of the row is not coloured yet)
            const unsigned c = cc[j];
            if (c < 64) mask |= (uint64_t)1 << c;
        }
        if (~mask == 0) return -1;
        const int c = __builtin_ctzll(~mask);
        cc[i] = (unsigned char)c;
        if (c >= ncol) ncol = c + 1;
    }
    return ncol;
}
// a colouring made ahead of make_ordering (natural visit order): the bytes and the number of colours
struct PreColoring { RawVec<unsigned char> c8; int n_colors = -1; };

// (Round 6 built the same colouring as a dataflow over strips of the visit sequence on several threads -- a vertex waits for the bytes of the
// neighbours visited before it -- and removed it again: on a mesh in a local order the colouring is a chain along every mesh row, each row two
// columns behind the one above, so threads either hold consecutive strips of ONE chain (cyclic deal: no parallelism) or read colour bytes a peer
// wrote nanoseconds ago (a coherence miss per vertex: 40 -> 280 ms on eight threads at 3 M vertices).  docs/rounds/round6.md.)
template <class Mat>
inline int greedy_coloring(const Mat& A, std::vector<int>& color, const std::vector<int>& order = std::vector<int>()) {
    RawVec<unsigned char> c8;
    const int ncol = greedy_coloring_bytes(A, c8, order);
    if (ncol < 0) return greedy_coloring_general(A, color, order);
    const int n = A.n_outer;
    color.resize((size_t)n);
    parallel_ranges(n, hw_threads(), [&](int lo, int hi, int) { for (int i = lo; i < hi; ++i) color[i] = c8[i]; }, 1 << 16);
    return ncol;
}

// Breadth-first growth of compact patches of <= block_rows vertices over the matrix graph: seeds are taken on the
// frontier of what is already assigned, so consecutive patches are neighbours.  members = visit order (all patches
// back to back), mem_begin = patch boundaries.
template <class Mat>
inline void grow_patches(const Mat& A, int block_rows, std::vector<int>& block_of, std::vector<int>& members, std::vector<int>& mem_begin) {
    const int n = A.n_outer;
    block_of.assign(n, -1);
    members.clear(); members.reserve(n);
    mem_begin.assign(1, 0);
    std::vector<int> cand;
    cand.reserve((size_t)n / 2 + 16);
    size_t cand_head = 0;
    int scan = 0;
    while (true) {
        int seed = -1;
        while (cand_head < cand.size()) {
            int c = cand[cand_head++];
            if (block_of[c] < 0) { seed = c; break; }
        }
        if (seed < 0) {
            while (scan < n && block_of[scan] >= 0) ++scan;
            if (scan >= n) break;
            seed = scan;
        }
        const int b = (int)mem_begin.size() - 1;
        const size_t first = members.size();
        block_of[seed] = b; members.push_back(seed);
        size_t head = first;
        while (head < members.size() && (int)(members.size() - first) < block_rows) {
            int v = members[head++];
            for (int p = A.ptr[v]; p < A.ptr[v + 1] && (int)(members.size() - first) < block_rows; ++p) {
                int w = A.idx[p];
                if (block_of[w] < 0) { block_of[w] = b; members.push_back(w); }
            }
        }
        // one unassigned neighbour of each unexpanded tail vertex becomes a seed candidate for the next patches
        for (size_t q = head; q < members.size(); ++q) {
            int v = members[q];
            for (int p = A.ptr[v]; p < A.ptr[v + 1]; ++p)
                if (block_of[A.idx[p]] < 0) { cand.push_back(A.idx[p]); break; }
        }
        mem_begin.push_back((int)members.size());
    }
}

// Mean |row - column| over the stored entries (sampled): a cheap measure of how far a row's x-gathers reach.
template <class Mat>
inline double mean_index_distance(const Mat& A) {
    const int n = A.n_outer;
    const int step = std::max(1, n / 65536);
    double sum = 0.0; long cnt = 0;
    for (int i = 0; i < n; i += step)
        for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) { sum += std::abs(A.idx[p] - i); ++cnt; }
    return cnt ? sum / cnt : 0.0;
}

// multicolor = false -> a single "colour" holding every row (Jacobi-type smoothers, coarsest level).
// reorder: 0 never, 1 always, 2 automatic -- when the input vertex order has poor locality (a randomly ordered scan
// or point cloud: every gather of x misses the caches, measured 3x per V-cycle at 3 M vertices) the rows inside a
// colour follow a breadth-first patch order instead of the input order.
// ext_base: a locality-preserving visit order supplied by the caller (the hierarchy's cluster order, see
// cluster_order below); used instead of growing patches when a reordering is due.
template <class Mat>
inline bool wants_locality_reorder(const Mat& A, int reorder) {
    const int n = A.n_outer;
    return reorder == 1 || (reorder == 2 && n > 65536 && mean_index_distance(A) > std::max(32768.0, n / 32.0));
}

// The level-0 point graph of a hierarchy (`neigh`: n x K neighbour table, -1 = no neighbour) as a canonical pattern: row i = its distinct
// neighbours and i itself, ascending -- what `tau M + S` / `M + tau S` of the mesh the table came from looks like (the table IS the
// off-diagonal pattern of S, gravomg_bindings/src/gravomg/util.py:36-44).  Threaded; ptr: n + 1, idx: entries.
inline void neigh_pattern(const int* neigh, int n, int K, RawVec<int>& ptr, RawVec<int>& idx) {
    ptr.resize((size_t)n + 1);
    const int T = std::max(1, std::min(hw_threads(), 16));
    auto row_of = [&](int i, int* out) {          // sorted distinct neighbours + self; returns the count (out holds K + 1 ints)
        int m = 0;
        const int* r = neigh + (size_t)i * K;
        for (int k = 0; k < K; ++k) if (r[k] >= 0 && r[k] != i) out[m++] = r[k];
        out[m++] = i;
        std::sort(out, out + m);
        return (int)(std::unique(out, out + m) - out);
    };
    parallel_ranges(n, T, [&](int lo, int hi, int) { std::vector<int> tmp((size_t)K + 1); for (int i = lo; i < hi; ++i) ptr[(size_t)i + 1] = row_of(i, tmp.data()); }, 1 << 14);
    ptr[0] = 0;
    for (int i = 0; i < n; ++i) ptr[(size_t)i + 1] += ptr[i];
    idx.resize((size_t)ptr[n]);
    parallel_ranges(n, T, [&](int lo, int hi, int) { std::vector<int> tmp((size_t)K + 1); for (int i = lo; i < hi; ++i) { const int m = row_of(i, tmp.data()); std::memcpy(idx.data() + ptr[i], tmp.data(), sizeof(int) * (size_t)m); } }, 1 << 14);
}

template <class Mat>
inline LevelOrdering make_ordering(const Mat& A, bool multicolor, int row_align, int sigma, int reorder = 0, const std::vector<int>* ext_base = nullptr, bool idx_sorted = false,
                                   PreColoring* pre = nullptr) {
    LevelOrdering o;
    const int n = A.n_outer;
    o.n = n;
    const bool trace = EnvSwitches::get().trace_setup;
    auto tph = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!trace) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gmg setup] make_ordering n=%d %-12s %.2f ms\n", n, what, std::chrono::duration<double, std::milli>(now - tph).count());
        tph = now;
    };
    std::vector<int> base;      // visit order (empty = natural)
    if (wants_locality_reorder(A, reorder)) {
        if (ext_base && (int)ext_base->size() == n) base = *ext_base;
        else {
            std::vector<int> block_of, mem_begin;
            grow_patches(A, 4096, block_of, base, mem_begin);
        }
        o.reordered = true;
    }
    phase("locality");
    // While the (sequential) colouring runs, another thread first-touches the two index arrays of the result: fresh
    // memory costs ~0.2 ms per MB of page faults on one core, more than the bucketing that fills them.
    const int n_pad_max = n + 256 * std::max(row_align, 1);
    auto prefill = std::async(std::launch::async, [&] {
        o.new2old.assign((size_t)n_pad_max, -1);
        o.old2new.assign((size_t)n, -1);
    });
    std::vector<int> color;
    RawVec<unsigned char> c8_own;
    const unsigned char* c8 = nullptr;
    if (multicolor && pre && pre->n_colors >= 0 && base.empty() && pre->c8.n >= (size_t)n) {      // coloured ahead (greedy_coloring_ahead): the same bytes
        c8_own = std::move(pre->c8);
        o.n_colors = pre->n_colors;
        c8 = c8_own.data();
    } else if (multicolor) {
        o.n_colors = greedy_coloring_bytes(A, c8_own, base, idx_sorted);
        if (o.n_colors < 0) { c8_own.resize(0); o.n_colors = greedy_coloring_general(A, color, base); }
        else c8 = c8_own.data();
    } else { color.assign(n, 0); o.n_colors = n > 0 ? 1 : 0; }
    const bool bytes = c8 != nullptr;
    auto colour_of = [&](int i) -> int { return bytes ? (int)c8[(size_t)i] : color[(size_t)i]; };
    phase("colouring");
    // Stable counting sort of the visit sequence by colour, threaded: per-chunk histograms give every chunk its write
    // offsets, so the result does not depend on the number of threads.
    const int T = std::max(1, std::min(hw_threads(), 32));
    const int nchunk = n >= 65536 ? T : 1;
    const int chunk = (n + nchunk - 1) / std::max(nchunk, 1);
    std::vector<std::vector<int>> hist(nchunk, std::vector<int>(o.n_colors, 0));
    parallel_ranges(nchunk, nchunk, [&](int lo, int hi, int) {
        for (int q = lo; q < hi; ++q) {
            const int t0 = q * chunk, t1 = std::min(n, t0 + chunk);
            std::vector<int>& hq = hist[q];
            for (int t = t0; t < t1; ++t) hq[colour_of(base.empty() ? t : base[t])]++;
        }
    }, 1);
    std::vector<int> count(o.n_colors, 0);
    for (int q = 0; q < nchunk; ++q) for (int c = 0; c < o.n_colors; ++c) count[c] += hist[q][c];
    // Device order of the colour classes: ascending size (stable, so classes of equal size -- a regular mesh -- keep the order the
    // greedy colouring gave them).  Any order of the classes is a Gauss-Seidel ordering; this one ends every sweep with the largest
    // class, whose launch also produces its rows' residual / residual-check sums (engine_cycle.hip.hpp::fold_residual, fold_norm), and
    // gets the stragglers an irregular graph leaves behind (two classes of < 2 000 rows on an 8 M-vertex torus) out of the way first.
    std::vector<int> colour_at(o.n_colors), slot_of(o.n_colors);
    std::iota(colour_at.begin(), colour_at.end(), 0);
    std::stable_sort(colour_at.begin(), colour_at.end(), [&](int a, int b) { return count[a] < count[b]; });
    for (int sl = 0; sl < o.n_colors; ++sl) slot_of[colour_at[sl]] = sl;
    o.color_begin.assign(o.n_colors + 1, 0);
    for (int sl = 0; sl < o.n_colors; ++sl) o.color_begin[sl + 1] = o.color_begin[sl] + round_up(count[colour_at[sl]], row_align);
    o.n_pad = o.n_colors ? o.color_begin[o.n_colors] : 0;
    if (o.n_pad == 0) o.n_pad = row_align;
    prefill.wait();
    if (o.n_pad <= n_pad_max) o.new2old.resize((size_t)o.n_pad);       // shrinks: no reallocation
    else o.new2old.assign((size_t)o.n_pad, -1);
    {
        std::vector<int> run(o.n_colors);
        for (int c = 0; c < o.n_colors; ++c) run[c] = o.color_begin[slot_of[c]];
        for (int q = 0; q < nchunk; ++q) for (int c = 0; c < o.n_colors; ++c) { const int h = hist[q][c]; hist[q][c] = run[c]; run[c] += h; }
    }
    parallel_ranges(nchunk, nchunk, [&](int lo, int hi, int) {
        for (int q = lo; q < hi; ++q) {
            const int t0 = q * chunk, t1 = std::min(n, t0 + chunk);
            std::vector<int>& fill = hist[q];
            for (int t = t0; t < t1; ++t) { const int i = base.empty() ? t : base[t]; o.new2old[fill[colour_of(i)]++] = i; }
        }
    }, 1);
    phase("bucket");
    if (sigma > 0) {
        std::vector<std::pair<int, int>> windows;
        for (int sl = 0; sl < o.n_colors; ++sl) {
            const int lo = o.color_begin[sl], hi = lo + count[colour_at[sl]];
            for (int w = lo; w < hi; w += sigma) windows.emplace_back(w, std::min(hi, w + sigma));
        }
        parallel_ranges((int)windows.size(), T, [&](int lo, int hi, int) {
            for (int q = lo; q < hi; ++q) {
                auto longer = [&](int a, int b) { return (A.ptr[a + 1] - A.ptr[a]) > (A.ptr[b + 1] - A.ptr[b]); };
                const auto w0 = o.new2old.begin() + windows[q].first, w1 = o.new2old.begin() + windows[q].second;
                // a regular mesh has one row length: nothing to do (a stable sort leaves a sorted window as it is)
                if (!std::is_sorted(w0, w1, longer)) std::stable_sort(w0, w1, longer);
            }
        }, 64);
    }
    phase("window sort");
    parallel_ranges(o.n_pad, T, [&](int lo, int hi, int) {
        for (int r = lo; r < hi; ++r)
            if (o.new2old[r] >= 0) o.old2new[o.new2old[r]] = r;
    });
    phase("inverse");
    return o;
}

// The compact patches of a blocked level (grow_patches output).
struct PatchSet {
    std::vector<int> block_of, members, mem_begin;
    int n = 0;
    bool valid() const { return !mem_begin.empty(); }
};

template <class Mat>
inline PatchSet grow_patch_set(const Mat& G, int block_rows) {
    PatchSet p;
    p.n = G.n_outer;
    grow_patches(G, block_rows, p.block_of, p.members, p.mem_begin);
    return p;
}

// Graph on the coarse points of one prolongation: p ~ q when some fine row interpolates from both (the edges of the
// coarse triangulation the barycentric weights come from).  pattern(U^T U), symmetric, sorted, with the diagonal.
// It is known as soon as the hierarchy is, so the patches of level k + 1 can be grown before any system arrives.
// U: by coarse column (outer = coarse); Urows: the same matrix by fine row (outer = fine).
inline Compressed coarse_point_graph(const Compressed& U, const Compressed& Urows) {
    const int nc = U.n_outer;
    Compressed G;
    G.n_outer = nc; G.n_inner = nc;
    std::vector<std::vector<int>> adj(nc);
    parallel_ranges(nc, std::min(hw_threads(), 32), [&](int lo, int hi, int) {
        for (int p = lo; p < hi; ++p) {
            std::vector<int>& a = adj[p];
            a.reserve((size_t)(U.ptr[p + 1] - U.ptr[p]) * 3 + 1);      // exact upper bound: one allocation per point
            for (int e = U.ptr[p]; e < U.ptr[p + 1]; ++e) {
                const int i = U.idx[e];
                for (int f = Urows.ptr[i]; f < Urows.ptr[i + 1]; ++f) a.push_back(Urows.idx[f]);
            }
            a.push_back(p);
            std::sort(a.begin(), a.end());
            a.erase(std::unique(a.begin(), a.end()), a.end());
        }
    });
    G.ptr.assign((size_t)nc + 1, 0);
    for (int p = 0; p < nc; ++p) G.ptr[p + 1] = G.ptr[p] + (int)adj[p].size();
    G.idx.resize(G.ptr[nc]);
    parallel_ranges(nc, std::min(hw_threads(), 32), [&](int lo, int hi, int) {
        for (int p = lo; p < hi; ++p) std::copy(adj[p].begin(), adj[p].end(), G.idx.begin() + G.ptr[p]);
    });
    return G;
}

// Block ordering for the block-hybrid smoother.  Blocks are grown breadth-first over a graph of the level
// from seeds taken on the frontier of what is already assigned, so they are compact patches whatever the
// input vertex order is (a contiguous range of a row-major mesh ordering would be a thin strip with nearly
// every edge cut).  Inside a block: greedy colouring of the in-block subgraph of A in BFS order, rows sorted by
// colour.  Every block is padded to a multiple of 64 rows so SELL slices never straddle blocks.
// `patches`: grown beforehand (over the coarse point graph of the hierarchy) or, if null, grown here over A's graph.
template <class Mat>
inline LevelOrdering make_block_ordering(const Mat& A, int block_rows, const PatchSet* patches = nullptr, int block_align = 1) {
    LevelOrdering o;
    const int n = A.n_outer;
    o.n = n;
    o.blocked = true;
    const bool trace = EnvSwitches::get().trace_setup;
    auto tph = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
        if (!trace) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gmg setup] make_block_ordering n=%d %-12s %.2f ms\n", n, what, std::chrono::duration<double, std::milli>(now - tph).count());
        tph = now;
    };
    // ---- phase 1 (sequential graph traversal): grow the blocks, record their members in BFS order
    PatchSet own;
    if (!patches || !patches->valid() || patches->n != n) { own = grow_patch_set(A, block_rows); patches = &own; }
    const std::vector<int>&block_of = patches->block_of, &members = patches->members, &mem_begin = patches->mem_begin;
    phase("grow");
    const int nb = (int)mem_begin.size() - 1;
    const int T = std::max(1, std::min(hw_threads(), 32));
    // ---- phase 2 (threaded over blocks): greedy colouring of the in-block subgraph in BFS order, colour sort, padding
    // block_align > 1 (a level 0 that P ranks cut into P runs of whole blocks, gmg_config::row_align = 64 P): empty 64-row blocks are appended
    // until the block count is a multiple of it (padding rows: no entries, unit diagonal, zero right-hand side)
    const int nb_all = block_align > 1 ? round_up(std::max(nb, 1), block_align) : nb;
    o.blk_begin.assign((size_t)nb_all + 1, 0);
    for (int b = 0; b < nb_all; ++b) o.blk_begin[b + 1] = o.blk_begin[b] + (b < nb ? round_up(mem_begin[b + 1] - mem_begin[b], kSlice) : kSlice);
    o.n_pad = nb_all ? o.blk_begin[nb_all] : kSlice;
    o.new2old.assign(o.n_pad, -1);
    o.row_color.assign(o.n_pad, 0);
    o.blk_ncolors.assign(nb_all, 0);
    std::vector<int> color(n, -1);
    // visit order of a block's members for the first-fit colouring: SMALLEST-LAST (repeatedly remove a vertex of least remaining in-block degree; colour
    // in reverse removal order; default since round 6) or breadth-first (the members' order; GMG_BLOCK_COLOURING=bfs) -- on the Galerkin levels' in-block
    // graphs smallest-last needs ~13 % fewer colours, i.e. fewer sequential steps of the block sweep's in-block solve (kernels.hip.hpp::ep_block_lower):
    // 3 M d = 3 0.998 -> 0.984 ms per cycle, point cloud 0.563 -> 0.556, d = 1 -2.8 us, same cycle counts.  A first version (local graphs as index
    // lists, a scan per removal, then buckets) made the ordering of the 506 k-row level 1.8 x as slow and a cold gmg_set_system 9 ms longer; with a
    // 64-row block's graph as 64 words and its degrees as 64 bytes the ordering tasks take 1-2 ms more than breadth-first and run beside each other
    // (profiles/r06/block_colouring_order_ab.txt).
    const int max_members = [&] { int m = 0; for (int b = 0; b < nb; ++b) m = std::max(m, mem_begin[b + 1] - mem_begin[b]); return m; }();
    const bool smallest_last = EnvSwitches::get().block_smallest_last && nb < (1 << 21) && max_members <= 1024;
    // block and place among its block's members of every vertex in one word (one random access per neighbour instead of two): block << 10 | place
    std::vector<unsigned> packed;
    if (smallest_last) {
        packed.assign((size_t)n, 0u);
        parallel_ranges(nb, T, [&](int lo, int hi, int) { for (int b = lo; b < hi; ++b) for (int m = mem_begin[b]; m < mem_begin[b + 1]; ++m) packed[(size_t)members[m]] = ((unsigned)b << 10) | (unsigned)(m - mem_begin[b]); }, 64);
    }
    parallel_ranges(nb, T, [&](int lo, int hi, int) {
        std::vector<char> forbid;
        std::vector<int> mem, visit, deg, lptr, lidx, lcol;
        std::vector<uint64_t> bucket;
        std::vector<char> gone;
        // (the scratch arrays of a block are a few hundred bytes each and are rewritten thousands of times per millisecond: as small heap blocks the arrays
        // of different threads came to lie in the same cache lines, and eight threads took longer than one -- every array gets its capacity for the
        // largest block at once, far beyond a cache line)
        if (smallest_last) {
            visit.reserve(4096); deg.reserve(4096); lptr.reserve(4096); lidx.reserve(65536); lcol.reserve(4096); gone.reserve(16384);
            bucket.reserve((size_t)1025 * 16);
        }
        for (int b = lo; b < hi; ++b) {
            int ncol = 0;
            const int m0 = mem_begin[b], mcount = mem_begin[b + 1] - m0;
            visit.resize((size_t)mcount);
            for (int i = 0; i < mcount; ++i) visit[(size_t)i] = members[m0 + i];
            bool coloured = false;
            if (smallest_last && mcount > 2 && mcount <= 64) {
                // a block of one wavefront's rows: the in-block graph as one 64-bit word per member, remaining degrees as bytes (a removed member's
                // degree becomes 255, so the least degree is a plain minimum over 64 bytes), first fit over the words -- ~1.5 us per block on top of the
                // row scan every ordering of the block needs
                uint64_t adj[64];
                unsigned char dg[64];
                for (int i = 0; i < 64; ++i) { adj[i] = 0; dg[i] = 255; }
                for (int i = 0; i < mcount; ++i) {
                    const int v = members[m0 + i];
                    if (i + 6 < mcount) { const int vn = members[m0 + i + 6]; __builtin_prefetch(&A.idx[A.ptr[vn]]); __builtin_prefetch(&A.idx[A.ptr[vn]] + 16); }
                    uint64_t a = 0;
                    for (int p = A.ptr[v]; p < A.ptr[v + 1]; ++p) { const int w = A.idx[p]; const unsigned pk = packed[(size_t)w]; if (w != v && (int)(pk >> 10) == b) a |= (uint64_t)1 << (pk & 63u); }
                    adj[i] = a;
                    dg[i] = (unsigned char)__builtin_popcountll(a);
                }
                int order[64];
                for (int step = mcount - 1; step >= 0; --step) {
                    int best = 0;
                    unsigned char bd = dg[0];
                    for (int i = 1; i < 64; ++i) if (dg[i] < bd) { bd = dg[i]; best = i; }      // (the earliest member among equals)
                    dg[best] = 255;
                    order[step] = best;
                    for (uint64_t nb_bits = adj[best]; nb_bits; nb_bits &= nb_bits - 1) { const int w = __builtin_ctzll(nb_bits); if (dg[w] != 255) --dg[w]; }
                }
                signed char lc8[64];
                for (int i = 0; i < 64; ++i) lc8[i] = -1;
                for (int m = 0; m < mcount; ++m) {
                    const int i = order[m];
                    uint64_t mask = 0;
                    for (uint64_t nb_bits = adj[i]; nb_bits; nb_bits &= nb_bits - 1) { const int cw = lc8[__builtin_ctzll(nb_bits)]; if (cw >= 0) mask |= (uint64_t)1 << cw; }
                    const int c = __builtin_ctzll(~mask);                    // (at most 63 neighbours: a colour below 64 is free)
                    lc8[i] = (signed char)c;
                    if (c >= ncol) ncol = c + 1;
                }
                for (int i = 0; i < mcount; ++i) color[(size_t)members[m0 + i]] = lc8[i];
                coloured = true;
            }
            if (!coloured && smallest_last && mcount > 2) {
                // (bigger blocks -- the 256-row blocks of the quad layout: the same order with buckets by remaining degree)
                // local adjacency of the in-block subgraph (places among the block's members)
                lptr.assign((size_t)mcount + 1, 0); lidx.clear(); deg.assign((size_t)mcount, 0); gone.assign((size_t)mcount, 0);
                for (int i = 0; i < mcount; ++i) {
                    const int v = members[m0 + i];
                    if (i + 6 < mcount) { const int vn = members[m0 + i + 6]; __builtin_prefetch(&A.idx[A.ptr[vn]]); __builtin_prefetch(&A.idx[A.ptr[vn]] + 16); }
                    for (int p = A.ptr[v]; p < A.ptr[v + 1]; ++p) { const int w = A.idx[p]; const unsigned pk = packed[(size_t)w]; if (w != v && (int)(pk >> 10) == b) lidx.push_back((int)(pk & 1023u)); }
                    lptr[(size_t)i + 1] = (int)lidx.size();
                }
                // buckets by remaining degree as bit sets over the block's members: the vertex of least degree -- the earliest member among equals -- is
                // the lowest bit of the first non-empty bucket, and the least degree drops by at most one per removal
                const int W = (mcount + 63) / 64;
                int dmax = 0;
                for (int i = 0; i < mcount; ++i) { deg[(size_t)i] = lptr[(size_t)i + 1] - lptr[(size_t)i]; dmax = std::max(dmax, deg[(size_t)i]); }
                bucket.assign((size_t)(dmax + 1) * W, 0);
                for (int i = 0; i < mcount; ++i) bucket[(size_t)deg[(size_t)i] * W + (i >> 6)] |= (uint64_t)1 << (i & 63);
                int dmin = 0;
                for (int step = mcount - 1; step >= 0; --step) {
                    int best = -1;
                    for (dmin = std::max(dmin - 1, 0); dmin <= dmax && best < 0; ++dmin)
                        for (int q = 0; q < W; ++q) { const uint64_t bits = bucket[(size_t)dmin * W + q]; if (bits) { best = q * 64 + __builtin_ctzll(bits); break; } }
                    --dmin;                                                    // (the bucket the vertex came from)
                    gone[(size_t)best] = 1;
                    bucket[(size_t)deg[(size_t)best] * W + (best >> 6)] &= ~((uint64_t)1 << (best & 63));
                    visit[(size_t)step] = best;                                // (a place, not a vertex: the colouring below stays in the local graph)
                    for (int p = lptr[(size_t)best]; p < lptr[(size_t)best + 1]; ++p) {
                        const int w = lidx[(size_t)p];
                        if (gone[(size_t)w]) continue;
                        bucket[(size_t)deg[(size_t)w] * W + (w >> 6)] &= ~((uint64_t)1 << (w & 63));
                        --deg[(size_t)w];
                        bucket[(size_t)deg[(size_t)w] * W + (w >> 6)] |= (uint64_t)1 << (w & 63);
                    }
                }
                // first fit in that order, on the local graph
                lcol.assign((size_t)mcount, -1);
                for (int m = 0; m < mcount; ++m) {
                    const int i = visit[(size_t)m];
                    uint64_t mask = 0;
                    bool wide = false;
                    for (int p = lptr[(size_t)i]; p < lptr[(size_t)i + 1]; ++p) { const int cw = lcol[(size_t)lidx[(size_t)p]]; if (cw >= 64) wide = true; else if (cw >= 0) mask |= (uint64_t)1 << cw; }
                    int c;
                    if (~mask != 0) c = __builtin_ctzll(~mask);
                    else {
                        (void)wide;
                        forbid.assign((size_t)ncol + 1, 0);
                        for (int p = lptr[(size_t)i]; p < lptr[(size_t)i + 1]; ++p) { const int cw = lcol[(size_t)lidx[(size_t)p]]; if (cw >= 0) forbid[(size_t)cw] = 1; }
                        c = 64;
                        while (c < ncol && forbid[(size_t)c]) ++c;
                    }
                    lcol[(size_t)i] = c;
                    if (c >= ncol) ncol = c + 1;
                }
                for (int i = 0; i < mcount; ++i) color[(size_t)members[m0 + i]] = lcol[(size_t)i];
                coloured = true;
            }
            if (!coloured)
            for (int m = 0; m < mcount; ++m) {
                const int v = visit[(size_t)m];
                // the members come in breadth-first order, their rows from all over A: fetch the row a few members ahead
                if (m + 6 < mcount) { const int vn = visit[(size_t)m + 6]; __builtin_prefetch(&A.idx[A.ptr[vn]]); __builtin_prefetch(&A.idx[A.ptr[vn]] + 16); }
                // first free colour from a 64-bit mask of the neighbours' colours (the general list only beyond 64 colours)
                uint64_t mask = 0;
                for (int p = A.ptr[v]; p < A.ptr[v + 1]; ++p) {
                    const int w = A.idx[p];
                    if (w == v || block_of[w] != b) continue;
                    const int cw = color[w];
                    if (cw >= 0 && cw < 64) mask |= (uint64_t)1 << cw;
                }
                int c;
                if (~mask != 0) c = __builtin_ctzll(~mask);
                else {
                    forbid.assign((size_t)ncol + 1, 0);
                    for (int p = A.ptr[v]; p < A.ptr[v + 1]; ++p) {
                        const int w = A.idx[p];
                        if (w != v && block_of[w] == b && color[w] >= 0) forbid[color[w]] = 1;
                    }
                    c = 64;
                    while (c < ncol && forbid[c]) ++c;
                }
                color[v] = c;
                if (c >= ncol) ncol = c + 1;
            }
            mem.assign(members.begin() + mem_begin[b], members.begin() + mem_begin[b + 1]);
            std::stable_sort(mem.begin(), mem.end(), [&](int x, int y) { return color[x] < color[y]; });
            const int begin = o.blk_begin[b];
            for (size_t i = 0; i < mem.size(); ++i) {
                o.new2old[begin + i] = mem[i];
                o.row_color[begin + i] = (unsigned char)std::min(color[mem[i]], 255);
            }
            o.blk_ncolors[b] = ncol;
        }
    }, 64);
    phase("colour+sort");
    for (int b = 0; b < nb; ++b) o.n_colors = std::max(o.n_colors, o.blk_ncolors[b]);
    o.color_begin = {0, o.n_pad};      // not colour-major: a single range
    o.old2new.assign(n, -1);
    parallel_ranges(o.n_pad, T, [&](int lo, int hi, int) {
        for (int r = lo; r < hi; ++r)
            if (o.new2old[r] >= 0) o.old2new[o.new2old[r]] = r;
    }, 65536);
    phase("inverse");
    return o;
}

// Locality-preserving order of the finest level's points derived from the hierarchy alone: the coarsest points in
// breadth-first order over their point graph, and on every finer level the points grouped by parent (the coarse point
// with the largest prolongation weight), parents in the order just built, siblings in index order.  Points that are
// close on the surface end up close in the order whatever the input numbering is.  Urows[k]: U_k by fine row;
// GL: point graph of the coarsest level.  Returns new -> old of level 0.
inline std::vector<int> cluster_order(const std::vector<Compressed>& Urows, const Compressed& GL) {
    const int L = (int)Urows.size();
    const int nL = GL.n_outer;
    // coarsest level: BFS order over GL (restarting at the lowest unvisited index)
    std::vector<int> pos(nL, -1);
    {
        std::vector<int> queue;
        queue.reserve(nL);
        for (int s = 0; s < nL; ++s) {
            if (pos[s] >= 0) continue;
            pos[s] = (int)queue.size(); queue.push_back(s);
            for (size_t h = queue.size() - 1; h < queue.size(); ++h) {
                const int v = queue[h];
                for (int p = GL.ptr[v]; p < GL.ptr[v + 1]; ++p) { const int w = GL.idx[p]; if (pos[w] < 0) { pos[w] = (int)queue.size(); queue.push_back(w); } }
            }
        }
    }
    std::vector<int> order;
    for (int k = L - 1; k >= 0; --k) {
        const Compressed& R = Urows[k];
        const int nf = R.n_outer, nc = R.n_inner;
        const int T = std::max(1, std::min(hw_threads(), 32));
        // key[i] = position of i's parent in the coarse order (orphans last)
        std::vector<int> key(nf);
        parallel_ranges(nf, T, [&](int lo, int hi, int) {
            for (int i = lo; i < hi; ++i) {
                int best = -1; double bw = -1.0;
                for (int p = R.ptr[i]; p < R.ptr[i + 1]; ++p) {
                    const double w = std::abs(R.val[p]);
                    if (w > bw || (w == bw && R.idx[p] < best)) { bw = w; best = R.idx[p]; }
                }
                key[i] = best >= 0 ? pos[best] : nc;
            }
        });
        // stable counting sort by key
        std::vector<int> start((size_t)nc + 2, 0);
        for (int i = 0; i < nf; ++i) start[key[i] + 1]++;
        for (int c = 0; c <= nc; ++c) start[c + 1] += start[c];
        order.assign(nf, 0);
        for (int i = 0; i < nf; ++i) order[start[key[i]]++] = i;
        pos.assign(nf, 0);
        parallel_ranges(nf, T, [&](int lo, int hi, int) { for (int r = lo; r < hi; ++r) pos[order[r]] = r; });
    }
    return order;
}

// Breadth-first order of the points of a neighbour table (n x K, rows padded with -1, as the reference's `neigh`): the order in
// which a sequential breadth-first search from point 0 (restarting at the lowest unvisited index) takes them out of its queue.
// Consecutive points lie next to each other on a wavefront and their neighbours next to each other on the wavefronts before and
// after it, so the j-th gathers of 64 consecutive rows fall into a few cache lines: on a randomly numbered 3 M-vertex mesh the
// fine-level residual runs at 49 us with this base order, 72 us with the hierarchy's cluster order (gathers scattered over a 2-D
// patch) -- a kNN point-cloud graph, whose wavefronts are thick and ragged, is the other way round (93 / 83 us), which is why
// gmg_set_system scores both on the actual matrix (engine_setup.hip.hpp::choose_base_order).  Sequential by definition; the rows
// of the queue's next entries are prefetched.  Returns new -> old.
inline std::vector<int> bfs_point_order(const int* neigh, int n, int K) {
    std::vector<int> order;
    order.reserve((size_t)n);
    std::vector<unsigned char> seen((size_t)n, 0);
    for (int s = 0; s < n; ++s) {
        if (seen[s]) continue;
        seen[s] = 1; order.push_back(s);
        for (size_t h = order.size() - 1; h < order.size(); ++h) {
            if (h + 8 < order.size()) __builtin_prefetch(neigh + (size_t)order[h + 8] * K);
            const int* row = neigh + (size_t)order[h] * K;
            for (int j = 0; j < K; ++j) {
                const int w = row[j];
                if (w < 0) break;
                if (!seen[w]) { seen[w] = 1; order.push_back(w); }
            }
        }
    }
    return order;
}

// Host twin of gmgs::order_gather_score (setup_kernels.hip.hpp; the specification of the score): distinct 128-byte lines touched
// by the j-th gathers of 64 rows taken at every fourth position of `order`, summed over n_win sampled windows, and the number of
// entries.  Integer sums: the same numbers as the device kernel's.
template <class Mat>
inline void order_gather_score_host(const Mat& A, const std::vector<int>& order, int n_win, unsigned long long out[2]) {
    const int n = A.n_outer;
    out[0] = out[1] = 0;
    if (n < 512) return;
    std::vector<int> inv((size_t)n);
    for (int r = 0; r < n; ++r) inv[order[r]] = r;
    constexpr int kMaxRow = 32;
    std::vector<int> p((size_t)64 * kMaxRow), len(64);
    for (int w = 0; w < n_win; ++w) {
        const int start = (int)((long long)w * (n - 256) / n_win);
        int maxlen = 0;
        for (int l = 0; l < 64; ++l) {
            const int v = order[start + 4 * l], b = A.ptr[v];
            len[l] = std::min(A.ptr[v + 1] - b, kMaxRow);
            for (int q = 0; q < len[l]; ++q) p[(size_t)l * kMaxRow + q] = inv[A.idx[b + q]];
            std::sort(p.begin() + (size_t)l * kMaxRow, p.begin() + (size_t)l * kMaxRow + len[l]);
            maxlen = std::max(maxlen, len[l]);
        }
        for (int j = 0; j < maxlen; ++j)
            for (int l = 0; l < 64; ++l) {
                if (j >= len[l]) continue;
                ++out[1];
                const int line = p[(size_t)l * kMaxRow + j] >> 4;
                bool first = true;
                for (int m = 0; m < l && first; ++m) first = !(j < len[m] && (p[(size_t)m * kMaxRow + j] >> 4) == line);
                out[0] += first ? 1 : 0;
            }
    }
}

// mean |i - neighbour| of a neighbour table (sampled): the same locality measure as mean_index_distance
inline double mean_index_distance_table(const int* neigh, int n, int K) {
    const int step = std::max(1, n / 65536);
    double sum = 0.0; long cnt = 0;
    for (int i = 0; i < n; i += step)
        for (int j = 0; j < K && neigh[(size_t)i * K + j] >= 0; ++j) { sum += std::abs(neigh[(size_t)i * K + j] - i); ++cnt; }
    return cnt ? sum / cnt : 0.0;
}

inline LevelOrdering identity_ordering(int n) {
    LevelOrdering o;
    o.n = n; o.n_pad = round_up(std::max(n, 1), kSlice); o.n_colors = 1;
    o.color_begin = {0, o.n_pad};
    o.new2old.assign(o.n_pad, -1);
    o.old2new.resize(n);
    for (int i = 0; i < n; ++i) { o.new2old[i] = i; o.old2new[i] = i; }
    return o;
}
// Row-wise staging of a matrix in DEVICE numbering: ptr (prefix sums of row lengths), then (idx, val) sorted by column.
struct RowStage {
    std::vector<int64_t> ptr;
    RawVec<int> idx;
    RawVec<double> val;
};

// lengths[r] for r < np computed in parallel by `len_of(r)`, then one sequential prefix sum.
template <class LenFn>
inline void stage_lengths(int np, RowStage& st, LenFn&& len_of) {
    st.ptr.assign((size_t)np + 1, 0);
    parallel_ranges(np, hw_threads(), [&](int lo, int hi, int) {
        for (int r = lo; r < hi; ++r) st.ptr[r + 1] = len_of(r);
    });
    for (int r = 0; r < np; ++r) st.ptr[r + 1] += st.ptr[r];
    st.idx.resize((size_t)st.ptr[np]);
    st.val.resize((size_t)st.ptr[np]);
}

// Staged rows -> SELL-64.  If sort_sigma > 0 the rows are re-sorted by length inside windows and row_of records
// the output row.  lpr = lanes per row: 1 (one row per lane, 64 rows per slice) or 4 ("quad" layout for the
// latency-bound coarse levels: 16 rows per slice, entry e of a row goes to sub-lane e % 4 at depth e / 4, the
// kernel adds the four partial sums with two cross-lane steps -> 4x shorter dependency chains, 4x more waves).
inline SellHost csr_to_sell(int n_rows_pad, int n_cols, const RowStage& st, int sort_sigma, int lpr = 1) {
    const std::vector<int64_t>& ptr = st.ptr;
    const int rps = kSlice / lpr;                 // rows per slice
    SellHost s;
    s.lpr = lpr;
    s.n_rows_pad = n_rows_pad; s.n_cols = n_cols; s.n_slices = n_rows_pad / rps;
    s.nnz_real = ptr[n_rows_pad];
    std::vector<int> order;
    const bool sorted = sort_sigma > 0;
    if (sorted) {
        order.resize(n_rows_pad);
        const int nwin = (n_rows_pad + sort_sigma - 1) / sort_sigma;
        parallel_ranges(nwin, hw_threads(), [&](int lo, int hi, int) {
            for (int wi = lo; wi < hi; ++wi) {
                int w = wi * sort_sigma, we = std::min(n_rows_pad, w + sort_sigma);
                std::iota(order.begin() + w, order.begin() + we, w);
                std::stable_sort(order.begin() + w, order.begin() + we, [&](int a, int b) { return (ptr[a + 1] - ptr[a]) > (ptr[b + 1] - ptr[b]); });
            }
        });
        s.row_of = order;
    }
    auto row_at = [&](int i) { return sorted ? order[i] : i; };
    s.slice_ptr.assign((size_t)s.n_slices + 1, 0);
    parallel_ranges(s.n_slices, hw_threads(), [&](int lo, int hi, int) {
        for (int sl = lo; sl < hi; ++sl) {
            int64_t w = 0;
            for (int l = 0; l < rps; ++l) { int r = row_at(sl * rps + l); w = std::max<int64_t>(w, (ptr[r + 1] - ptr[r] + lpr - 1) / lpr); }
            s.slice_ptr[sl + 1] = w * kSlice;
        }
    });
    for (int sl = 0; sl < s.n_slices; ++sl) s.slice_ptr[sl + 1] += s.slice_ptr[sl];
    s.col.resize((size_t)s.stored());
    s.val.resize((size_t)s.stored());
    parallel_ranges(s.n_slices, hw_threads(), [&](int lo, int hi, int) {
        for (int sl = lo; sl < hi; ++sl) {
            const int64_t base = s.slice_ptr[sl];
            const int64_t w = (s.slice_ptr[sl + 1] - base) / kSlice;
            for (int l = 0; l < rps; ++l) {
                const int r = row_at(sl * rps + l);
                const int64_t len = ptr[r + 1] - ptr[r], p0 = ptr[r];
                for (int64_t e = 0; e < w * lpr; ++e) {
                    const int64_t q = base + (e / lpr) * kSlice + l * lpr + (e % lpr);
                    if (e < len) { s.col[q] = st.idx[p0 + e]; s.val[q] = st.val[p0 + e]; }
                    else { s.col[q] = 0; s.val[q] = 0.0; }                     // padding: 0 * x[0]
                }
            }
        }
    });
    return s;
}

inline void sort_row(std::vector<std::pair<int, double>>& row) {
    std::sort(row.begin(), row.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
}

// A (symmetric, natural numbering) -> off-diagonal SELL + diagonal, in the level's device numbering.
// Returns false (and sets err) if a real row has no / a zero diagonal entry.
inline bool build_operator_sell(const Compressed& A, const LevelOrdering& o, int lpr, SellHost& out,
                                std::vector<double>& diag, std::string& err) {
    if (lpr != 4) lpr = 1;
    const int np = o.n_pad;
    diag.assign(np, 1.0);
    std::atomic<int> bad_row{-1};
    RowStage st;
    stage_lengths(np, st, [&](int r) -> int64_t {
        int old = o.new2old[r];
        if (old < 0) return 0;
        int64_t len = 0;
        bool has = false;
        for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) {
            if (A.idx[p] == old) { has = true; diag[r] = A.val[p]; }
            else ++len;
        }
        if (!has || diag[r] == 0.0) bad_row.store(old);
        return len;
    });
    if (bad_row.load() >= 0) { err = "system matrix has a missing or zero diagonal entry at row " + std::to_string(bad_row.load()); return false; }
    parallel_ranges(np, hw_threads(), [&](int lo, int hi, int) {
        std::vector<std::pair<int, double>> row;
        for (int r = lo; r < hi; ++r) {
            int old = o.new2old[r];
            if (old < 0) continue;
            row.clear();
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p)
                if (A.idx[p] != old) row.emplace_back(o.old2new[A.idx[p]], A.val[p]);
            sort_row(row);
            int64_t q = st.ptr[r];
            for (auto& e : row) { st.idx[q] = e.first; st.val[q] = e.second; ++q; }
        }
    });
    out = csr_to_sell(np, np, st, 0, lpr);
    return true;
}

// Blocked level: split the off-diagonal part of A into the entries that stay inside the row's block
// (`in`: column = device row MINUS the block's first row, < 65536, for the LDS-resident sweep) and the
// entries that leave it (`out`: device column; applied to the previous iterate, Jacobi-style).
inline void build_operator_sell_split(const Compressed& A, const LevelOrdering& o, SellHost& in, SellHost& out, int lpr = 1) {
    const int np = o.n_pad;
    std::vector<int> blk_of_row(np, 0);
    for (int b = 0; b < o.n_blocks(); ++b)
        for (int r = o.blk_begin[b]; r < o.blk_begin[b + 1]; ++r) blk_of_row[r] = b;
    RowStage sin, sout;
    auto count = [&](int r, bool inside) -> int64_t {
        int old = o.new2old[r];
        if (old < 0) return 0;
        int64_t len = 0;
        for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) {
            if (A.idx[p] == old) continue;
            if ((blk_of_row[o.old2new[A.idx[p]]] == blk_of_row[r]) == inside) ++len;
        }
        return len;
    };
    stage_lengths(np, sin, [&](int r) { return count(r, true); });
    stage_lengths(np, sout, [&](int r) { return count(r, false); });
    parallel_ranges(np, hw_threads(), [&](int lo_, int hi_, int) {
        std::vector<std::pair<int, double>> row;
        for (int r = lo_; r < hi_; ++r) {
            int old = o.new2old[r];
            if (old < 0) continue;
            row.clear();
            for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p)
                if (A.idx[p] != old) row.emplace_back(o.old2new[A.idx[p]], A.val[p]);
            sort_row(row);
            int64_t qi = sin.ptr[r], qo = sout.ptr[r];
            const int base = o.blk_begin[blk_of_row[r]];
            for (auto& e : row) {
                if (blk_of_row[e.first] == blk_of_row[r]) { sin.idx[qi] = e.first - base; sin.val[qi] = e.second; ++qi; }
                else { sout.idx[qo] = e.first; sout.val[qo] = e.second; ++qo; }
            }
        }
    });
    in = csr_to_sell(np, np, sin, 0, lpr);
    out = csr_to_sell(np, np, sout, 0, lpr);
}

// Generic: rows of `Mrows` (compressed, outer = rows in natural numbering of the row space) mapped into
// device numbering of the row space (orow) and column space (ocol).
inline SellHost build_transfer_sell(const Compressed& Mrows, const LevelOrdering& orow, const LevelOrdering& ocol,
                                    int sort_sigma, int lpr = 1) {
    const int np = orow.n_pad;
    RowStage st;
    stage_lengths(np, st, [&](int r) -> int64_t {
        int old = orow.new2old[r];
        return old >= 0 ? Mrows.ptr[old + 1] - Mrows.ptr[old] : 0;
    });
    parallel_ranges(np, hw_threads(), [&](int lo, int hi, int) {
        std::vector<std::pair<int, double>> row;
        for (int r = lo; r < hi; ++r) {
            int old = orow.new2old[r];
            if (old < 0) continue;
            row.clear();
            for (int p = Mrows.ptr[old]; p < Mrows.ptr[old + 1]; ++p)
                row.emplace_back(ocol.old2new[Mrows.idx[p]], Mrows.val[p]);
            sort_row(row);
            int64_t q = st.ptr[r];
            for (auto& e : row) { st.idx[q] = e.first; st.val[q] = e.second; ++q; }
        }
    });
    return csr_to_sell(np, ocol.n_pad, st, sort_sigma, lpr);
}

// Block-ordered CSR of part of a blocked level's operator in device numbering (the specification of the device builders
// setup_kernels.hip.hpp::csr_fill / csr_fill_plain): the kept entries of device row r, ascending device column, at
// [ptr[r], ptr[r + 1]).  part 2: the entries that LEAVE r's block (device column; mid[r] = end of the row -- csr_fill writes
// in-block entries after it when its filter keeps them; here there are none).  Parts of the unpadded block sweep:
// 3 = "explicit" (entries leaving the block + in-block entries with a later device column; device column),
// 4 = "lower" (in-block entries with an earlier device column; column local to the block).
struct BlockCsrHost {
    std::vector<int> ptr, mid;      // n_pad + 1, n_pad
    RawVec<int> col;
    RawVec<double> val;
    int max_block_entries = 0;
};

inline void build_operator_blockcsr(const Compressed& A, const LevelOrdering& ord, BlockCsrHost& out, int part = 2) {
    const int np = ord.n_pad;
    out.ptr.assign((size_t)np + 1, 0);
    out.mid.assign(np, 0);
    std::vector<int> blk_of(np, 0);
    for (int b = 0; b < ord.n_blocks(); ++b) for (int r = ord.blk_begin[b]; r < ord.blk_begin[b + 1]; ++r) blk_of[r] = b;
    auto keep = [part](int r, int c, int r0, int r1) {
        const bool inside = c >= r0 && c < r1;
        return part == 2 ? !inside : (part == 3 ? (!inside || c > r) : (inside && c < r));
    };
    parallel_ranges(np, hw_threads(), [&](int lo, int hi, int) {
        for (int r = lo; r < hi; ++r) {
            const int old = ord.new2old[r];
            int n = 0;
            if (old >= 0) {
                const int b = blk_of[r], r0 = ord.blk_begin[b], r1 = ord.blk_begin[b + 1];
                for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) { const int c = ord.old2new[A.idx[p]]; n += A.idx[p] != old && keep(r, c, r0, r1); }
            }
            out.ptr[r + 1] = n;
        }
    });
    for (int r = 0; r < np; ++r) out.ptr[r + 1] += out.ptr[r];
    out.col.resize(out.ptr[np]); out.val.resize(out.ptr[np]);
    const int nb = ord.n_blocks();
    parallel_ranges(nb, hw_threads(), [&](int lo, int hi, int) {
        std::vector<std::pair<int, double>> e;
        for (int b = lo; b < hi; ++b) {
            const int r0 = ord.blk_begin[b], r1 = ord.blk_begin[b + 1];
            for (int r = r0; r < r1; ++r) {
                const int old = ord.new2old[r];
                e.clear();
                if (old >= 0) for (int p = A.ptr[old]; p < A.ptr[old + 1]; ++p) if (A.idx[p] != old) e.emplace_back(ord.old2new[A.idx[p]], A.val[p]);
                std::stable_sort(e.begin(), e.end(), [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
                int q = out.ptr[r];
                for (auto& t : e) if (keep(r, t.first, r0, r1)) { out.col[q] = part == 4 ? t.first - r0 : t.first; out.val[q] = t.second; ++q; }
                out.mid[r] = q;
            }
        }
    }, 64);
    out.max_block_entries = 0;
    for (int b = 0; b < nb; ++b) out.max_block_entries = std::max(out.max_block_entries, out.ptr[ord.blk_begin[b + 1]] - out.ptr[ord.blk_begin[b]]);
}

// Rows of U (fine-row major) from its CSC storage, threaded: per-row counts with atomics, prefix sum, scatter.
// Entry order inside a row is arbitrary (the builders above sort every row by device column anyway).
inline Compressed transpose_parallel(const Compressed& a) {
    Compressed t;
    t.n_outer = a.n_inner; t.n_inner = a.n_outer;
    const int nnz = a.nnz();
    std::unique_ptr<std::atomic<int>[]> cnt(new std::atomic<int>[(size_t)a.n_inner + 1]);
    parallel_ranges(a.n_inner + 1, hw_threads(), [&](int lo, int hi, int) { for (int i = lo; i < hi; ++i) cnt[i].store(0, std::memory_order_relaxed); });
    parallel_ranges(a.n_outer, hw_threads(), [&](int lo, int hi, int) {
        for (int j = lo; j < hi; ++j)
            for (int p = a.ptr[j]; p < a.ptr[j + 1]; ++p) cnt[a.idx[p]].fetch_add(1, std::memory_order_relaxed);
    });
    t.ptr.assign((size_t)a.n_inner + 1, 0);
    for (int i = 0; i < a.n_inner; ++i) t.ptr[i + 1] = t.ptr[i] + cnt[i].load(std::memory_order_relaxed);
    parallel_ranges(a.n_inner, hw_threads(), [&](int lo, int hi, int) { for (int i = lo; i < hi; ++i) cnt[i].store(t.ptr[i], std::memory_order_relaxed); });
    t.idx.resize(nnz); t.val.resize(nnz);
    parallel_ranges(a.n_outer, hw_threads(), [&](int lo, int hi, int) {
        for (int j = lo; j < hi; ++j)
            for (int p = a.ptr[j]; p < a.ptr[j + 1]; ++p) {
                int q = cnt[a.idx[p]].fetch_add(1, std::memory_order_relaxed);
                t.idx[q] = j; t.val[q] = a.val[p];
            }
    });
    return t;
}

}  // namespace gmg
