// setup_kernels.hip.hpp -- device-side construction of the SELL-64 layouts (once per system).
//
// Input: a matrix in the reference's compressed storage (natural numbering, as uploaded) plus the level orderings
// computed on the host (host_plan.hpp).  Output: exactly the arrays host_plan.hpp::csr_to_sell produces -- the host
// builder stays the specification, tests/test_gpu_setup.py compares the two bit for bit.  These kernels replace ~0.5 s
// of latency-bound host loops at 3 M vertices by a few ms of GPU time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmgs {

constexpr int kMaxRow = 96;          // longest row (kept entries) the device builder takes: host planner beyond

// mode: 0 = every entry, 1 = only entries whose column lies in the row's block, 2 = only entries leaving the block,
// 3 = "explicit" part of the unpadded block sweep (entries leaving the block + in-block entries with a LATER device column),
// 4 = "lower" part (in-block entries with an EARLIER device column; local column like mode 1)
struct RowFilter {
    const int* new2old_row;      // [n_rows_pad]  -1 = padding row
    const int* old2new_col;      // [n_cols]
    const int* blk_of_row;       // [n_pad] device row -> block (modes 1/2), else null
    const int* blk_begin;        // block -> first device row (mode 1: local column = device column - blk_begin)
    int mode;
    int drop_diag;               // skip entries with natural column == natural row (and report them through diag)
};

__device__ __forceinline__ bool keep_entry(const RowFilter& f, int dev_row, int old_row, int old_col, int& out_col) {
    if (f.drop_diag && old_col == old_row) return false;
    const int nc = f.old2new_col[old_col];
    if (f.mode == 0) { out_col = nc; return true; }
    const bool inside = f.blk_of_row[nc] == f.blk_of_row[dev_row];
    if (f.mode == 1) { out_col = nc - f.blk_begin[f.blk_of_row[dev_row]]; return inside; }
    if (f.mode == 4) { out_col = nc - f.blk_begin[f.blk_of_row[dev_row]]; return inside && nc < dev_row; }
    out_col = nc;
    if (f.mode == 3) return !inside || nc > dev_row;
    return !inside;
}

// Row masks of a partitioned set-up (one process per GPU; engine_part.hip.hpp): out[r] = new2old[r] for the rows this rank owns, -1 ("padding
// row": no entries, unit diagonal) for everybody else's -- the layout builders below then give the other ranks' rows zero-width slices / empty
// chunks, in the global numbering, so that the cycle's kernels run unchanged on this rank's slices and blocks.
// colour-major level: every colour class is cut into `world` equal contiguous pieces, rank p owns piece p of every colour (p2p_owner)
__global__ void mask_rows_by_colour(const int* __restrict__ new2old, int n_pad, const int* __restrict__ color_begin, int n_colors, int world, int rank,
                                    int* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    int c = 0;
    while (c + 1 < n_colors && r >= color_begin[c + 1]) ++c;
    const int piece = (color_begin[c + 1] - color_begin[c]) / world;
    const int owner = piece > 0 ? (r - color_begin[c]) / piece : 0;
    out[r] = owner == rank ? new2old[r] : -1;
}
// blocked level (64-row blocks, block b = rows 64 b ..): a block belongs to blk_owner[b]
__global__ void mask_rows_by_block(const int* __restrict__ new2old, int n_pad, const int* __restrict__ blk_owner, int rank, int* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    out[r] = blk_owner[r >> 6] == rank ? new2old[r] : -1;
}

// len[i] = number of kept entries of the row at slice position i (order[i] if given, else i)
// (pbeg[row], pend[row]) delimit a row's entries: pend = pbeg + 1 for ordinary compressed storage.
__global__ void row_lengths(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx, RowFilter f,
                            const int* __restrict__ order, int n_rows_pad, int* __restrict__ len, int* __restrict__ err_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows_pad) return;
    const int r = order ? order[i] : i;
    const int old = f.new2old_row[r];
    int n = 0;
    if (old >= 0) {
        for (int p = pbeg[old]; p < pend[old]; ++p) { int c; if (keep_entry(f, r, old, idx[p], c)) ++n; }
        if (n > kMaxRow) atomicExch(err_flag, 1);
    }
    len[i] = n;
}

// widths[s] = 64 * max over the slice's rows of ceil(len / lpr)
__global__ void slice_widths(const int* __restrict__ len, int lpr, int n_slices, int64_t* __restrict__ widths) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slices) return;
    const int rps = 64 / lpr;
    int w = 0;
    for (int l = 0; l < rps; ++l) { int v = (len[s * rps + l] + lpr - 1) / lpr; w = v > w ? v : w; }
    widths[s] = (int64_t)w * 64;
}

// ---- a row's kept entries, in device-column order ---------------------------------------------------------------------------
// The fill kernels below give one thread a row: gather the kept entries, put them in device-column order, write them out.  The
// first version kept them in a private array of kMaxRow entries and insertion-sorted it: 1 152 bytes of scratch per thread that
// no cache holds for 3 M threads -- `sell_fill` moved 5.2 GB for a 252 MB matrix (rocprofv3 FETCH_SIZE / WRITE_SIZE,
// profiles/r02).  Now nothing is sorted and nothing leaves the registers: pass 1 collects the kept DEVICE COLUMNS of the row in
// RowCols<CAP> (append = a chain of selects with compile-time indices: no dynamic register indexing, so no scratch); pass 2 walks
// the row again (its index / map entries are in the cache) and writes every kept entry straight to its place, its RANK = the number
// of collected columns smaller than its own (CAP compares, unrolled).  CAP in {8, 32, 96} is chosen per WAVEFRONT by its longest
// stored row (wave-uniform branch).  Same entries at the same places as before: the output is bit-identical.
template <int CAP>
struct RowCols {
    int c[CAP];
    int n;
};

template <int CAP>
__device__ __forceinline__ void cols_append(RowCols<CAP>& R, int col) {
#pragma unroll
    for (int j = 0; j < CAP; ++j) R.c[j] = R.n == j ? col : R.c[j];
    ++R.n;
}

// pass 1: the kept device columns of natural row `old` (device row r); dg / has_diag: the dropped diagonal entry.  false (and the
// error flag) when the row keeps more than CAP entries (CAP = kMaxRow only: the caller picked CAP >= the stored length otherwise).
template <int CAP>
__device__ __forceinline__ bool cols_gather(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx,
                                            const double* __restrict__ val, const RowFilter& f, int r, int old, RowCols<CAP>& R, double& dg,
                                            bool& has_diag, int* __restrict__ err_flag) {
    R.n = 0;
#pragma unroll
    for (int j = 0; j < CAP; ++j) R.c[j] = 0x7fffffff;        // unused slots: larger than every column, never counted by a rank
    if (old < 0) return true;
    for (int p = pbeg[old]; p < pend[old]; ++p) {
        const int oc = idx[p];
        if (f.drop_diag && oc == old) { dg = val[p]; has_diag = true; continue; }
        int c;
        if (!keep_entry(f, r, old, oc, c)) continue;
        if (R.n >= CAP) { atomicExch(err_flag, 1); return false; }
        cols_append<CAP>(R, c);
    }
    return true;
}

template <int CAP>
__device__ __forceinline__ int cols_rank(const RowCols<CAP>& R, int col) {
    int k = 0;
#pragma unroll
    for (int j = 0; j < CAP; ++j) k += R.c[j] < col ? 1 : 0;
    return k;
}

// longest stored row of the wavefront (an upper bound of the kept entries): picks the register capacity
__device__ __forceinline__ int wave_max_raw_len(const int* __restrict__ pbeg, const int* __restrict__ pend, int old) {
    int len = old >= 0 ? pend[old] - pbeg[old] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off, 64));
    return len;
}

// One thread per slice row position: the kept entries in device-column order into the slice, padding behind them.
// diag (optional) receives the dropped diagonal entry (1.0 for padding rows).
template <class ColT, int CAP>
__device__ __forceinline__ void sell_fill_row(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx,
                                              const double* __restrict__ val, const RowFilter& f, int lpr, int r, int old, int s, int l,
                                              const int64_t* __restrict__ slice_ptr, ColT* __restrict__ col, double* __restrict__ out_val,
                                              double* __restrict__ diag, int* __restrict__ err_flag, int* __restrict__ src, int* __restrict__ diag_src) {
    RowCols<CAP> R;
    double dg = 1.0;
    bool has_diag = old < 0;
    const bool ok = cols_gather<CAP>(pbeg, pend, idx, val, f, r, old, R, dg, has_diag, err_flag);
    if (diag) {
        if (ok && f.drop_diag && (!has_diag || dg == 0.0)) atomicExch(err_flag, 2);      // missing / zero diagonal
        diag[r] = dg;
    }
    if (diag_src) diag_src[r] = -1;
    const int64_t base = slice_ptr[s];
    const int w = (int)((slice_ptr[s + 1] - base) >> 6);
    auto at = [&](int e) { return base + (int64_t)(e / lpr) * 64 + l * lpr + (e % lpr); };
    if (old >= 0 && ok)
        for (int p = pbeg[old]; p < pend[old]; ++p) {
            const int oc = idx[p];
            if (f.drop_diag && oc == old) { if (diag_src) diag_src[r] = p; continue; }
            int c;
            if (!keep_entry(f, r, old, oc, c)) continue;
            const int64_t q = at(cols_rank<CAP>(R, c));
            col[q] = (ColT)c; out_val[q] = val[p];
            if (src) src[q] = p;
        }
    for (int e = ok ? R.n : 0; e < w * lpr; ++e) { const int64_t q = at(e); col[q] = (ColT)0; out_val[q] = 0.0; if (src) src[q] = -1; }
}

template <class ColT>
__global__ void sell_fill(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx,
                          const double* __restrict__ val, RowFilter f,
                          const int* __restrict__ order, int lpr, int n_rows_pad, const int64_t* __restrict__ slice_ptr,
                          ColT* __restrict__ col, double* __restrict__ out_val, double* __restrict__ diag, int* __restrict__ err_flag,
                          int* __restrict__ src = nullptr, int* __restrict__ diag_src = nullptr) {
    // src / diag_src (optional): where every slot / diagonal entry came from in `val` (-1: padding) -- what a system with the same sparsity pattern
    // needs to refresh the values by a plain gather (refill_values) instead of ranking the columns of every row again
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_rows_pad;
    const int rps = 64 / lpr;
    const int s = live ? i / rps : 0, l = live ? i % rps : 0;
    const int r = live ? (order ? order[i] : i) : 0;
    const int old = live ? f.new2old_row[r] : -1;
    const int wmax = wave_max_raw_len(pbeg, pend, old);          // (all lanes take part in the shuffles)
    if (!live) return;
    if (wmax <= 8) sell_fill_row<ColT, 8>(pbeg, pend, idx, val, f, lpr, r, old, s, l, slice_ptr, col, out_val, diag, err_flag, src, diag_src);
    else if (wmax <= 32) sell_fill_row<ColT, 32>(pbeg, pend, idx, val, f, lpr, r, old, s, l, slice_ptr, col, out_val, diag, err_flag, src, diag_src);
    else sell_fill_row<ColT, kMaxRow>(pbeg, pend, idx, val, f, lpr, r, old, s, l, slice_ptr, col, out_val, diag, err_flag, src, diag_src);
}

// Block-CSR of a big blocked level (kernels.hip.hpp::gs_blockcsr): off-diagonal entries of device row r at
// [row_ptr[r], row_ptr[r + 1]), device column numbers, the entries that leave the row's block first (ascending column),
// then the in-block ones (ascending column) from row_mid[r] on.  row_ptr comes from row_lengths (mode 0) + a prefix sum.
template <int CAP>
__device__ __forceinline__ void csr_fill_row(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx,
                                             const double* __restrict__ val, const RowFilter& f, int r, int old, int r0, int r1,
                                             const int* __restrict__ row_ptr, int* __restrict__ row_mid, int* __restrict__ col,
                                             double* __restrict__ out_val, int* __restrict__ err_flag) {
    RowCols<CAP> R;
    double dg = 1.0; bool hd = false;
    const int q0 = row_ptr[r];
    if (!cols_gather<CAP>(pbeg, pend, idx, val, f, r, old, R, dg, hd, err_flag)) { row_mid[r] = q0; return; }
    int n_out = 0;                                              // entries that leave the block: they come first
#pragma unroll
    for (int j = 0; j < CAP; ++j) n_out += (j < R.n && (R.c[j] < r0 || R.c[j] >= r1)) ? 1 : 0;
    row_mid[r] = q0 + n_out;
    if (old < 0) return;
    for (int p = pbeg[old]; p < pend[old]; ++p) {
        int c;
        if (!keep_entry(f, r, old, idx[p], c)) continue;
        const bool inside = c >= r0 && c < r1;
        int k = 0;                                              // rank among the entries of its own group
#pragma unroll
        for (int j = 0; j < CAP; ++j) k += (R.c[j] < c && ((R.c[j] >= r0 && R.c[j] < r1) == inside)) ? 1 : 0;
        const int q = q0 + (inside ? n_out : 0) + k;
        col[q] = c; out_val[q] = val[p];
    }
}

__global__ void csr_fill(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx, const double* __restrict__ val,
                         RowFilter f, const int* __restrict__ blk_of_row, const int* __restrict__ blk_begin, int n_rows_pad,
                         const int* __restrict__ row_ptr, int* __restrict__ row_mid, int* __restrict__ col, double* __restrict__ out_val,
                         int* __restrict__ err_flag) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < n_rows_pad;
    const int old = live ? f.new2old_row[r] : -1;
    const int wmax = wave_max_raw_len(pbeg, pend, old);
    if (!live) return;
    const int b = blk_of_row[r];
    const int r0 = blk_begin[b], r1 = blk_begin[b + 1];
    if (wmax <= 8) csr_fill_row<8>(pbeg, pend, idx, val, f, r, old, r0, r1, row_ptr, row_mid, col, out_val, err_flag);
    else if (wmax <= 32) csr_fill_row<32>(pbeg, pend, idx, val, f, r, old, r0, r1, row_ptr, row_mid, col, out_val, err_flag);
    else csr_fill_row<kMaxRow>(pbeg, pend, idx, val, f, r, old, r0, r1, row_ptr, row_mid, col, out_val, err_flag);
}

// Plain block-ordered CSR of the entries the filter keeps (the in-block operator of the entry-parallel sweep, 16-bit local
// columns): row r's kept entries, ascending column, at [row_ptr[r], row_ptr[r + 1]).
template <class ColT, int CAP>
__device__ __forceinline__ void csr_fill_plain_row(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx,
                                                   const double* __restrict__ val, const RowFilter& f, int r, int old, const int* __restrict__ row_ptr,
                                                   ColT* __restrict__ col, double* __restrict__ out_val, int* __restrict__ err_flag, int* __restrict__ src) {
    RowCols<CAP> R;
    double dg = 1.0; bool hd = false;
    if (!cols_gather<CAP>(pbeg, pend, idx, val, f, r, old, R, dg, hd, err_flag)) return;
    const int q0 = row_ptr[r];
    for (int p = pbeg[old]; p < pend[old]; ++p) {
        int c;
        if (!keep_entry(f, r, old, idx[p], c)) continue;
        const int q = q0 + cols_rank<CAP>(R, c);
        col[q] = (ColT)c; out_val[q] = val[p];
        if (src) src[q] = p;
    }
}

template <class ColT>
__global__ void csr_fill_plain(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx, const double* __restrict__ val,
                               RowFilter f, int n_rows_pad, const int* __restrict__ row_ptr, ColT* __restrict__ col, double* __restrict__ out_val,
                               int* __restrict__ err_flag, int* __restrict__ src = nullptr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < n_rows_pad;
    const int old = live ? f.new2old_row[r] : -1;
    const int wmax = wave_max_raw_len(pbeg, pend, old);
    if (!live || old < 0) return;
    if (wmax <= 8) csr_fill_plain_row<ColT, 8>(pbeg, pend, idx, val, f, r, old, row_ptr, col, out_val, err_flag, src);
    else if (wmax <= 32) csr_fill_plain_row<ColT, 32>(pbeg, pend, idx, val, f, r, old, row_ptr, col, out_val, err_flag, src);
    else csr_fill_plain_row<ColT, kMaxRow>(pbeg, pend, idx, val, f, r, old, row_ptr, col, out_val, err_flag, src);
}

// Values-only refresh of a layout through its source map (sell_fill / csr_fill_plain with `src`): out[q] = val[src[q]], 0 for padding slots.
__global__ void refill_values(const int* __restrict__ src, const double* __restrict__ val, int64_t n, double* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int p = __builtin_nontemporal_load(src + q);
    out[q] = p >= 0 ? val[p] : 0.0;
}
// ... and of the diagonal: diag[r] = val[diag_src[r]] (1 for padding rows); a zero diagonal entry raises the error flag (2) like sell_fill
__global__ void refill_diag(const int* __restrict__ diag_src, const int* __restrict__ new2old, const double* __restrict__ val, int n_pad, double* __restrict__ diag,
                            int* __restrict__ err_flag) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    const int p = diag_src[r];
    double dg = 1.0;
    if (p >= 0) { dg = val[p]; if (dg == 0.0) atomicExch(err_flag, 2); }
    else if (new2old[r] >= 0) atomicExch(err_flag, 2);
    diag[r] = dg;
}

// 16-bit column codes of a SELL operator (kernels.hip.hpp, "16-bit column codes"): one wavefront per slice, NW windows of 65536 / NW
// columns.  The windows of a slice are chosen greedily in ascending order (base_0 = the smallest column, base_k = the smallest column
// at or beyond base_{k-1} + span), which covers any column set with the fewest windows of that length; padding entries get code 0 = the
// slice's first base, a valid index that is multiplied by 0 like column 0 before.  What is padding: for the operator (a_ptr != null)
// entry j of a row when j >= the row's stored off-diagonal entries, known from the source matrix -- never judged by a value, which a
// values-only refresh may change; for a transfer (a_ptr == null; its values never change after the layout) an entry whose value is 0,
// which contributes 0 wherever it points.  A slice that needs more than NW windows (tests: also every test_fail-th slice, or the first -test_fail slices) gets -1 as its
// first base -- the kernels then read its 32-bit indices -- and is counted in fail[0]; fail[1] = 1 + the index of the last such slice.
template <int NW>
__global__ __launch_bounds__(256) void compress_cols(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col, const int* __restrict__ a_ptr,
                                                     const int* __restrict__ new2old, const double* __restrict__ val, int n_slices, int test_fail,
                                                     unsigned* __restrict__ col16, int* __restrict__ win_base, int* __restrict__ fail,
                                                     int w_expected = -1, int* __restrict__ other_width = nullptr) {
    constexpr int kSpan = 65536 / NW;
    constexpr int kDbits = NW == 8 ? 13 : 11;
    static_assert(NW == 8 || NW == 32, "8 windows of 8192 or 32 windows of 2048");
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (s >= n_slices) return;
    const int64_t p0 = slice_ptr[s];
    const int w = (int)((slice_ptr[s + 1] - p0) >> 6);
    // (are all slices w_expected wide?  then the kernels need no slice pointers: DevSell::uniform_w.  Counted until the answer is no)
    if (other_width && w != w_expected && lane == 0 && __hip_atomic_load(other_width, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicAdd(other_width, 1);
    int len = 0;
    if (a_ptr) { const int old = new2old[s * 64 + lane]; len = old >= 0 ? a_ptr[old + 1] - a_ptr[old] - 1 : 0; }
    auto real = [&](int j) { return a_ptr ? j < len : val[p0 + (int64_t)j * 64 + lane] != 0.0; };
    constexpr int kNone = 0x7fffffff;
    int base[NW];
    int lo = 0;                                                  // columns below lo are covered
    bool more = true;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        int m = kNone;
        if (more)
            for (int j = 0; j < w; ++j) {
                const int c = col[p0 + (int64_t)j * 64 + lane];
                if (c >= lo && c < m && real(j)) m = c;
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = min(m, __shfl_xor(m, off, 64));
        if (m == kNone) { more = false; base[k] = k ? base[k - 1] : 0; }
        else { base[k] = m; lo = m + kSpan; }
    }
    // anything left beyond the last window?
    bool bad = false;
    if (more)
        for (int j = 0; j < w; ++j) { const int c = col[p0 + (int64_t)j * 64 + lane]; if (c >= lo && real(j)) bad = true; }
    bad = __ballot(bad) != 0ull || (test_fail > 0 && s % test_fail == 1) || (test_fail < 0 && s < -test_fail);
    unsigned word = 0;
    for (int j = 0; j < w; ++j) {
        const int c = col[p0 + (int64_t)j * 64 + lane];
        unsigned code = 0;
        if (real(j) && !bad) {
            int k = 0;
#pragma unroll
            for (int q = 1; q < NW; ++q) k += (base[q] > base[q - 1] && c >= base[q]) ? 1 : 0;
            int b = base[0];
#pragma unroll
            for (int q = 1; q < NW; ++q) b = k == q ? base[q] : b;
            code = ((unsigned)k << kDbits) | ((unsigned)(c - b) & (unsigned)(kSpan - 1));
        }
        word = (j & 1) ? (word | (code << 16)) : code;
        if ((j & 1) || j == w - 1) col16[p0 + (int64_t)(j >> 1) * 64 + lane] = word;
    }
    if (bad && lane == 0) { atomicAdd(fail, 1); atomicMax(fail + 1, s + 1); }      // how many, and the end of the last one
    if (lane < NW) {
        int b = base[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) b = lane == q ? base[q] : b;
        win_base[(int64_t)s * NW + lane] = (bad && lane == 0) ? -1 : b;
    }
}

// max over the blocks of their entry count (LDS capacity the sweep kernel needs)
__global__ void block_entry_max(const int* __restrict__ blk_begin, int n_blocks, const int* __restrict__ row_ptr, int* __restrict__ out_max) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    atomicMax(out_max, row_ptr[blk_begin[b + 1]] - row_ptr[blk_begin[b]]);
}

// ---- exclusive prefix sums on the device (slice pointers, RAP row pointers): out[0] = 0, out[i + 1] = in[0] + .. + in[i].
// Three small launches: per-tile sums, one block scanning the tile sums, per-tile rescan with the tile offset.
constexpr int kScanTile = 2048;     // items per 256-thread block (8 per thread)

template <class TIn, class TOut>
__global__ __launch_bounds__(256) void scan_tile_sums(const TIn* __restrict__ in, int n, TOut* __restrict__ tile_sum) {
    __shared__ TOut red[256];
    const int base = blockIdx.x * kScanTile + threadIdx.x * 8;
    TOut s = 0;
    for (int j = 0; j < 8; ++j) if (base + j < n) s += (TOut)in[base + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0];
}

// one block: tile_sum[t] -> exclusive offsets (in place); the grand total goes to *total
template <class TOut>
__global__ __launch_bounds__(1024) void scan_tile_offsets(TOut* __restrict__ tile_sum, int n_tiles, TOut* __restrict__ total) {
    __shared__ TOut buf[1024];
    __shared__ TOut carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const TOut v = i < n_tiles ? tile_sum[i] : (TOut)0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            TOut t = (int)threadIdx.x >= off ? buf[threadIdx.x - off] : (TOut)0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_tiles) tile_sum[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

template <class TIn, class TOut>
__global__ __launch_bounds__(256) void scan_tile_apply(const TIn* __restrict__ in, int n, const TOut* __restrict__ tile_off, TOut* __restrict__ out) {
    __shared__ TOut part[256];
    const int base = blockIdx.x * kScanTile + threadIdx.x * 8;
    TOut v[8];
    TOut s = 0;
    for (int j = 0; j < 8; ++j) { v[j] = base + j < n ? (TOut)in[base + j] : (TOut)0; s += v[j]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        TOut t = (int)threadIdx.x >= off ? part[threadIdx.x - off] : (TOut)0;
        __syncthreads();
        part[threadIdx.x] += t;
        __syncthreads();
    }
    TOut run = tile_off[blockIdx.x] + part[threadIdx.x] - s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
    for (int j = 0; j < 8; ++j) { run += v[j]; if (base + j < n) out[base + j + 1] = run; }
}

// Slice order of the restriction rows: inside every window of `sigma` device rows, rows sorted by descending length,
// ties in row order (what std::stable_sort gives the host planner).  One block per window, bitonic sort in LDS of the
// unique keys (32767 - len) << 16 | position.  len = entries of the U column of the row's natural index.
constexpr int kWindowSortMax = 4096;
__global__ __launch_bounds__(256) void window_order_by_length(const int* __restrict__ u_cptr, const int* __restrict__ new2old, int np, int sigma,
                                                              int pow2, int* __restrict__ order) {
    __shared__ int key[kWindowSortMax];
    const int w0 = blockIdx.x * sigma;
    const int cnt = min(sigma, np - w0);
    for (int i = threadIdx.x; i < pow2; i += blockDim.x) {
        int k = 0x7fffffff;
        if (i < cnt) {
            const int old = new2old[w0 + i];
            const int len = old >= 0 ? min(u_cptr[old + 1] - u_cptr[old], 32767) : 0;
            k = ((32767 - len) << 16) | i;
        }
        key[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= pow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < pow2; i += blockDim.x) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool up = (i & size) == 0;
                    const int a = key[i], b = key[j];
                    if ((a > b) == up) { key[i] = b; key[j] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) order[w0 + i] = w0 + (key[i] & 0xffff);
}

// The sparsity pattern permuted symmetrically: new row r = old row perm[r], columns renumbered by inv (old -> new).
// Two launches around a prefix sum; column order inside a row is kept (the colouring that consumes it does not care).
__global__ void perm_row_lengths(const int* __restrict__ ptr, const int* __restrict__ perm, int n, int* __restrict__ len) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int old = perm[r];
    len[r] = ptr[old + 1] - ptr[old];
}

__global__ void perm_fill(const int* __restrict__ ptr, const int* __restrict__ idx, const int* __restrict__ perm, const int* __restrict__ inv,
                          const int* __restrict__ pptr, int n, int* __restrict__ pidx) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int old = perm[r];
    const int b = ptr[old], e = ptr[old + 1], o = pptr[r];
    for (int q = b; q < e; ++q) pidx[o + (q - b)] = inv[idx[q]];
}

// How well do the x-gathers of 64 consecutive rows coalesce under a candidate base order of the level-0 points?  One wavefront per
// sampled window: lane l stands for the point at position start + 4 l of the order (every fourth one: the rows of one colour
// class), sorts the positions of its row's columns, and for every entry slot j the wave counts the distinct 128-byte lines
// (16 doubles) its 64 j-th gathers touch -- what a gather instruction of the sweep / residual kernels pays in the vector cache.
// out[0] += distinct lines, out[1] += entries (integers: the sums do not depend on the order of the atomics).
constexpr int kScoreMaxRow = 32;
__global__ __launch_bounds__(256) void order_gather_score(const int* __restrict__ ptr, const int* __restrict__ idx, const int* __restrict__ order,
                                                          const int* __restrict__ inv, int n, int n_win, unsigned long long* __restrict__ out) {
    __shared__ int lines[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave;
    if (w >= n_win || n < 512) return;
    const int start = (int)((long long)w * (n - 256) / n_win);
    const int v = order[start + 4 * lane];
    const int b = ptr[v];
    const int len = min(ptr[v + 1] - b, kScoreMaxRow);
    int p[kScoreMaxRow];
    for (int q = 0; q < len; ++q) {                        // insertion sort of the column positions
        const int c = inv[idx[b + q]];
        int k = q;
        while (k > 0 && p[k - 1] > c) { p[k] = p[k - 1]; --k; }
        p[k] = c;
    }
    int maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
    unsigned distinct = 0, entries = 0;
    for (int j = 0; j < maxlen; ++j) {
        const bool has = j < len;
        const int line = has ? p[j] >> 4 : -1;
        lines[wave][lane] = line;
        __builtin_amdgcn_wave_barrier();
        bool first = has;
        for (int m = 0; m < lane && first; ++m) first = lines[wave][m] != line;
        __builtin_amdgcn_wave_barrier();
        distinct += first ? 1u : 0u;
        entries += has ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { distinct += __shfl_xor(distinct, off, 64); entries += __shfl_xor(entries, off, 64); }
    if (lane == 0) { atomicAdd(out, (unsigned long long)distinct); atomicAdd(out + 1, (unsigned long long)entries); }
}

// row -> block map of a blocked ordering (blk_begin: n_blocks + 1 device rows)
__global__ void block_of_rows(const int* __restrict__ blk_begin, int n_blocks, int* __restrict__ blk_of_row) {
    const int b = blockIdx.x;
    if (b >= n_blocks) return;
    for (int r = blk_begin[b] + threadIdx.x; r < blk_begin[b + 1]; r += blockDim.x) blk_of_row[r] = b;
}

// Rows of U (<= 3 entries each, gravomg/src/multigrid_solver.cpp:371-373) from its CSC storage: ELL-3 staging written
// as a CSR with a fixed stride of 3 (ptr3[i] = 3 i, unused slots hold column -1).
__global__ void ell3_from_csc(const int* __restrict__ colptr, const int* __restrict__ rowidx, const double* __restrict__ val, int n_coarse,
                              int* __restrict__ cnt, int* __restrict__ ecol, double* __restrict__ eval, int* __restrict__ err_flag) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_coarse) return;
    for (int p = colptr[c]; p < colptr[c + 1]; ++p) {
        const int i = rowidx[p];
        const int slot = atomicAdd(&cnt[i], 1);
        if (slot >= 3) { atomicExch(err_flag, 3); continue; }
        ecol[i * 3 + slot] = c;
        eval[i * 3 + slot] = val[p];
    }
}

// row delimiters of the ELL-3 staging for row_lengths / sell_fill: begin = 3 i, end = 3 i + cnt[i]
__global__ void ell3_ptr(const int* __restrict__ cnt, int n, int* __restrict__ pbeg, int* __restrict__ pend) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pbeg[i] = 3 * i;
    pend[i] = 3 * i + cnt[i];
}

// Sort the (<= 3) entries of every ELL-3 row by column (ell3_from_csc fills the slots in arbitrary order).
__global__ void ell3_sort(const int* __restrict__ cnt, int n, int* __restrict__ ecol, double* __restrict__ eval) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int m = cnt[i] < 3 ? cnt[i] : 3;
    int c[3]; double v[3];
    for (int s = 0; s < 3; ++s) { c[s] = s < m ? ecol[3 * i + s] : 0x7fffffff; v[s] = s < m ? eval[3 * i + s] : 0.0; }
#define GMG_CSWAP(a, b) if (c[a] > c[b]) { int tc = c[a]; c[a] = c[b]; c[b] = tc; double tv = v[a]; v[a] = v[b]; v[b] = tv; }
    GMG_CSWAP(0, 1) GMG_CSWAP(1, 2) GMG_CSWAP(0, 1)
#undef GMG_CSWAP
    for (int s = 0; s < m; ++s) { ecol[3 * i + s] = c[s]; eval[3 * i + s] = v[s]; }
}

// Galerkin product Ac = U^T A U (gravomg/src/multigrid_solver.cpp:1387-1392), one wavefront (= one 64-thread block) per
// coarse row p.  Inputs: A in compressed storage (symmetric: rows == columns), U by coarse column (its CSC storage) and
// by fine row (ELL-3, sorted).  Step 1: the distinct output columns q of row p are collected in an LDS hash set (order
// independent) and sorted.  Step 2 (PASS 1): every lane owns one output column and the whole wave walks the triples
//     for i in column p of U (ascending) / for j in row i of A (as stored) / for c in row j of U (ascending)
// in the SAME order as the host implementation (host_sparse.hpp::galerkin_rap), adding w = (u_ip a_ij) * u_jq to its
// accumulator when q matches -- with separately rounded multiply and add, so the result is bitwise the host's.
// PASS 0 only counts the distinct columns (row lengths for the prefix sum); PASS 2 is the numeric step alone, for a
// product whose pattern (c_ptr, c_idx) is already known.
constexpr int kRapSlots = 192;     // (column, product) list of one chunk of children in LDS: 3 per entry of A
constexpr int kRapSet = 256;      // hash-set capacity per coarse row (rows with more distinct columns -> host fallback)
template <int PASS>
__global__ __launch_bounds__(64) void rap_rows(const int* __restrict__ a_ptr, const int* __restrict__ a_idx, const double* __restrict__ a_val,
                                               const int* __restrict__ u_cptr, const int* __restrict__ u_ridx, const double* __restrict__ u_cval,
                                               const int* __restrict__ ur_cnt, const int* __restrict__ ur_col, const double* __restrict__ ur_val,
                                               int n_coarse, const int* __restrict__ c_ptr, int* __restrict__ c_cnt, int* __restrict__ c_idx,
                                               double* __restrict__ c_val, int* __restrict__ err_flag, int p_begin = 0,
                                               const int* __restrict__ row_list = nullptr) {
#pragma clang fp contract(off)      // multiply and add rounded separately, like the host implementation (no FMA)
    __shared__ int keys[kRapSet];
    // (p_begin / row_list: a range of coarse rows, or of a list of them -- the numeric pass pipelined behind the upload of the values, engine.hip)
    if (p_begin + (int)blockIdx.x >= n_coarse) return;
    const int p = row_list ? row_list[p_begin + (int)blockIdx.x] : p_begin + (int)blockIdx.x;
    const int lane = threadIdx.x;
    for (int s = lane; s < kRapSet; s += 64) keys[s] = 0x7fffffff;
    __syncthreads();
    const int ub = u_cptr[p], ue = u_cptr[p + 1];
    int cnt = 0;
    if (PASS == 2) {
        // the row's sorted columns are known (same sparsity pattern as the product c_idx was computed from): no hash set,
        // no sort -- straight to the numeric step
        const int o0 = c_ptr[p];
        cnt = c_ptr[p + 1] - o0;
        for (int s = lane; s < cnt; s += 64) keys[s] = c_idx[o0 + s];
        __syncthreads();
    } else {
    // ---- step 1: set of distinct coarse columns
    for (int t = ub + lane; t < ue; t += 64) {
        const int i = u_ridx[t];
        for (int b = a_ptr[i]; b < a_ptr[i + 1]; ++b) {
            const int j = a_idx[b];
            const int m = ur_cnt[j] < 3 ? ur_cnt[j] : 3;
            for (int s = 0; s < m; ++s) {
                const int q = ur_col[3 * j + s];
                unsigned hsh = ((unsigned)q * 2654435761u) >> 24;
                int probes = 0;
                while (true) {
                    const int old = atomicCAS(&keys[hsh], 0x7fffffff, q);
                    if (old == 0x7fffffff || old == q) break;
                    hsh = (hsh + 1) & (kRapSet - 1);
                    if (++probes >= kRapSet) { atomicExch(err_flag, 4); break; }
                }
            }
        }
    }
    __syncthreads();
    if (PASS == 0) {               // counting pass: the number of occupied slots is all that is needed -- no sort (2.4 -> 1.25 ms on level 1)
        for (int s = lane; s < kRapSet; s += 64) cnt += keys[s] != 0x7fffffff;
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if (lane == 0) c_cnt[p] = cnt;
        return;
    }
    // ---- compact the occupied slots to the front (ballot ranks), then bitonic-sort only the next power of two >= their number
    // (a coarse row has ~21 distinct columns: a 32-element sort instead of the 256-slot one)
    {
        int kk[kRapSet / 64];
        int base = 0;
#pragma unroll
        for (int g = 0; g < kRapSet / 64; ++g) kk[g] = keys[g * 64 + lane];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < kRapSet / 64; ++g) {
            const bool occ = kk[g] != 0x7fffffff;
            const unsigned long long m = __ballot(occ);
            if (occ) keys[base + __popcll(m & ((1ull << lane) - 1ull))] = kk[g];
            base += __popcll(m);
        }
        cnt = base;
        int n2 = 2;
        while (n2 < cnt) n2 <<= 1;
        __syncthreads();
        for (int s = cnt + lane; s < n2; s += 64) keys[s] = 0x7fffffff;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int idx = lane; idx < n2; idx += 64) {
                    const int partner = idx ^ j;
                    if (partner > idx) {
                        const int a = keys[idx], b = keys[partner];
                        const bool up = (idx & k) == 0;
                        if ((a > b) == up) { keys[idx] = b; keys[partner] = a; }
                    }
                }
                __syncthreads();
            }
        for (int s = n2 + lane; s < kRapSet; s += 64) keys[s] = 0x7fffffff;       // (the numeric step reads keys[g * 64 + lane] only below cnt)
        __syncthreads();
    }
    }
    // ---- step 2: numeric.  Every lane owns up to 4 output columns (cnt <= 256).  The triples of a chunk of children
    // are first gathered IN PARALLEL into LDS as (q, (u_ip a_ij) * u_jq) in the host's walk order -- slot of (child,
    // entry, s) = 3 * (entries of the earlier children + entry) + s -- then every lane runs down the list once and adds
    // the products of its columns: the same additions in the same order as the host, at LDS-broadcast cost.
    __shared__ __attribute__((aligned(16))) int sq[kRapSlots];
    __shared__ __attribute__((aligned(16))) double sv[kRapSlots];
    __shared__ int c_incl[64], c_beg[64];
    __shared__ double c_uip[64];
    const int out0 = c_ptr[p];
    int mine[4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < 4; ++g) mine[g] = g * 64 + lane < cnt ? keys[g * 64 + lane] : -1;
    const int groups = (cnt + 63) >> 6;
    int t0 = ub;
    while (t0 < ue) {
        // lane l looks at child t0 + l: how many children fit the LDS list?
        const int tt = t0 + lane;
        const int ci = tt < ue ? u_ridx[tt] : -1;
        const int cb = ci >= 0 ? a_ptr[ci] : 0;
        const int len = ci >= 0 ? a_ptr[ci + 1] - cb : 0;
        int incl = len;
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        const unsigned long long fits = __ballot(tt < ue && incl * 3 <= kRapSlots);
        const int m = __popcll(fits);          // incl is non-decreasing: the fitting children are a prefix
        __syncthreads();                       // the previous chunk's list has been consumed
        if (m == 0) {
            // a single row of A longer than the list: walk it straight from memory (all lanes, same order)
            const int i = u_ridx[t0];
            const double uip = u_cval[t0];
            for (int e = a_ptr[i]; e < a_ptr[i + 1]; ++e) {
                const int j = a_idx[e];
                const double w = uip * a_val[e];
                const int mj = ur_cnt[j] < 3 ? ur_cnt[j] : 3;
                for (int s2 = 0; s2 < mj; ++s2) {
                    const int q = ur_col[3 * j + s2];
                    const double pr = w * ur_val[3 * j + s2];
                    for (int g = 0; g < groups; ++g) if (q == mine[g]) acc[g] = acc[g] + pr;
                }
            }
            t0 += 1;
            continue;
        }
        c_incl[lane] = incl; c_beg[lane] = cb; c_uip[lane] = tt < ue ? u_cval[tt] : 0.0;
        __syncthreads();
        const int pairs = c_incl[m - 1];
        for (int x = lane; x < pairs; x += 64) {
            int c = 0;                          // child of pair x: first c with c_incl[c] > x
            while (c_incl[c] <= x) ++c;
            const int e = c_beg[c] + (x - (c > 0 ? c_incl[c - 1] : 0));
            const int j = a_idx[e];
            const double w = c_uip[c] * a_val[e];
            const int mj = ur_cnt[j] < 3 ? ur_cnt[j] : 3;
            for (int s2 = 0; s2 < 3; ++s2) {
                const bool on = s2 < mj;
                sq[3 * x + s2] = on ? ur_col[3 * j + s2] : -2;
                sv[3 * x + s2] = on ? w * ur_val[3 * j + s2] : 0.0;
            }
        }
        __syncthreads();
        const int slots = 3 * pairs;
        if (groups == 1) {
            // (four slots per step through 128-bit LDS reads; the additions stay one after the other in slot order)
            double a0 = acc[0];
            const int m0 = mine[0];
            int x = 0;
            for (; x + 4 <= slots; x += 4) {
                const int4 q4 = *reinterpret_cast<const int4*>(&sq[x]);
                const double2 v01 = *reinterpret_cast<const double2*>(&sv[x]);
                const double2 v23 = *reinterpret_cast<const double2*>(&sv[x + 2]);
                if (q4.x == m0) a0 = a0 + v01.x;
                if (q4.y == m0) a0 = a0 + v01.y;
                if (q4.z == m0) a0 = a0 + v23.x;
                if (q4.w == m0) a0 = a0 + v23.y;
            }
            for (; x < slots; ++x) if (sq[x] == m0) a0 = a0 + sv[x];
            acc[0] = a0;
        } else {
            for (int x = 0; x < slots; ++x) {
                const int q = sq[x];
                const double pr = sv[x];
                for (int g = 0; g < groups; ++g) if (q == mine[g]) acc[g] = acc[g] + pr;
            }
        }
        t0 += m;
    }
    for (int g = 0; g < groups; ++g)
        if (mine[g] >= 0) { c_idx[out0 + g * 64 + lane] = mine[g]; c_val[out0 + g * 64 + lane] = acc[g]; }
}

// ---- dense inverse of the coarsest operator, built on the device from the host's sparse LDL^T factor --------------------------------
// (multigrid_solver.cpp:1401 + :1075: the reference factors A_L once per solve() and back-substitutes once per cycle; GMG_COARSE_DEVICE_INVERSE
// turns the per-cycle part into ONE dense symmetric matrix-vector product on the device -- no host round trip inside the cycle.)
// X = A_L^-1 = P^T (L^-T D^-1 L^-1) P, stored in the FACTOR's numbering (the product kernel permutes the vectors): n right-hand sides e_c,
// every one an independent pair of sparse triangular solves.  One workgroup owns a TILE of 64 columns c (one lane per column: a row of X
// restricted to the tile is 512 contiguous bytes) and carries it through the whole factor:
//   * down (L y = e_c): y is nonzero only at ancestors of c in the elimination tree, so only chunks whose subtree meets the tile are visited
//     (q_fdesc); one wave walks them in ascending order -- push form: the chunk's own columns are finished in registers, then every row below
//     receives its update (a lane only ever re-reads what it wrote itself: program order);
//   * up (L^T x = D^-1 y), needed for rows j >= c only (the inverse is symmetric; the other triangle is mirrored afterwards): pull form by
//     LEVELS of the elimination tree from the roots down, the chunks of a level dealt to the workgroup's waves, a barrier per level.
// Every entry of X is produced by one lane with a fixed order of operations: the result is deterministic (ranks of a multi-GPU job that
// replicate the coarsest level get the same bits).  Traffic: a tile reads sum(rows of the chunks) x 512 B per pass from its own 64-column slab,
// which the L2 / memory-side cache serve; arithmetic nnz(L) x n fused multiply-adds in all -- microseconds of the chip's fp64 rate, which is why
// the build is latency-bound (one dependent global round trip per level) and not a candidate for MFMA.
struct InvFactor {
    int n, nq, nlev;
    // chunk records {first column | width << 24, first row entry, end of its row entries, chunk number}: lev_meta in the order of the way up (level by
    // level, the chunks with >= kInvBigRows rows first: lev_big[level] of them), tile_meta in the order of a tile's way down (tile_ptr)
    const int4 *lev_meta, *tile_meta;
    const int *rows, *lev_ptr, *lev_big, *tile_ptr;
    const double *vals, *tri, *dinv;
};
constexpr int kInvChunk = 8;            // = gmg::SupernodalLDLT::kChunk
constexpr int kInvWaves = 16;
constexpr int kInvBigRows = 192;        // a chunk with at least this many rows below it is worked off by ALL waves of the workgroup together

// W = columns of a tile (16 / 32 / 64).  A wave covers W columns x S = 64 / W ROW SLOTS: lane = slot * W + column, so that one load instruction
// fetches S different rows of the tile's slab (a small coarsest level has few tiles: the width is chosen so that there are enough workgroups,
// and the slots keep all 64 lanes busy).  Way down: the chunks on the tile's paths to the root (tile_meta, ascending), one after the other, every
// wave pushing its share of the rows below.  Way up: level by level; the big chunks of a level (separators: hundreds of rows below them; the host
// lists them first) are worked off one by one by ALL waves, their partial sums added in wave order through LDS; the others go to the waves
// one chunk each, side by side.  Every value has one fixed summation order: same bits on every device.
// The kernel is a chain of dependent memory round trips (record -> row numbers -> rows of X -> store), so the record of the NEXT chunk is
// requested before the current one is worked on, and a chunk's own rows of X are requested before its gathers.
template <int W>
__global__ __launch_bounds__(64 * kInvWaves) void coarse_inverse_tiles(InvFactor F, double* X) {
    constexpr int S = 64 / W;
    __shared__ double red[kInvWaves][kInvChunk][W];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cl = lane % W, slot = lane / W;
    const int c0 = blockIdx.x * W, c = c0 + cl;
    const bool valid = c < F.n;
    const int64_t ld = F.n;
    double* Xc = X + (valid ? c : 0);
    // partial sums of the S slots -> every slot holds the total (fixed order: slot pairs (0,1)(2,3), then the two pairs)
    auto slots_sum = [&](double v) {
        if constexpr (S >= 2) v += __shfl_xor(v, W, 64);
        if constexpr (S >= 4) v += __shfl_xor(v, 2 * W, 64);
        return v;
    };
    if (wave == 0 && slot == 0 && valid) Xc[(int64_t)c * ld] = 1.0;
    __syncthreads();
    // ---- way down (L y = e_c, push form).  y of the previous chunk is stored by wave 0 only after the barrier that follows the other waves' loads of it
    {
        double y[kInvChunk];
        int pcol0 = -1, pw = 0;
        const int t0 = F.tile_ptr[blockIdx.x], t1 = F.tile_ptr[blockIdx.x + 1];
        int4 next = t0 < t1 ? F.tile_meta[t0] : make_int4(0, 0, 0, 0);
        for (int t = t0; t <= t1; ++t) {
            if (wave == 0 && slot == 0 && valid && pcol0 >= 0) {
#pragma unroll
                for (int jj = 1; jj < kInvChunk; ++jj) if (jj < pw) Xc[(int64_t)(pcol0 + jj) * ld] = y[jj];
            }
            if (t == t1) break;
            const int4 m = next;
            if (t + 1 < t1) next = F.tile_meta[t + 1];
            const int col0 = m.x & 0xffffff, w = m.x >> 24, r0 = m.y, r1 = m.z;
#pragma unroll
            for (int jj = 0; jj < kInvChunk; ++jj) y[jj] = (jj < w && valid) ? Xc[(int64_t)(col0 + jj) * ld] : 0.0;
            const double* T = F.tri + (int64_t)m.w * kInvChunk * kInvChunk;
#pragma unroll
            for (int jj = 0; jj < kInvChunk; ++jj)
#pragma unroll
                for (int ii = jj + 1; ii < kInvChunk; ++ii) y[ii] -= T[ii * kInvChunk + jj] * y[jj];
            pcol0 = col0; pw = w;
            // rows r0 + (wave * S + slot) + k * (kInvWaves * S): every (row, column) has exactly one lane; four rows in flight per lane
            for (int i = r0 + wave * S + slot; i < r1; i += 4 * kInvWaves * S) {
                double xv[4], acc[4];
                int row[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int iu = i + u * kInvWaves * S;
                    const bool in = iu < r1 && valid;
                    row[u] = in ? F.rows[iu] : -1;
                    xv[u] = in ? Xc[(int64_t)row[u] * ld] : 0.0;
                    const double* V = F.vals + (int64_t)(iu < r1 ? iu : r0) * kInvChunk;
                    acc[u] = 0.0;
#pragma unroll
                    for (int jj = 0; jj < kInvChunk; ++jj) acc[u] += V[jj] * y[jj];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (row[u] >= 0) Xc[(int64_t)row[u] * ld] = xv[u] - acc[u];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- way up (L^T x = D^-1 y, pull form), rows >= c0 only
    auto own_rows = [&](int col0, int w, double (&yv)[kInvChunk]) {                 // D^-1 y of the chunk's own columns (requested before the gathers)
#pragma unroll
        for (int jj = 0; jj < kInvChunk; ++jj) yv[jj] = (jj < w && valid) ? Xc[(int64_t)(col0 + jj) * ld] * F.dinv[col0 + jj] : 0.0;
    };
    auto gather_rows = [&](int r0, int r1, int first, int stride, double (&acc)[kInvChunk]) {
        constexpr int U = 8;                                                 // gathers in flight per lane (a big level's slabs live in HBM: latency, not bytes)
        for (int i = r0 + first; i < r1; i += U * stride) {
            double xi[U];
            int iu[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { iu[u] = i + u * stride; xi[u] = (iu[u] < r1 && valid) ? Xc[(int64_t)F.rows[iu[u]] * ld] : 0.0; }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double* V = F.vals + (int64_t)(iu[u] < r1 ? iu[u] : r0) * kInvChunk;
#pragma unroll
                for (int jj = 0; jj < kInvChunk; ++jj) acc[jj] -= V[jj] * xi[u];
            }
        }
    };
    auto finish_chunk = [&](int q, int col0, int w, double (&acc)[kInvChunk], const double (&yv)[kInvChunk]) {        // acc: sums over the rows below; every slot holds them
#pragma unroll
        for (int jj = 0; jj < kInvChunk; ++jj) acc[jj] += yv[jj];
        const double* T = F.tri + (int64_t)q * kInvChunk * kInvChunk;
#pragma unroll
        for (int jj = kInvChunk - 2; jj >= 0; --jj)
#pragma unroll
            for (int ii = jj + 1; ii < kInvChunk; ++ii) acc[jj] -= T[ii * kInvChunk + jj] * acc[ii];
        if (slot == 0 && valid) {
#pragma unroll
            for (int jj = 0; jj < kInvChunk; ++jj) if (jj < w) Xc[(int64_t)(col0 + jj) * ld] = acc[jj];
        }
    };
    for (int lev = 0; lev < F.nlev; ++lev) {
        const int k0 = F.lev_ptr[lev], k1 = F.lev_ptr[lev + 1], kb = k0 + F.lev_big[lev];
        for (int k = k0; k < kb; ++k) {                                      // the big ones, together
            const int4 m = F.lev_meta[k];
            const int col0 = m.x & 0xffffff, w = m.x >> 24;
            if (col0 + w - 1 < c0) continue;                                // (workgroup-uniform)
            double acc[kInvChunk], yv[kInvChunk];
#pragma unroll
            for (int jj = 0; jj < kInvChunk; ++jj) { acc[jj] = 0.0; yv[jj] = 0.0; }
            if (wave == 0) own_rows(col0, w, yv);
            gather_rows(m.y, m.z, wave * S + slot, kInvWaves * S, acc);
#pragma unroll
            for (int jj = 0; jj < kInvChunk; ++jj) { const double v = slots_sum(acc[jj]); if (slot == 0) red[wave][jj][cl] = v; }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int jj = 0; jj < kInvChunk; ++jj) {
                    double v = 0.0;
#pragma unroll
                    for (int wv = 0; wv < kInvWaves; ++wv) v += red[wv][jj][cl];
                    acc[jj] = v;
                }
                finish_chunk(m.w, col0, w, acc, yv);
            }
            __syncthreads();
        }
        int k = kb + wave;                                                   // the others: one per wave, the next record in flight
        int4 next = k < k1 ? F.lev_meta[k] : make_int4(0, 0, 0, 0);
        for (; k < k1; k += kInvWaves) {
            const int4 m = next;
            if (k + kInvWaves < k1) next = F.lev_meta[k + kInvWaves];
            const int col0 = m.x & 0xffffff, w = m.x >> 24;
            if (col0 + w - 1 < c0) continue;
            double acc[kInvChunk], yv[kInvChunk];
#pragma unroll
            for (int jj = 0; jj < kInvChunk; ++jj) acc[jj] = 0.0;
            own_rows(col0, w, yv);
            gather_rows(m.y, m.z, slot, S, acc);
#pragma unroll
            for (int jj = 0; jj < kInvChunk; ++jj) acc[jj] = slots_sum(acc[jj]);
            finish_chunk(m.w, col0, w, acc, yv);
        }
        __syncthreads();
    }
}

// X[j][c] for j < c from X[c][j] (the tiles computed rows j >= c0 of their columns): 64 x 64 blocks through LDS, both sides coalesced.  A block on
// the diagonal mirrors its own lower triangle.  grid: (nb, nb), nb = ceil(n / 64); blocks below the diagonal return.
__global__ __launch_bounds__(256) void mirror_lower_to_upper(double* __restrict__ X, int n) {
    const int bj = blockIdx.y, bc = blockIdx.x;           // writes block (row block bj, column block bc), bj <= bc, from block (bc, bj)
    if (bj > bc) return;
    __shared__ double t[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int row = bc * 64 + r, col = bj * 64 + tx;
        t[r][tx] = (row < n && col < n) ? X[(int64_t)row * n + col] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int row = bj * 64 + r, col = bc * 64 + tx;   // X[row][col] = X[col][row] = t[tx][r]
        if (row < n && col < n && row < col) X[(int64_t)row * n + col] = t[tx][r];
    }
}

// out[perm[j]][perm[c]] = X[j][c]: the inverse from the factor's numbering into the level's (perm: factor -> level, inv its inverse), one
// workgroup per row: the row goes through LDS, so that both the read of X and the write of `out` are contiguous.  Dynamic LDS: n doubles.
__global__ __launch_bounds__(256) void permute_symmetric(const double* __restrict__ X, const int* __restrict__ perm, const int* __restrict__ inv, int n,
                                                         double* __restrict__ out, int ldo) {
    extern __shared__ double row[];
    const int j = blockIdx.x;
    const double* src = X + (int64_t)j * n;
    for (int c = threadIdx.x; c < n; c += 256) row[c] = src[c];
    __syncthreads();
    double* dst = out + (int64_t)perm[j] * ldo;       // (ldo >= n: rows of the result start at 16-byte boundaries, kernels.hip.hpp::dense_symv_v2)
    for (int c = threadIdx.x; c < n; c += 256) dst[c] = row[inv[c]];
}

// the same for rows that do not fit LDS (n > 8 192: only with GMG_COARSE_DEVICE_INVERSE forced on a large coarsest level): scattered writes
__global__ __launch_bounds__(256) void permute_symmetric_scatter(const double* __restrict__ X, const int* __restrict__ perm, int n, double* __restrict__ out, int ldo) {
    const int j = blockIdx.x;
    const double* src = X + (int64_t)j * n;
    double* dst = out + (int64_t)perm[j] * ldo;
    for (int c = threadIdx.x; c < n; c += 256) dst[perm[c]] = src[c];
}

}  // namespace gmgs
