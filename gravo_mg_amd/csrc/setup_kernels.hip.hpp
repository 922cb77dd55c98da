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

constexpr int kMaxRow = 96;          // longest row the device builder sorts in private memory (host fallback beyond)

// mode: 0 = every entry, 1 = only entries whose column lies in the row's block, 2 = only entries leaving the block
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
    out_col = nc;
    return !inside;
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

// One thread per slice row position: gather the kept entries, sort them by device column, write them (and the
// padding) into the slice.  diag (optional) receives the dropped diagonal entry (1.0 for padding rows).
template <class ColT>
__global__ void sell_fill(const int* __restrict__ pbeg, const int* __restrict__ pend, const int* __restrict__ idx,
                          const double* __restrict__ val, RowFilter f,
                          const int* __restrict__ order, int lpr, int n_rows_pad, const int64_t* __restrict__ slice_ptr,
                          ColT* __restrict__ col, double* __restrict__ out_val, double* __restrict__ diag, int* __restrict__ err_flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows_pad) return;
    const int rps = 64 / lpr;
    const int s = i / rps, l = i % rps;
    const int r = order ? order[i] : i;
    const int old = f.new2old_row[r];
    int cs[kMaxRow];
    double vs[kMaxRow];
    int n = 0;
    double dg = 1.0;
    bool has_diag = old < 0;
    if (old >= 0) {
        for (int p = pbeg[old]; p < pend[old]; ++p) {
            const int oc = idx[p];
            if (f.drop_diag && oc == old) { dg = val[p]; has_diag = true; continue; }
            int c;
            if (!keep_entry(f, r, old, oc, c)) continue;
            if (n >= kMaxRow) { atomicExch(err_flag, 1); break; }
            // insertion sort by device column (rows are short; stable, deterministic)
            int q = n;
            const double v = val[p];
            while (q > 0 && cs[q - 1] > c) { cs[q] = cs[q - 1]; vs[q] = vs[q - 1]; --q; }
            cs[q] = c; vs[q] = v;
            ++n;
        }
    }
    if (diag) {
        if (f.drop_diag && (!has_diag || dg == 0.0)) atomicExch(err_flag, 2);      // missing / zero diagonal
        diag[r] = dg;
    }
    const int64_t base = slice_ptr[s];
    const int w = (int)((slice_ptr[s + 1] - base) >> 6);
    for (int e = 0; e < w * lpr; ++e) {
        const int64_t q = base + (int64_t)(e / lpr) * 64 + l * lpr + (e % lpr);
        if (e < n) { col[q] = (ColT)cs[e]; out_val[q] = vs[e]; }
        else { col[q] = (ColT)0; out_val[q] = 0.0; }
    }
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

}  // namespace gmgs
