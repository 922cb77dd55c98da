// engine_setup.hip.hpp -- part of libgravomg_hip.so's single translation unit (included by engine.hip, in this order:
// engine_state, engine_setup, engine_cycle).  Device-side construction of the level data: SELL / block-CSR layouts, Galerkin products, pattern checks.
#pragma once

namespace {

// ---- device-side layout construction (setup_kernels.hip.hpp) ------------------------------------------------
int upload_csr(gmg_handle h, DevCsr& d, const Compressed& m) {
    free_csr(d);
    d.n_outer = m.n_outer;
    int rc;
    if ((rc = upload(h, &d.ptr, m.ptr)) || (rc = upload(h, &d.idx, m.idx)) || (rc = upload(h, &d.val, m.val))) return rc;
    return GMG_OK;
}

int upload_csr_raw(gmg_handle h, DevCsr& d, int n_outer, const int* ptr, const int* idx, const double* val) {
    free_csr(d);
    d.n_outer = n_outer;
    const size_t nnz = (size_t)ptr[n_outer];
    HIPCHK(dev_malloc((void**)&d.ptr, sizeof(int) * ((size_t)n_outer + 1)));
    HIPCHK(dev_malloc((void**)&d.idx, sizeof(int) * std::max<size_t>(nnz, 1)));
    HIPCHK(dev_malloc((void**)&d.val, sizeof(double) * std::max<size_t>(nnz, 1)));
    int rc;
    if ((rc = h2d(h, d.ptr, ptr, sizeof(int) * ((size_t)n_outer + 1))) || (rc = h2d(h, d.idx, idx, sizeof(int) * nnz)) || (rc = h2d(h, d.val, val, sizeof(double) * nnz))) return rc;
    return GMG_OK;
}

template <class T>
struct DevTmp {                       // scratch device array released at scope exit (stream-ordered reuse through the pool)
    T* p = nullptr;
    ~DevTmp() { if (p) (void)dev_free(p); }
    int alloc(gmg_handle h, size_t n) { HIPCHK(dev_malloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T))); return GMG_OK; }
};

// out[0..n] = exclusive prefix sums of in[0..n) on the stream; *total_host (pinned or pageable) receives out[n] after
// the caller synchronises.
template <class TIn, class TOut>
int device_scan(gmg_handle h, const TIn* in, int n, TOut* out, TOut* total_host) {
    const int tiles = std::max(1, (n + gmgs::kScanTile - 1) / gmgs::kScanTile);
    DevTmp<TOut> tile;
    int rc;
    if ((rc = tile.alloc(h, (size_t)tiles + 1))) return rc;
    hipLaunchKernelGGL((gmgs::scan_tile_sums<TIn, TOut>), dim3(tiles), dim3(256), 0, h->stream, in, n, tile.p);
    hipLaunchKernelGGL((gmgs::scan_tile_offsets<TOut>), dim3(1), dim3(1024), 0, h->stream, tile.p, tiles, tile.p + tiles);
    hipLaunchKernelGGL((gmgs::scan_tile_apply<TIn, TOut>), dim3(tiles), dim3(256), 0, h->stream, in, n, (const TOut*)tile.p, out);
    if (total_host) HIPCHK(hipMemcpyAsync(total_host, tile.p + tiles, sizeof(TOut), hipMemcpyDeviceToHost, h->stream));
    return GMG_OK;
}

// Builds one SELL matrix on the device.  pbeg/pend/idx/val: source rows (natural numbering); f: row/column maps and
// filter; d_order: optional slice-position -> device-row map (uploaded by the caller; also stored as row_of);
// col16: write 16-bit columns into *col16_out (in-block part of a blocked level) instead of out.col.
// src_out / diag_src_out (optional, first build only): allocate and fill the source maps of the layout (gmgs::sell_fill's `src` / `diag_src`)
int device_build_sell(gmg_handle h, DevSell& out, const int* pbeg, const int* pend, const int* idx, const double* val, gmgs::RowFilter f,
                      const int* d_order, int n_rows_pad, int lpr, unsigned short** col16_out, double* d_diag, int* d_err, bool refill = false,
                      int** src_out = nullptr, int** diag_src_out = nullptr) {
    auto fill = [&]() {
        const dim3 grid((n_rows_pad + 255) / 256);
        int* src = src_out ? *src_out : nullptr;
        int* dsrc = diag_src_out ? *diag_src_out : nullptr;
        if (col16_out) hipLaunchKernelGGL(gmgs::sell_fill<unsigned short>, grid, dim3(256), 0, h->stream, pbeg, pend, idx, val, f, d_order, lpr, n_rows_pad, out.slice_ptr, *col16_out, out.val, d_diag, d_err, src, dsrc);
        else hipLaunchKernelGGL(gmgs::sell_fill<int>, grid, dim3(256), 0, h->stream, pbeg, pend, idx, val, f, d_order, lpr, n_rows_pad, out.slice_ptr, out.col, out.val, d_diag, d_err, src, dsrc);
    };
    if (refill) {        // same pattern as the matrix this layout was built from: slice pointers stand, entries are rewritten
        if (!out.slice_ptr || !out.val) return fail(h, GMG_ERR_STATE, "refill of a layout that was never built");
        fill();
        return GMG_OK;
    }
    free_sell(out);
    const int rps = 64 / lpr;
    out.lpr = lpr;
    out.n_slices = n_rows_pad / rps;
    DevTmp<int> len;
    DevTmp<int64_t> widths;
    int rc;
    if ((rc = len.alloc(h, n_rows_pad)) || (rc = widths.alloc(h, out.n_slices))) return rc;
    HIPCHK(dev_malloc((void**)&out.slice_ptr, sizeof(int64_t) * ((size_t)out.n_slices + 1)));
    hipLaunchKernelGGL(gmgs::row_lengths, dim3((n_rows_pad + 255) / 256), dim3(256), 0, h->stream, pbeg, pend, idx, f, d_order, n_rows_pad, len.p, d_err);
    hipLaunchKernelGGL(gmgs::slice_widths, dim3((out.n_slices + 255) / 256), dim3(256), 0, h->stream, len.p, lpr, out.n_slices, widths.p);
    int64_t total = 0;
    if ((rc = device_scan<int64_t, int64_t>(h, widths.p, out.n_slices, out.slice_ptr, &total))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    out.stored = total;
    HIPCHK(dev_malloc((void**)&out.val, std::max<int64_t>(out.stored, 1) * sizeof(double)));
    if (col16_out) {
        if (*col16_out) { (void)dev_free(*col16_out); *col16_out = nullptr; }
        HIPCHK(dev_malloc((void**)col16_out, std::max<int64_t>(out.stored, 1) * sizeof(unsigned short)));
    } else HIPCHK(dev_malloc((void**)&out.col, std::max<int64_t>(out.stored, 1) * sizeof(int)));
    if (src_out) { if (*src_out) { (void)dev_free(*src_out); *src_out = nullptr; } HIPCHK(dev_malloc((void**)src_out, std::max<int64_t>(out.stored, 1) * sizeof(int))); }
    if (diag_src_out) { if (*diag_src_out) { (void)dev_free(*diag_src_out); *diag_src_out = nullptr; } HIPCHK(dev_malloc((void**)diag_src_out, (size_t)std::max(n_rows_pad, 1) * sizeof(int))); }
    fill();
    return GMG_OK;
}

int build_ell3(gmg_handle h, DevEll3& e, const DevCsr& dU, int n_fine, int* d_err) {
    free_ell3(e);
    e.n = n_fine;
    HIPCHK(dev_malloc((void**)&e.cnt, sizeof(int) * std::max(n_fine, 1)));
    HIPCHK(dev_malloc((void**)&e.col, sizeof(int) * (size_t)std::max(n_fine, 1) * 3));
    HIPCHK(dev_malloc((void**)&e.val, sizeof(double) * (size_t)std::max(n_fine, 1) * 3));
    HIPCHK(hipMemsetAsync(e.cnt, 0, sizeof(int) * n_fine, h->stream));
    hipLaunchKernelGGL(gmgs::ell3_from_csc, dim3((dU.n_outer + 255) / 256), dim3(256), 0, h->stream, dU.ptr, dU.idx, dU.val, dU.n_outer, e.cnt, e.col, e.val, d_err);
    hipLaunchKernelGGL(gmgs::ell3_sort, dim3((n_fine + 255) / 256), dim3(256), 0, h->stream, e.cnt, n_fine, e.col, e.val);
    return GMG_OK;
}

// Device copies of every U_k (by coarse column, and regrouped by fine row): built once per hierarchy.
int ensure_device_transfers(gmg_handle h) {
    const int L = h->L;
    bool complete = h->dU_ready && (int)h->dU.size() == L;
    for (int k = 0; k < L && complete; ++k) complete = h->dU[k].ptr != nullptr;      // (a partitioned set-up releases U_0 when it is done)
    if (complete) return GMG_OK;
    if (!h->dU_ready || (int)h->dU.size() != L) {
        drop_device_transfers(h);
        h->dU.assign(L, DevCsr());
        h->dE3.assign(L, DevEll3());
    }
    DevTmp<int> d_err;
    int rc, herr = 0;
    if ((rc = d_err.alloc(h, 1))) return rc;
    HIPCHK(hipMemsetAsync(d_err.p, 0, sizeof(int), h->stream));
    for (int k = 0; k < L; ++k) {
        if (h->dU[k].ptr) continue;
        rc = upload_csr(h, h->dU[k], h->U[k]);
        if (rc == GMG_OK) rc = build_ell3(h, h->dE3[k], h->dU[k], h->U[k].n_inner, d_err.p);
        if (rc) return rc;
    }
    HIPCHK(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));      // pageable host arrays have been consumed
    h->dU_flagged = h->dU_flagged || herr != 0;      // a U row with more than 3 entries: the device RAP / layout builder cannot take it
    h->dU_ready = true;
    return GMG_OK;
}

// The base order of a reordered level 0 in use (gmg_set_system decides per system, choose_base_order): new -> old.
inline const std::vector<int>& base_order(gmg_handle h) { return h->base_order_choice == 1 ? h->bfs_order : h->cluster_order; }

static int ensure_device_order(gmg_handle h, const std::vector<int>& ord, int** d_ord, int** d_inv) {
    if (*d_ord) return GMG_OK;
    const int n = (int)ord.size();
    std::vector<int> inv(n);
    parallel_ranges(n, std::min(h->cfg.host_threads, 32), [&](int lo, int hi, int) { for (int r = lo; r < hi; ++r) inv[ord[r]] = r; });
    int rc;
    if ((rc = upload(h, d_ord, ord)) || (rc = upload(h, d_inv, inv))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMG_OK;
}

// Which locality order serves THIS matrix better -- the hierarchy's cluster order (points grouped by parent, level by level:
// compact 2-D patches) or the breadth-first order over the point graph (wavefronts; only when the caller / the hierarchy object
// supplied one, gmg_set_fine_order)?  Both are scored on the uploaded pattern by the number of distinct cache lines the j-th
// gathers of 64 consecutive rows touch (gmgs::order_gather_score, 4 096 sampled windows, < 0.1 ms): a triangle mesh scores
// 0.27 lines per entry in breadth-first order and 0.37 in cluster order, a kNN point cloud 0.46 / 0.37, and the kernels follow
// (fine-level residual 49 / 72 us and 93 / 83 us).  Ties go to the cluster order.  Sets h->base_order_choice.
int choose_base_order(gmg_handle h, const DevCsr& dA, int n) {
    h->base_order_choice = 0;
    h->timing["base_order_score_cluster"] = h->timing["base_order_score_bfs"] = h->timing["base_order_choice"] = 0.0;
    if ((int)h->bfs_order.size() != n || (int)h->cluster_order.size() != n) { h->base_order_choice = (int)h->bfs_order.size() == n ? 1 : 0; return GMG_OK; }
    int rc;
    if ((rc = ensure_device_order(h, h->cluster_order, &h->d_cluster_order, &h->d_cluster_inv)) || (rc = ensure_device_order(h, h->bfs_order, &h->d_bfs_order, &h->d_bfs_inv))) return rc;
    DevTmp<unsigned long long> acc;
    if ((rc = acc.alloc(h, 4))) return rc;
    HIPCHK(hipMemsetAsync(acc.p, 0, sizeof(unsigned long long) * 4, h->stream));
    const int n_win = 4096;
    hipLaunchKernelGGL(gmgs::order_gather_score, dim3(n_win / 4), dim3(256), 0, h->stream, dA.ptr, dA.idx, h->d_cluster_order, h->d_cluster_inv, n, n_win, acc.p);
    hipLaunchKernelGGL(gmgs::order_gather_score, dim3(n_win / 4), dim3(256), 0, h->stream, dA.ptr, dA.idx, h->d_bfs_order, h->d_bfs_inv, n, n_win, acc.p + 2);
    unsigned long long host[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpyAsync(host, acc.p, sizeof host, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const double sc = host[1] ? (double)host[0] / (double)host[1] : 0.0, sb = host[3] ? (double)host[2] / (double)host[3] : 0.0;
    h->timing["base_order_score_cluster"] = sc; h->timing["base_order_score_bfs"] = sb;
    h->base_order_choice = sb < sc ? 1 : 0;
    h->timing["base_order_choice"] = h->base_order_choice;
    return GMG_OK;
}

// LHS pattern in the chosen base order (base_order(h)), made on the device from the uploaded LHS and copied to h->reo_ptr /
// h->reo_idx: the level-0 colouring then walks a locally ordered graph instead of chasing pointers through a randomly numbered
// one (3 M vertices in random order: 450 ms -> 20 ms).
int device_permute_pattern(gmg_handle h, const DevCsr& dA, int n, int64_t nnz) {
    int rc;
    const bool bfs = h->base_order_choice == 1;
    int** d_ord = bfs ? &h->d_bfs_order : &h->d_cluster_order;
    int** d_inv = bfs ? &h->d_bfs_inv : &h->d_cluster_inv;
    if ((rc = ensure_device_order(h, base_order(h), d_ord, d_inv))) return rc;
    DevTmp<int> len, pptr, pidx;
    if ((rc = len.alloc(h, n)) || (rc = pptr.alloc(h, (size_t)n + 1)) || (rc = pidx.alloc(h, (size_t)nnz))) return rc;
    hipLaunchKernelGGL(gmgs::perm_row_lengths, dim3((n + 255) / 256), dim3(256), 0, h->stream, dA.ptr, *d_ord, n, len.p);
    if ((rc = device_scan<int, int>(h, len.p, n, pptr.p, nullptr))) return rc;
    hipLaunchKernelGGL(gmgs::perm_fill, dim3((n + 255) / 256), dim3(256), 0, h->stream, dA.ptr, dA.idx, *d_ord, *d_inv, pptr.p, n, pidx.p);
    h->reo_ptr.resize((size_t)n + 1);
    h->reo_idx.resize((size_t)nnz);
    if ((rc = d2h(h, h->reo_ptr.data(), pptr.p, sizeof(int) * ((size_t)n + 1))) || (rc = d2h(h, h->reo_idx.data(), pidx.p, sizeof(int) * (size_t)nnz))) return rc;
    return GMG_OK;
}

// Host copy of A_k (natural numbering), on demand: the device keeps the master copy (Level::dA).
int ensure_host_A(gmg_handle h, int k, bool values) {
    Level& l = h->lv[k];
    if (l.hostA_pattern && (l.hostA_values || !values)) return GMG_OK;
    if (!l.dA.ptr) return fail(h, GMG_ERR_STATE, "level operator is neither on the host nor on the device");
    const int n = l.dA.n_outer;
    l.A.n_outer = n; l.A.n_inner = n;
    if (!l.hostA_pattern) {
        l.A.ptr.resize((size_t)n + 1);
        { int r2 = d2h(h, l.A.ptr.data(), l.dA.ptr, sizeof(int) * ((size_t)n + 1)); if (r2) return r2; }
        l.A.idx.resize((size_t)l.A.ptr[n]);
        { int r2 = d2h(h, l.A.idx.data(), l.dA.idx, sizeof(int) * l.A.idx.size()); if (r2) return r2; }
    }
    if (values && !l.hostA_values) {
        l.A.val.resize((size_t)l.nnz);
        { int r2 = d2h(h, l.A.val.data(), l.dA.val, sizeof(double) * l.A.val.size()); if (r2) return r2; }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    l.hostA_pattern = true;
    l.hostA_values = l.hostA_values || values;
    return GMG_OK;
}

// Ac = U^T A U on the device (setup_kernels.hip.hpp::rap_rows): count pass, device prefix sum, fill pass.  The result
// stays on the device (dC); `pattern` (row pointers + column indices, for the host ordering of that level) and
// `values` say what is copied to the host as well.  Returns 1 when the device kernel cannot take the input.
// reuse_pattern: dC still holds the product of the previous system, whose sparsity pattern (and hierarchy) is the same as
// this one's -- row pointers and column indices are valid already, only the numeric pass runs (*nnz_out must hold the
// known count).
int device_rap(gmg_handle h, const DevCsr& dA, const DevCsr& dU, const DevEll3& e3, DevCsr& dC, Compressed& C, bool pattern, bool values,
               int64_t* nnz_out, int* d_err, bool reuse_pattern = false, int rows_done = 0) {
    const int nc = dU.n_outer;
    if (reuse_pattern && dC.ptr && dC.idx && dC.val && dC.n_outer == nc && *nnz_out > 0 && !pattern) {
        const int64_t nnz = *nnz_out;
        // (rows_done: the first rows were computed already -- queued behind the chunks of the values upload, engine.hip::set_system_impl)
        if (rows_done > 0 && rows_done < nc) rows_done = 0;      // (the pipelined pieces follow a row LIST: a partial set is not a prefix of the natural order -- all rows again)
        if (rows_done < nc)
            hipLaunchKernelGGL(gmgs::rap_rows<2>, dim3(nc - rows_done), dim3(64), 0, h->stream, dA.ptr, dA.idx, dA.val, dU.ptr, dU.idx, dU.val, e3.cnt, e3.col, e3.val, nc,
                               (const int*)dC.ptr, (int*)nullptr, dC.idx, dC.val, d_err, rows_done);
        C.n_outer = nc; C.n_inner = nc;
        if (values) {
            int r2;
            C.ptr.resize((size_t)nc + 1); C.idx.resize((size_t)nnz); C.val.resize((size_t)nnz);
            if ((r2 = d2h(h, C.ptr.data(), dC.ptr, sizeof(int) * ((size_t)nc + 1))) || (r2 = d2h(h, C.idx.data(), dC.idx, sizeof(int) * (size_t)nnz)) ||
                (r2 = d2h(h, C.val.data(), dC.val, sizeof(double) * (size_t)nnz))) return r2;
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        return GMG_OK;
    }
    const bool trace = EnvSwitches::get().trace_setup;      // synchronising phase timers on stderr
    auto tph = clk::now();
    auto phase = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(h->stream);
        std::fprintf(stderr, "[gmg setup] rap nc=%d %-10s %.2f ms\n", nc, what, ms_since(tph));
        tph = clk::now();
    };
    free_csr(dC);
    dC.n_outer = nc;
    DevTmp<int> cnt;
    int rc;
    if ((rc = cnt.alloc(h, nc))) return rc;
    HIPCHK(dev_malloc((void**)&dC.ptr, sizeof(int) * ((size_t)nc + 1)));
    hipLaunchKernelGGL(gmgs::rap_rows<0>, dim3(nc), dim3(64), 0, h->stream, dA.ptr, dA.idx, dA.val, dU.ptr, dU.idx, dU.val, e3.cnt, e3.col, e3.val, nc,
                       (const int*)nullptr, cnt.p, (int*)nullptr, (double*)nullptr, d_err);
    phase("count");
    int nnz = 0, herr = 0;
    if ((rc = device_scan<int, int>(h, cnt.p, nc, dC.ptr, &nnz))) return rc;
    HIPCHK(hipMemcpyAsync(&herr, d_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    C.n_outer = nc; C.n_inner = nc;
    if (pattern || values) {
        C.ptr.resize((size_t)nc + 1);
        { int r2 = d2h(h, C.ptr.data(), dC.ptr, sizeof(int) * ((size_t)nc + 1)); if (r2) return r2; }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    phase("scan+ptr");
    if (herr) { free_csr(dC); return 1; }       // a coarse row overflows the device hash set (or a U row has > 3 entries): host fallback
    *nnz_out = nnz;
    HIPCHK(dev_malloc((void**)&dC.idx, sizeof(int) * std::max(nnz, 1)));
    HIPCHK(dev_malloc((void**)&dC.val, sizeof(double) * std::max(nnz, 1)));
    hipLaunchKernelGGL(gmgs::rap_rows<1>, dim3(nc), dim3(64), 0, h->stream, dA.ptr, dA.idx, dA.val, dU.ptr, dU.idx, dU.val, e3.cnt, e3.col, e3.val, nc,
                       (const int*)dC.ptr, (int*)nullptr, dC.idx, dC.val, d_err);
    phase("numeric");
    if (pattern || values) { C.idx.resize(nnz); { int r2 = d2h(h, C.idx.data(), dC.idx, sizeof(int) * nnz); if (r2) return r2; } }
    if (values) { C.val.resize(nnz); { int r2 = d2h(h, C.val.data(), dC.val, sizeof(double) * nnz); if (r2) return r2; } }
    if (pattern || values) HIPCHK(hipStreamSynchronize(h->stream));
    phase("fetch");
    return GMG_OK;
}

// the block-CSR fills of a blocked level
void launch_csr_fill_plain(gmg_handle h, Level& l, const DevCsr& dA, const gmgs::RowFilter& fe, const gmgs::RowFilter& fl, int* d_err) {
    const dim3 gr((l.n_pad + 255) / 256);
    hipLaunchKernelGGL(gmgs::csr_fill_plain<int>, gr, dim3(256), 0, h->stream, dA.ptr, dA.ptr + 1, dA.idx, dA.val, fe, l.n_pad, l.ee_ptr, l.ee_col, l.ee_val, d_err, l.src_ee);
    hipLaunchKernelGGL(gmgs::csr_fill_plain<unsigned short>, gr, dim3(256), 0, h->stream, dA.ptr, dA.ptr + 1, dA.idx, dA.val, fl, l.n_pad, l.ep_ptr, l.ep_col, l.ep_val, d_err, l.src_ep);
}
void launch_csr_fill(gmg_handle h, Level& l, const DevCsr& dA, const gmgs::RowFilter& fout, const int* d_blk_of_row, int* d_err) {
    const dim3 gr((l.n_pad + 255) / 256);
    hipLaunchKernelGGL(gmgs::csr_fill, gr, dim3(256), 0, h->stream, dA.ptr, dA.ptr + 1, dA.idx, dA.val, fout, d_blk_of_row, l.d_blk_begin, l.n_pad, l.bc_ptr, l.bc_mid, l.bc_col, l.bc_val, d_err);
}

// Layout of level k (operator, split operator) and of the transfers k -> k+1, built on the device from Level::dA and the
// device copies of U_k.  A row longer than gmgs::kMaxRow or a prolongation row with more than 3 entries raises *d_err:
// the caller then falls back to the host planner.
// own_rows / own_rows_next (partitioned set-up, engine_part.hip.hpp): masked copies of the row maps of level k / k + 1 -- new2old with -1 for the
// rows of other ranks -- used for everything whose ROWS are that level's (operator, block-CSR parts, prolongation; restriction: level k + 1);
// null = all rows.  The other ranks' rows come out as zero-width slices / empty chunks of the same global numbering.
int device_layout_level(gmg_handle h, int k, int* d_err, const int* own_rows = nullptr, const int* own_rows_next = nullptr) {
    const int L = h->L;
    Level& l = h->lv[k];
    const int* rows_k = own_rows ? own_rows : l.d_new2old;
    int rc;
    const bool trace = EnvSwitches::get().trace_setup;      // synchronising phase timers on stderr
    auto tph = clk::now();
    auto phase = [&](const char* what) {
        if (!trace) return;
        (void)hipStreamSynchronize(h->stream);
        std::fprintf(stderr, "[gmg setup] level %d %-10s %.2f ms\n", k, what, ms_since(tph));
        tph = clk::now();
    };
    struct Ref {
        int*& p;
        int alloc(gmg_handle h, size_t n) {
            if (p) { (void)dev_free(p); p = nullptr; }
            HIPCHK(dev_malloc((void**)&p, std::max<size_t>(n, 1) * sizeof(int)));
            return GMG_OK;
        }
    };
    Ref d_old2new{l.d_old2new}, d_blk_of_row{l.d_blk_of_row};       // owned by the level (refresh_system_values reuses them)
    if (!l.dA.ptr) {
        if ((rc = ensure_host_A(h, k, true)) || (rc = upload_csr(h, l.dA, l.A))) return rc;
    }
    const DevCsr& dA = l.dA;
    if ((rc = upload(h, &d_old2new.p, l.ord.old2new))) return rc;
    phase("old2new");
    gmgs::RowFilter f{rows_k, d_old2new.p, nullptr, nullptr, 0, 1};
    const int lanes_auto = l.n < kQuadLevelRows ? 4 : 1;
    const int lpr = (l.ord.blocked && k > 0) ? (h->cfg.block_lanes ? h->cfg.block_lanes : lanes_auto) : 1;
    HIPCHK(dev_malloc((void**)&l.diag, sizeof(double) * l.n_pad));
    const bool maps = h->part_world <= 1;      // (a partitioned handle releases dA: nothing to refresh from)
    if ((rc = device_build_sell(h, l.Aoff, dA.ptr, dA.ptr + 1, dA.idx, dA.val, f, nullptr, l.n_pad, lpr, nullptr, l.diag, d_err, false, maps ? &l.src_A : nullptr,
                                maps ? &l.src_diag : nullptr))) return rc;
    l.Aoff.nnz_real = l.nnz - l.n;
    phase("A");
    if (l.ord.blocked) {
        if ((rc = d_blk_of_row.alloc(h, l.n_pad)) || (rc = upload(h, &l.d_blk_begin, l.ord.blk_begin)) ||
            (rc = upload(h, &l.d_blk_ncolors, l.ord.blk_ncolors)) || (rc = upload(h, &l.d_row_color, l.ord.row_color))) return rc;
        hipLaunchKernelGGL(gmgs::block_of_rows, dim3(std::max(1, l.ord.n_blocks())), dim3(64), 0, h->stream, l.d_blk_begin, l.ord.n_blocks(), d_blk_of_row.p);
        gmgs::RowFilter fin{rows_k, d_old2new.p, d_blk_of_row.p, l.d_blk_begin, 1, 1};
        gmgs::RowFilter fout{rows_k, d_old2new.p, d_blk_of_row.p, l.d_blk_begin, 2, 1};
        if (wants_block_ep(h, lpr)) {
            // unpadded block sweep (gs_block_ep): "explicit" and "lower" parts as block-ordered CSRs.  Row pointers first:
            // the largest block's chunks size the sweep's LDS buffers
            gmgs::RowFilter fe{rows_k, d_old2new.p, d_blk_of_row.p, l.d_blk_begin, 3, 1};
            gmgs::RowFilter fl{rows_k, d_old2new.p, d_blk_of_row.p, l.d_blk_begin, 4, 1};
            DevTmp<int> len, d_max;
            int nnz_e = 0, nnz_l = 0, bmax[2] = {0, 0};
            if ((rc = len.alloc(h, l.n_pad)) || (rc = d_max.alloc(h, 2))) return rc;
            HIPCHK(dev_malloc((void**)&l.ee_ptr, sizeof(int) * ((size_t)l.n_pad + 1)));
            HIPCHK(dev_malloc((void**)&l.ep_ptr, sizeof(int) * ((size_t)l.n_pad + 1)));
            HIPCHK(hipMemsetAsync(d_max.p, 0, 2 * sizeof(int), h->stream));
            const dim3 gr((l.n_pad + 255) / 256), gb((l.ord.n_blocks() + 255) / 256);
            hipLaunchKernelGGL(gmgs::row_lengths, gr, dim3(256), 0, h->stream, dA.ptr, dA.ptr + 1, dA.idx, fe, (const int*)nullptr, l.n_pad, len.p, d_err);
            if ((rc = device_scan<int, int>(h, len.p, l.n_pad, l.ee_ptr, &nnz_e))) return rc;
            hipLaunchKernelGGL(gmgs::block_entry_max, gb, dim3(256), 0, h->stream, l.d_blk_begin, l.ord.n_blocks(), l.ee_ptr, d_max.p);
            hipLaunchKernelGGL(gmgs::row_lengths, gr, dim3(256), 0, h->stream, dA.ptr, dA.ptr + 1, dA.idx, fl, (const int*)nullptr, l.n_pad, len.p, d_err);
            if ((rc = device_scan<int, int>(h, len.p, l.n_pad, l.ep_ptr, &nnz_l))) return rc;
            hipLaunchKernelGGL(gmgs::block_entry_max, gb, dim3(256), 0, h->stream, l.d_blk_begin, l.ord.n_blocks(), l.ep_ptr, d_max.p + 1);
            HIPCHK(hipMemcpyAsync(bmax, d_max.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (bmax[0] <= kEpMaxBlockEntries && bmax[1] <= kEpMaxBlockLower) {
                l.use_ep = true;
                l.ee_nnz = nnz_e; l.ep_nnz = nnz_l;
                l.ep_cap_e = (bmax[0] + 63) / 64 * 64; l.ep_cap_l = std::max(bmax[1], 1);
                HIPCHK(dev_malloc((void**)&l.ee_col, sizeof(int) * (size_t)std::max(nnz_e, 1)));
                HIPCHK(dev_malloc((void**)&l.ee_val, sizeof(double) * (size_t)std::max(nnz_e, 1)));
                HIPCHK(dev_malloc((void**)&l.ep_col, sizeof(unsigned short) * (size_t)std::max(nnz_l, 1)));
                HIPCHK(dev_malloc((void**)&l.ep_val, sizeof(double) * (size_t)std::max(nnz_l, 1)));
                if (maps) {
                    for (int** q : {&l.src_ee, &l.src_ep}) { if (*q) (void)dev_free(*q); *q = nullptr; }
                    HIPCHK(dev_malloc((void**)&l.src_ee, sizeof(int) * (size_t)std::max(nnz_e, 1)));
                    HIPCHK(dev_malloc((void**)&l.src_ep, sizeof(int) * (size_t)std::max(nnz_l, 1)));
                }
                launch_csr_fill_plain(h, l, dA, fe, fl, d_err);
            } else {
                (void)dev_free(l.ee_ptr); l.ee_ptr = nullptr;
                (void)dev_free(l.ep_ptr); l.ep_ptr = nullptr;
            }
        }
        if (!l.use_ep && wants_block_csr(h, lpr)) {
            // off-block operator as a block-ordered CSR; its row pointers first: they tell whether the largest block's
            // chunk fits the sweep's LDS budget
            DevTmp<int> len, d_max;
            int nnz = 0, bmax = 0;
            if ((rc = len.alloc(h, l.n_pad)) || (rc = d_max.alloc(h, 1))) return rc;
            HIPCHK(dev_malloc((void**)&l.bc_ptr, sizeof(int) * ((size_t)l.n_pad + 1)));
            HIPCHK(hipMemsetAsync(d_max.p, 0, sizeof(int), h->stream));
            hipLaunchKernelGGL(gmgs::row_lengths, dim3((l.n_pad + 255) / 256), dim3(256), 0, h->stream, dA.ptr, dA.ptr + 1, dA.idx, fout, (const int*)nullptr, l.n_pad, len.p, d_err);
            if ((rc = device_scan<int, int>(h, len.p, l.n_pad, l.bc_ptr, &nnz))) return rc;
            hipLaunchKernelGGL(gmgs::block_entry_max, dim3((l.ord.n_blocks() + 255) / 256), dim3(256), 0, h->stream, l.d_blk_begin, l.ord.n_blocks(), l.bc_ptr, d_max.p);
            HIPCHK(hipMemcpyAsync(&bmax, d_max.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            if (bmax <= kBcsrMaxBlockEntries) {
                l.use_bcsr = true;
                l.bc_cap = (bmax + 63) / 64 * 64;
                l.bc_nnz = nnz;
                HIPCHK(dev_malloc((void**)&l.bc_mid, sizeof(int) * (size_t)l.n_pad));
                HIPCHK(dev_malloc((void**)&l.bc_col, sizeof(int) * (size_t)std::max(nnz, 1)));
                HIPCHK(dev_malloc((void**)&l.bc_val, sizeof(double) * (size_t)std::max(nnz, 1)));
                launch_csr_fill(h, l, dA, fout, d_blk_of_row.p, d_err);
            } else { (void)dev_free(l.bc_ptr); l.bc_ptr = nullptr; }
        }
        if (!l.use_ep) {
            // (the unpadded sweep reads its two block-ordered CSR operators only: no padded SELL copies of the split operator)
            if ((rc = device_build_sell(h, l.Ain, dA.ptr, dA.ptr + 1, dA.idx, dA.val, fin, nullptr, l.n_pad, lpr, &l.ain_col16, nullptr, d_err))) return rc;
            // the padded SELL form of the off-block operator is kept as well: with one right-hand side the sweep that
            // streams it straight into registers is the faster one (42 vs 47 us on the 506 k-row level; 104 vs 60 us at d = 3)
            if ((rc = device_build_sell(h, l.Aout, dA.ptr, dA.ptr + 1, dA.idx, dA.val, fout, nullptr, l.n_pad, lpr, nullptr, nullptr, d_err))) return rc;
        }
    }
    phase("A split");
    if (k == L) return GMG_OK;
    // ---- transfers k <-> k+1
    Level& c = h->lv[k + 1];
    const int* rows_c = own_rows_next ? own_rows_next : c.d_new2old;
    const DevCsr& dU = h->dU[k];
    const DevEll3& e3 = h->dE3[k];
    DevTmp<int> d_old2new_c, d_order, d_pbeg, d_pend;
    if ((rc = upload(h, &d_old2new_c.p, c.ord.old2new))) return rc;
    // restriction: rows = coarse points (columns of the CSC U), sorted by length inside windows like the host planner
    {
        const Compressed& U = h->U[k];
        const int np = c.n_pad, sigma = h->cfg.restrict_sigma;
        if (sigma > 0 && sigma <= gmgs::kWindowSortMax) {
            int pow2 = 1;
            while (pow2 < sigma) pow2 <<= 1;
            if ((rc = d_order.alloc(h, np))) return rc;
            hipLaunchKernelGGL(gmgs::window_order_by_length, dim3((np + sigma - 1) / sigma), dim3(256), 0, h->stream, dU.ptr, rows_c, np, sigma, pow2, d_order.p);
        } else if (sigma > 0) {
            std::vector<int> order(np);
            auto len_of = [&](int r) { int old = c.ord.new2old[r]; return old >= 0 ? U.ptr[old + 1] - U.ptr[old] : 0; };
            const int nwin = (np + sigma - 1) / sigma;
            for (int wi = 0; wi < nwin; ++wi) {
                int w = wi * sigma, we = std::min(np, w + sigma);
                std::iota(order.begin() + w, order.begin() + we, w);
                std::stable_sort(order.begin() + w, order.begin() + we, [&](int a, int b) { return len_of(a) > len_of(b); });
            }
            if ((rc = upload(h, &d_order.p, order))) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));      // `order` (pageable) dies at scope end
        }
        phase("R order");
        gmgs::RowFilter fr{rows_c, d_old2new.p, nullptr, nullptr, 0, 0};
        const int lpr_r = h->cfg.block_lanes == 1 ? 1 : 4;
        if ((rc = device_build_sell(h, l.R, dU.ptr, dU.ptr + 1, dU.idx, dU.val, fr, sigma > 0 ? d_order.p : nullptr, np, lpr_r, nullptr, nullptr, d_err))) return rc;
        l.R.nnz_real = U.nnz();
        if (sigma > 0) { l.R.row_of = d_order.p; d_order.p = nullptr; }     // the order array becomes the output-row map
        phase("R");
    }
    // prolongation: rows = fine points; U is stored by coarse column, so it was regrouped by fine row (<= 3 per row)
    {
        const int nf = l.n;
        if ((rc = d_pbeg.alloc(h, nf)) || (rc = d_pend.alloc(h, nf))) return rc;
        hipLaunchKernelGGL(gmgs::ell3_ptr, dim3((nf + 255) / 256), dim3(256), 0, h->stream, e3.cnt, nf, d_pbeg.p, d_pend.p);
        gmgs::RowFilter fp{rows_k, d_old2new_c.p, nullptr, nullptr, 0, 0};
        if ((rc = device_build_sell(h, l.P, d_pbeg.p, d_pend.p, e3.col, e3.val, fp, nullptr, l.n_pad, 1, nullptr, nullptr, d_err))) return rc;
        l.P.nnz_real = h->U[k].nnz();
    }
    // ---- 16-bit column codes of level 0's operator and transfers for the fine-level kernels (kernels.hip.hpp; gmg_config::fine_col16 = 0: 32-bit indices only): 2 of
    // the 12 bytes of an entry less to read per launch.  First with 8 windows of 8 192 columns per slice (meshes); an operator that
    // leaves more than an eighth of its slices uncovered that way (kNN graphs: the 64 rows of a slice reach into every colour class)
    // is coded again with 32 windows of 2 048, and keeps its 32-bit indices alone if that does not cover it either.
    DevSell* c16_ops[3] = {&l.Aoff, &l.R, &l.P};
    const char* c16_keys[3] = {"col16_l0", "col16_R_l0", "col16_P_l0"};
    int c16_failed[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // per operator: uncovered slices, 1 + index of the last of them; [6 + i]: slices of another width than the mean
    DevTmp<int> d_c16;
    const bool c16 = k == 0 && h->cfg.fine_col16 != 0;      // (a blocked level 0 too: its residual-check and transfer kernels are the colour-major level's)
    const int c16_test_fail = h->dbg_col16_uncovered;         // gmg_debug_set (gravomg_hip_internal.h): N > 0 every N-th slice "uncovered", N < 0 the first -N
    int c16_wexp[3] = {-1, -1, -1};
    auto c16_launch = [&](int i, int nw) -> int {
        DevSell& op = *c16_ops[i];
        if (op.win_base) { (void)dev_free(op.win_base); op.win_base = nullptr; }
        if (!op.col16) HIPCHK(dev_malloc((void**)&op.col16, sizeof(unsigned) * (size_t)op.stored));
        HIPCHK(dev_malloc((void**)&op.win_base, sizeof(int) * (size_t)op.n_slices * nw));
        HIPCHK(hipMemsetAsync(d_c16.p + 2 * i, 0, 2 * sizeof(int), h->stream));
        HIPCHK(hipMemsetAsync(d_c16.p + 6 + i, 0, sizeof(int), h->stream));
        const int64_t per_slice = op.stored / ((int64_t)op.n_slices * 64);
        const int w_exp = (per_slice >= 1 && per_slice <= 63 && per_slice * op.n_slices * 64 == op.stored && h->cfg.uniform_slices) ? (int)per_slice : -1;
        c16_wexp[i] = w_exp;
        const int* a_ptr = i == 0 ? dA.ptr : (const int*)nullptr;
        const int* n2o = i == 0 ? rows_k : (const int*)nullptr;
        if (nw == 8) hipLaunchKernelGGL(gmgs::compress_cols<8>, dim3((op.n_slices + 3) / 4), dim3(256), 0, h->stream, op.slice_ptr, op.col, a_ptr, n2o, op.val, op.n_slices, c16_test_fail, op.col16, op.win_base, d_c16.p + 2 * i, w_exp, d_c16.p + 6 + i);
        else hipLaunchKernelGGL(gmgs::compress_cols<32>, dim3((op.n_slices + 3) / 4), dim3(256), 0, h->stream, op.slice_ptr, op.col, a_ptr, n2o, op.val, op.n_slices, c16_test_fail, op.col16, op.win_base, d_c16.p + 2 * i, w_exp, d_c16.p + 6 + i);
        op.c16_dbits = nw == 8 ? 13 : 11;
        return GMG_OK;
    };
    // usable as it is?  uncovered slices all within the first 1/16 of the numbering (tiny colour classes come first): codes from the slice
    // behind them on, no per-slice test in the kernels (mode 1); else at most 1/8 of the slices uncovered: flagged one by one (mode 2)
    auto c16_mode_of = [&](int i) { const DevSell& op = *c16_ops[i]; return c16_failed[2 * i + 1] <= op.n_slices / 16 ? 1 : (c16_failed[2 * i] <= op.n_slices / 8 ? 2 : 0); };
    if (k == 0) for (const char* key : c16_keys) h->timing[key] = 0.0;
    if (c16) {
        if ((rc = d_c16.alloc(h, 9))) return rc;
        for (int i = 0; i < 3; ++i) {
            DevSell& op = *c16_ops[i];
            if (op.stored <= 0 || op.n_slices <= 0 || op.n_slices >= (1 << 24) || (i == 0 && op.lpr != 1)) continue;
            if ((rc = c16_launch(i, 8))) return rc;
        }
        HIPCHK(hipMemcpyAsync(c16_failed, d_c16.p, 9 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));      // uploads from the orderings' (pageable) arrays are done
    if (c16) {
        bool again = false;
        for (int i = 0; i < 3; ++i)
            if (c16_ops[i]->col16 && c16_mode_of(i) == 0) { if ((rc = c16_launch(i, 32))) return rc; again = true; }
        if (again) {
            HIPCHK(hipMemcpyAsync(c16_failed, d_c16.p, 9 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        for (int i = 0; i < 3; ++i) {
            DevSell& op = *c16_ops[i];
            h->timing[std::string(c16_keys[i]) + "_failed_slices"] = c16_failed[2 * i];
            if (!op.col16) { h->timing[std::string(c16_keys[i]) + "_windows"] = 0; continue; }
            op.c16_mode = c16_mode_of(i);
            op.c16_from = c16_failed[2 * i + 1];
            op.uniform_w = (c16_wexp[i] > 0 && c16_failed[6 + i] == 0) ? c16_wexp[i] : 0;
            h->timing[std::string(c16_keys[i]) + "_uniform_width"] = op.uniform_w;
            if (op.c16_mode == 0) { (void)dev_free(op.col16); (void)dev_free(op.win_base); op.col16 = nullptr; op.win_base = nullptr; op.c16_dbits = 13; op.c16_from = 0; op.uniform_w = 0; }
            else h->timing[c16_keys[i]] = 1.0;
            h->timing[std::string(c16_keys[i]) + "_windows"] = op.col16 ? (op.c16_dbits == 13 ? 8 : 32) : 0;
            h->timing[std::string(c16_keys[i]) + "_mode"] = op.c16_mode;
        }
        // a fine level whose three operators (10 B per stored entry) fit the memory-side cache beside the vectors is read with ordinary loads
        // (DevSell::resident -> the kernels' C16 template argument, c16_sel): 722 k vertices 95 MB, 3 M vertices 390 MB
        int64_t bytes = 0;
        for (int i = 0; i < 3; ++i) bytes += (int64_t)c16_ops[i]->stored * 10;
        for (int i = 0; i < 3; ++i) c16_ops[i]->resident = bytes <= ((int64_t)160 << 20);
        h->timing["fine_operators_resident"] = bytes <= ((int64_t)160 << 20) ? 1.0 : 0.0;
    }
    phase("P");
    return GMG_OK;
}

// The value arrays of level k's operator layouts rewritten from Level::dA (same sparsity pattern as when they were
// built, new values): the fill kernels of device_layout_level without its counting, scanning and allocating.
int device_refill_level(gmg_handle h, int k, int* d_err) {
    Level& l = h->lv[k];
    int rc;
    if (!l.dA.ptr || !l.d_old2new || !l.d_new2old || !l.diag) return fail(h, GMG_ERR_STATE, "level layout cannot be refilled");
    const DevCsr& dA = l.dA;
    // with the source maps of the layouts (made beside them): plain gathers, no ranking of the columns of every row again
    // (3 M vertices: 2.2 -> ~0.3 ms over the levels)
    const bool ep_ok = !l.ord.blocked || (l.use_ep && l.src_ee && l.src_ep);
    if (l.src_A && l.src_diag && ep_ok) {
        auto gather = [&](const int* src, double* out, int64_t cnt) {
            if (cnt > 0) hipLaunchKernelGGL(gmgs::refill_values, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, src, (const double*)dA.val, cnt, out);
        };
        gather(l.src_A, l.Aoff.val, l.Aoff.stored);
        hipLaunchKernelGGL(gmgs::refill_diag, dim3((l.n_pad + 255) / 256), dim3(256), 0, h->stream, (const int*)l.src_diag, (const int*)l.d_new2old, (const double*)dA.val, l.n_pad, l.diag, d_err);
        if (l.ord.blocked) { gather(l.src_ee, l.ee_val, l.ee_nnz); gather(l.src_ep, l.ep_val, l.ep_nnz); }
        return GMG_OK;
    }
    gmgs::RowFilter f{l.d_new2old, l.d_old2new, nullptr, nullptr, 0, 1};
    if ((rc = device_build_sell(h, l.Aoff, dA.ptr, dA.ptr + 1, dA.idx, dA.val, f, nullptr, l.n_pad, l.Aoff.lpr, nullptr, l.diag, d_err, true))) return rc;
    if (l.ord.blocked) {
        if (!l.d_blk_of_row || !l.d_blk_begin) return fail(h, GMG_ERR_STATE, "level layout cannot be refilled");
        gmgs::RowFilter fin{l.d_new2old, l.d_old2new, l.d_blk_of_row, l.d_blk_begin, 1, 1};
        gmgs::RowFilter fout{l.d_new2old, l.d_old2new, l.d_blk_of_row, l.d_blk_begin, 2, 1};
        if (l.use_ep) {
            gmgs::RowFilter fe{l.d_new2old, l.d_old2new, l.d_blk_of_row, l.d_blk_begin, 3, 1};
            gmgs::RowFilter fl{l.d_new2old, l.d_old2new, l.d_blk_of_row, l.d_blk_begin, 4, 1};
            const dim3 gr((l.n_pad + 255) / 256);
            launch_csr_fill_plain(h, l, dA, fe, fl, d_err);
            return GMG_OK;
        }
        if (l.use_bcsr) launch_csr_fill(h, l, dA, fout, l.d_blk_of_row, d_err);
        if ((rc = device_build_sell(h, l.Ain, dA.ptr, dA.ptr + 1, dA.idx, dA.val, fin, nullptr, l.n_pad, l.Ain.lpr, &l.ain_col16, nullptr, d_err, true)) ||
            (rc = device_build_sell(h, l.Aout, dA.ptr, dA.ptr + 1, dA.idx, dA.val, fout, nullptr, l.n_pad, l.Aout.lpr, nullptr, nullptr, d_err, true))) return rc;
    }
    return GMG_OK;
}

// true when every diagonal entry is positive and no stored off-diagonal entry is (threaded; stops at the first offender)
bool stieltjes_signs(int n, const int* ptr, const int* idx, const double* val, int threads) {
    std::atomic<int> bad{0};
    parallel_ranges(n, threads, [&](int lo, int hi, int) {
        for (int j = lo; j < hi; ++j) {
            if ((j & 1023) == 0 && bad.load(std::memory_order_relaxed)) return;
            bool diag = false;
            for (int p = ptr[j]; p < ptr[j + 1]; ++p) {
                if (idx[p] == j) { diag = val[p] > 0.0; if (!diag) { bad.store(1, std::memory_order_relaxed); return; } }
                else if (val[p] > 0.0) { bad.store(1, std::memory_order_relaxed); return; }
            }
            if (!diag) { bad.store(1, std::memory_order_relaxed); return; }
        }
    }, 1 << 14);
    return bad.load() == 0;
}

// Shall level 0 of this system run the block-hybrid sweep?  block_from_level = 0: yes (asked for).  Otherwise gmg_config::block_fine decides:
// the 64-row entry-parallel sweep must be the one that would run, the hierarchy must supply the blocks (cluster order), the rows must be
// long enough for the colour-major sweep to fall apart into many small launches (>= 9 entries per row on average: kNN graphs 10-14,
// triangle meshes 7), and the signs must make the block-hybrid sweep a regular splitting (stieltjes_signs).
constexpr double kFineBlockMinRow = 9.0;
bool fine_level_blocked(gmg_handle h, int n, const int* colptr, const int* rowidx, const double* val) {
    const gmg_config& c = h->cfg;
    if (c.smoother != GMG_SMOOTHER_MULTICOLOR_GS || c.block_rows <= 0 || h->L <= 0) return false;
    if (c.block_from_level <= 0) return true;
    if (h->part_world > 1) return false;      // the partitioned multi-GPU cycle exchanges per colour: level 0 stays colour-major (gmg_dist_partition)
    if (!c.block_fine || c.block_rows != 64 || !c.block_ep || !c.block_csr || c.reorder_fine == 0) return false;
    if ((double)colptr[n] < kFineBlockMinRow * (double)n) return false;
    return stieltjes_signs(n, colptr, rowidx, val, c.host_threads);
}

// One threaded pass over a compressed pattern: 0 = canonical (indices in range, strictly ascending inside each outer
// vector), 1 = in range but unsorted or with duplicates, 2 = an index out of range / a non-monotone pointer array.
int inspect_pattern(int n_outer, int n_inner, const int* ptr, const int* idx, int threads) {
    if (ptr[0] != 0) return 2;
    std::vector<int> worst(std::max(threads, 1) + 1, 0);
    parallel_ranges(n_outer, threads, [&](int lo, int hi, int t) {
        int w = 0;
        for (int j = lo; j < hi && w < 2; ++j) {
            if (ptr[j + 1] < ptr[j]) { w = 2; break; }
            int prev = -1;
            for (int p = ptr[j]; p < ptr[j + 1]; ++p) {
                const int i = idx[p];
                if (i < 0 || i >= n_inner) { w = 2; break; }
                if (i <= prev) w = 1;
                prev = i;
            }
        }
        worst[std::min(t, (int)worst.size() - 1)] = w;
    });
    int w = 0;
    for (int v : worst) w = std::max(w, v);
    return w;
}

// Sorted, duplicate-free copy of a compressed matrix (duplicates are summed, like Eigen's setFromTriplets / scipy's
// sum_duplicates): what the engine requires of the LHS, made here when the caller's storage is not canonical.
Compressed canonical_copy(int n_outer, int n_inner, const int* ptr, const int* idx, const double* val, int threads) {
    Compressed out;
    out.n_outer = n_outer; out.n_inner = n_inner;
    std::vector<int> cnt((size_t)n_outer + 1, 0);
    std::vector<std::vector<std::pair<int, double>>> cols(n_outer);
    parallel_ranges(n_outer, threads, [&](int lo, int hi, int) {
        for (int j = lo; j < hi; ++j) {
            auto& c = cols[j];
            c.reserve(ptr[j + 1] - ptr[j]);
            for (int p = ptr[j]; p < ptr[j + 1]; ++p) c.emplace_back(idx[p], val[p]);
            std::stable_sort(c.begin(), c.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
            size_t w = 0;
            for (size_t r = 0; r < c.size(); ++r) {
                if (w > 0 && c[w - 1].first == c[r].first) c[w - 1].second += c[r].second;
                else c[w++] = c[r];
            }
            c.resize(w);
            cnt[j + 1] = (int)w;
        }
    });
    for (int j = 0; j < n_outer; ++j) cnt[j + 1] += cnt[j];
    out.ptr = cnt;
    out.idx.resize(cnt[n_outer]); out.val.resize(cnt[n_outer]);
    parallel_ranges(n_outer, threads, [&](int lo, int hi, int) {
        for (int j = lo; j < hi; ++j) {
            int q = out.ptr[j];
            for (auto& e : cols[j]) { out.idx[q] = e.first; out.val[q] = e.second; ++q; }
        }
    });
    return out;
}

// 2 x 64-bit FNV-1a style digest of the LHS sparsity pattern (threaded; chunk digests combined in order)
void pattern_key(int n, const int* colptr, const int* rowidx, int threads, uint64_t key[2]) {
    const int64_t nnz = colptr[n];
    (void)threads;
    const int T = 32;       // fixed: the digest depends on the chunking
    std::vector<uint64_t> part((size_t)T * 2, 0);
    // two indices per step on four independent multiply chains (a single chain at one index per step ran at ~2 ns per
    // index: 3 ms for the 24 M indices of the 3 M-vertex system on 16 threads)
    auto digest = [](const int* p, int64_t cnt, uint64_t seed) {
        const uint64_t K = 1099511628211ull;
        uint64_t h0 = 1469598103934665603ull ^ seed, h1 = h0 ^ 0xa0761d6478bd642full, h2 = h0 ^ 0xe7037ed1a0b428dbull, h3 = h0 ^ 0x8ebc6af09c88c6e3ull;
        int64_t i = 0;
        for (; i + 8 <= cnt; i += 8) {
            uint64_t w[4];
            std::memcpy(w, p + i, 32);
            h0 = (h0 ^ w[0]) * K; h0 ^= h0 >> 29;
            h1 = (h1 ^ w[1]) * K; h1 ^= h1 >> 29;
            h2 = (h2 ^ w[2]) * K; h2 ^= h2 >> 29;
            h3 = (h3 ^ w[3]) * K; h3 ^= h3 >> 29;
        }
        for (; i < cnt; ++i) { h0 ^= (uint32_t)p[i]; h0 *= K; h0 ^= h0 >> 29; }
        h0 = (h0 ^ h1) * K; h0 = (h0 ^ h2) * K; h0 = (h0 ^ h3) * K;
        return h0;
    };
    parallel_ranges(T, T, [&](int t0, int t1, int) {
        for (int t = t0; t < t1; ++t) {
            int64_t lo = nnz * t / T, hi = nnz * (t + 1) / T;
            part[2 * t] = digest(rowidx + lo, hi - lo, 0x9e3779b97f4a7c15ull * (t + 1));
            int64_t plo = (int64_t)(n + 1) * t / T, phi = (int64_t)(n + 1) * (t + 1) / T;
            part[2 * t + 1] = digest(colptr + plo, phi - plo, 0xc2b2ae3d27d4eb4full * (t + 1));
        }
    }, 1);
    key[0] = 1469598103934665603ull ^ (uint64_t)n; key[1] = 0x84222325cbf29ce4ull ^ (uint64_t)nnz;
    for (int t = 0; t < T; ++t) { key[0] = (key[0] ^ part[2 * t]) * 1099511628211ull; key[1] = (key[1] ^ part[2 * t + 1]) * 1099511628211ull; }
}

constexpr int kNormBlocks = 2048;      // residual-norm partial sums: 8 blocks per CU, grid-stride

inline int grid_for(int n_slices) {
    int g = (n_slices + gmgk::kWavesPerBlock - 1) / gmgk::kWavesPerBlock;
    return (g + 7) / 8 * 8;    // multiple of 8 so the XCD swizzle is a bijection
}

inline int norm_grid(int n_slices) {
    int g = (n_slices + gmgk::kNormWaves - 1) / gmgk::kNormWaves;
    return (g + 7) / 8 * 8;
}

}  // namespace
