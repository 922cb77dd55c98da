// engine_cycle.hip.hpp -- part of libgravomg_hip.so's single translation unit (included by engine.hip, in this order:
// engine_state, engine_setup, engine_cycle).  Kernel launch helpers and the V-cycle legs.
#pragma once

namespace {

// ---- launch helpers (all on h->stream; column chunks of <= 4) ----------------------------------------

#define DISPATCH_D(dc, ...)              \
    switch (dc) {                        \
        case 1: { constexpr int D = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int D = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int D = 3; __VA_ARGS__; } break; \
        default: { constexpr int D = 4; __VA_ARGS__; } break; \
    }

// level 0 with 16-bit column codes (DevSell::col16) runs the C16 instantiation of its kernels
#define DISPATCH_C16(mode, ...)          \
    switch (mode) {                      \
        case 1: { constexpr int C16 = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int C16 = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int C16 = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int C16 = 4; __VA_ARGS__; } break; \
        default: { constexpr int C16 = 0; __VA_ARGS__; } break; \
    }

// Precision selector: the fp64 arrays, or their fp32 twins (same layout, same index arrays).
template <class T> struct Prec;
template <> struct Prec<double> {
    static const double* val(const DevSell& s) { return s.val; }
    static const double* diag(const Level& l) { return l.diag; }
    static const double* bcval(const Level& l) { return l.bc_val; }
    static const double* epval(const Level& l) { return l.ep_val; }
    static const double* eeval(const Level& l) { return l.ee_val; }
    static double* x(Level& l) { return l.x; }
    static double* b(Level& l) { return l.b; }
    static double* r(Level& l) { return l.r; }
    static double* tmp(Level& l) { return l.tmp; }
};
template <> struct Prec<float> {
    static const float* val(const DevSell& s) { return s.val32; }
    static const float* diag(const Level& l) { return l.diag32; }
    static const float* bcval(const Level& l) { return l.bc_val32; }
    static const float* epval(const Level& l) { return l.ep_val32; }
    static const float* eeval(const Level& l) { return l.ee_val32; }
    static float* x(Level& l) { return l.x32; }
    static float* b(Level& l) { return l.b32; }
    static float* r(Level& l) { return l.r32; }
    static float* tmp(Level& l) { return l.tmp32; }
};

// The last colour launch of the cycle's last level-0 sweep also forms its rows' share of the residual check (gs_color_norm) when
// the cycle being enqueued ends with one (h->fuse_norm_type, set by vcycle_legs): fp64, one group of <= 4 columns, >= 2 colours.
// Its partial sums go behind the ones the norm kernel will write for the rows of the other colours (launch_norm).
template <class T>
bool fold_norm(gmg_handle, Level&, int, bool, int, int) { return false; }
template <>
bool fold_norm<double>(gmg_handle h, Level& l, int d, bool last_launch, int sb, int se) {
    if (!last_launch || h->fuse_norm_type < 0 || &l != &h->lv[0] || d > 4 || l.ord.n_colors < 2 || se != l.Aoff.n_slices) return false;
    const int type = h->fuse_norm_type;
    const double* w = type == 1 ? h->d_minv : (type == 2 ? h->d_mass : nullptr);
    const int nblk = grid_for(se - sb), first = norm_grid(sb);
    if ((size_t)(first + nblk) > (size_t)h->partial_blocks) return false;
    DISPATCH_D(d, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::gs_color_norm<D, C16>), dim3(nblk), dim3(gmgk::kBlock), 0, h->stream, l.Aoff.slice_ptr,
                                     l.Aoff.col, l.Aoff.val, l.diag, l.b, l.x, l.n_pad, sb, se, h->cfg.gs_omega, w, h->d_partials + (size_t)first * 2 * d,
                                     l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
    h->fuse_norm_blocks = nblk;
    return true;
}

// ... and the last colour launch of the level-0 PRE-smoothing writes the residual of its rows (gs_color_residual) when the way
// down asks for it (h->fuse_res_out, set by enqueue_down); launch_spmv then covers the slices in front of that colour only.
template <class T>
bool fold_residual(gmg_handle, Level&, int, bool, int, int) { return false; }
template <>
bool fold_residual<double>(gmg_handle h, Level& l, int d, bool last_launch, int sb, int se) {
    if (!last_launch || !h->fuse_res_out || &l != &h->lv[0] || d > 4 || l.ord.n_colors < 2 || se != l.Aoff.n_slices || sb <= 0) return false;
    if (h->il_r0 && d > 1) {
        DISPATCH_D(d, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::gs_color_residual<D, C16, (D > 1)>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream,
                                         l.Aoff.slice_ptr, l.Aoff.col, l.Aoff.val, l.diag, l.b, l.x, h->fuse_res_out, l.n_pad, sb, se, h->cfg.gs_omega,
                                         l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
    } else {
        DISPATCH_D(d, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::gs_color_residual<D, C16>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream,
                                         l.Aoff.slice_ptr, l.Aoff.col, l.Aoff.val, l.diag, l.b, l.x, h->fuse_res_out, l.n_pad, sb, se, h->cfg.gs_omega,
                                         l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
    }
    h->fuse_res_from = sb;
    return true;
}

inline bool polled(gmg_handle h);

// one colour launch of a sweep over level l (columns c0 .. c0 + dc); go: see gmgk::gs_color
template <class T>
void launch_gs_color(gmg_handle h, Level& l, int c0, int dc, int sb, int se, const int* go = nullptr) {
    const int ld = l.n_pad;
    const bool fine = &l == &h->lv[0];
    const T omega = fine ? (T)h->cfg.gs_omega : (T)1.0;       // over-relaxation on the finest level only (gmg_config::gs_omega)
    T* x = Prec<T>::x(l);
    const T* b = Prec<T>::b(l);
    if (fine && l.Aoff.c16_mode != 0) {           // FINE = 1 + C16 (c16_sel: 1 / 2 streamed, 3 / 4 a fine level that stays on the chip)
        DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::gs_color<T, D, C16 + 1>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream,
                                          l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), b + (size_t)c0 * ld,
                                          x + (size_t)c0 * ld, ld, sb, se, 1, omega, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg(), (const unsigned long long*)nullptr, go)));
    } else if (fine) {
        DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_color<T, D, 1>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream,
                                          l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), b + (size_t)c0 * ld,
                                          x + (size_t)c0 * ld, ld, sb, se, 1, omega, (const unsigned*)nullptr, (const int*)nullptr, 0, (const unsigned long long*)nullptr, go));
    } else {
        DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_color<T, D, 0>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream,
                                          l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), b + (size_t)c0 * ld,
                                          x + (size_t)c0 * ld, ld, sb, se, 1, omega));
    }
}

inline int first_color(const Level& l) {
    for (int c = 0; c < l.ord.n_colors; ++c)
        if (l.ord.color_begin[c + 1] / 64 > l.ord.color_begin[c] / 64) return c;
    return -1;
}

// Head of the next cycle: may the first colour launch of the level-0 pre-smoothing be enqueued ahead of the solve loop's decision?  Stream
// launches with the polled check (the reduction publishes the decision), fp64, colour-major level 0 with a colour class in front of the one the
// residual rides on, one group of columns, the whole system on this device.
inline bool head_eligible(gmg_handle h, int d) {
    return h->cfg.speculate_head && polled(h) && !h->cfg.inner_precision && h->cfg.smoother == GMG_SMOOTHER_MULTICOLOR_GS && h->cfg.pre_iters > 0 && h->L >= 1 &&
           !h->lv[0].ord.blocked && h->lv[0].ord.n_colors >= 2 && first_color(h->lv[0]) >= 0 && first_color(h->lv[0]) < h->lv[0].ord.n_colors - 1 && d >= 1 && d <= 4 &&
           !h->partitioned && h->part_world <= 1 && h->d_watch && !h->prof_on;
}
inline void enqueue_head(gmg_handle h, int d) {
    Level& l = h->lv[0];
    const int c = first_color(l);
    launch_gs_color<double>(h, l, 0, d, l.ord.color_begin[c] / 64, l.ord.color_begin[c + 1] / 64, reinterpret_cast<const int*>(h->d_watch + 1));
    h->head_enqueued = true;
    h->timing["heads_enqueued"] += 1.0;
}

template <class T>
void launch_gs_sweeps(gmg_handle h, Level& l, int d, int iters) {
    const bool fine = &l == &h->lv[0];
    // (the first launch of this cycle is already in the stream: enqueue_head)
    const bool skip_head = fine && h->head_enqueued;
    if (fine) h->head_enqueued = false;
    const int c_head = skip_head ? first_color(l) : -1;
    for (int it = 0; it < iters; ++it)
        for (int c0 = 0; c0 < d; c0 += 4) {
            int dc = std::min(4, d - c0);
            for (int c = 0; c < l.ord.n_colors; ++c) {
                int sb = l.ord.color_begin[c] / 64, se = l.ord.color_begin[c + 1] / 64;
                if (se <= sb) continue;
                if (it == 0 && c0 == 0 && c == c_head) continue;
                if (fold_norm<T>(h, l, d, it == iters - 1 && c == l.ord.n_colors - 1, sb, se)) continue;
                if (fold_residual<T>(h, l, d, it == iters - 1 && c == l.ord.n_colors - 1, sb, se)) continue;
                launch_gs_color<T>(h, l, c0, dc, sb, se);
            }
        }
}

template <class T>
void launch_jacobi_sweeps(gmg_handle h, Level& l, int d, int iters) {
    const int ld = l.n_pad;
    T* in = Prec<T>::x(l); T* out = Prec<T>::tmp(l);
    const T* b = Prec<T>::b(l);
    for (int it = 0; it < iters; ++it) {
        for (int c0 = 0; c0 < d; c0 += 4) {
            int dc = std::min(4, d - c0);
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::jacobi_sweep<T, D>), dim3(grid_for(l.Aoff.n_slices)), dim3(gmgk::kBlock), 0, h->stream,
                                              l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), b + (size_t)c0 * ld,
                                              (in ? in + (size_t)c0 * ld : nullptr), out + (size_t)c0 * ld, ld, l.Aoff.n_slices, (T)h->cfg.jacobi_omega, 1));
        }
        std::swap(in, out);
    }
    if (in != Prec<T>::x(l)) (void)hipMemcpyAsync(Prec<T>::x(l), in, sizeof(T) * (size_t)ld * d, hipMemcpyDeviceToDevice, h->stream);
}

// Launch geometry of the entry-parallel block sweep (kernels.hip.hpp::gs_block_ep): one block per workgroup.  (The kernel can also run as
// persistent workgroups, grid < vgrid; measured in round 4 -- 8 .. 20 workgroups per compute unit -- the same or worse: profiles/r04/b_*.)
inline int ep_persistent_grid(gmg_handle, int vgrid) { return vgrid; }

// Does the entry-parallel block sweep of level l STREAM its operator (non-temporal loads)?  Yes when the operator's chunks (12 B per explicit,
// 10 B per lower entry in fp64) are more than the memory-side cache (256 MB on MI355X) holds beside the cycle's vectors between two of the
// launches that read them: level 0 of a point cloud (300 MB) -- measured 0.634 ms per cycle streamed, 0.699 with ordinary loads; the 506 k-row
// level 1 of the 3 M mesh (76 MB, five readers per cycle): 140 us per cycle streamed, 125 not.
constexpr int64_t kEpResidentBytes = (int64_t)128 << 20;
inline bool ep_streams(const Level& l) { return l.ee_nnz * 12 + l.ep_nnz * 10 > kEpResidentBytes; }

// block-hybrid Gauss-Seidel: one launch per sweep, ping-pong between x and tmp
// One block-hybrid sweep in -> out over blocks [b0, b0 + nb) of a blocked level (in == nullptr: the iterate is the zero vector).
// The kernels find their rows through blk_begin[block]: a sub-range is the same launch on offset block tables.
template <class T>
void launch_block_sweep_range(gmg_handle h, Level& l, int d, const T* in, T* out, int b0, int nb, const int* begin_table = nullptr,
                              const int* ncolors_table = nullptr, T* out_il = nullptr) {
    const int ld = l.n_pad;
    const T* b = Prec<T>::b(l);
    // (begin_table / ncolors_table: an explicit list of nb blocks instead of a range -- entry-parallel sweep only, which reads
    // nothing but the first row of its block from the table)
    const int* blk_begin = begin_table ? begin_table : l.d_blk_begin + b0;
    constexpr bool table_always = false;
    const int* blk_ncolors = ncolors_table ? ncolors_table : l.d_blk_ncolors + b0;
    if (nb <= 0) return;
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        if (l.use_ep) {
            const int vgrid = (nb + 7) / 8 * 8;         // multiple of 8: the kernel's XCD-aware block map is a bijection onto [0, vgrid)
            const int grid = ep_persistent_grid(h, vgrid);
            if (ep_streams(l)) {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_block_ep<T, D, true>), dim3(grid), dim3(64), gmgk::ep_lds_bytes<T>(D, l.ep_cap_e, l.ep_cap_l), h->stream,
                                                  ((begin_table || table_always) ? blk_begin : (const int*)nullptr), blk_ncolors, l.d_row_color, l.ep_ptr, l.ep_col, Prec<T>::epval(l), l.ee_ptr, l.ee_col,
                                                  Prec<T>::eeval(l), Prec<T>::diag(l), b + (size_t)c0 * ld, (in ? in + (size_t)c0 * ld : nullptr),
                                                  out + (size_t)c0 * ld, ld, l.ep_cap_e, l.ep_cap_l, nb, b0, vgrid, out_il));
            } else {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_block_ep<T, D, false>), dim3(grid), dim3(64), gmgk::ep_lds_bytes<T>(D, l.ep_cap_e, l.ep_cap_l), h->stream,
                                                  ((begin_table || table_always) ? blk_begin : (const int*)nullptr), blk_ncolors, l.d_row_color, l.ep_ptr, l.ep_col, Prec<T>::epval(l), l.ee_ptr, l.ee_col,
                                                  Prec<T>::eeval(l), Prec<T>::diag(l), b + (size_t)c0 * ld, (in ? in + (size_t)c0 * ld : nullptr),
                                                  out + (size_t)c0 * ld, ld, l.ep_cap_e, l.ep_cap_l, nb, b0, vgrid, out_il));
            }
        } else if (l.use_bcsr && d > 1) {
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_block_csrout<T, D, (D == 1 ? 32 : 24)>), dim3(nb), dim3(64),
                                              (size_t)l.bc_cap * (sizeof(T) + sizeof(int)) + (size_t)D * 64 * sizeof(T), h->stream, blk_begin,
                                              blk_ncolors, l.d_row_color, l.Ain.slice_ptr, l.ain_col16, Prec<T>::val(l.Ain), l.bc_ptr, l.bc_col,
                                              Prec<T>::bcval(l), Prec<T>::diag(l), b + (size_t)c0 * ld, (in ? in + (size_t)c0 * ld : nullptr), out + (size_t)c0 * ld,
                                              ld, l.bc_cap));
        } else if (l.Ain.lpr == 4) {
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_block4<T, D, 8>), dim3(nb), dim3(4 * h->cfg.block_rows), 0, h->stream, blk_begin,
                                              blk_ncolors, l.d_row_color, l.Ain.slice_ptr, l.ain_col16, Prec<T>::val(l.Ain), l.Aout.slice_ptr,
                                              l.Aout.col, Prec<T>::val(l.Aout), Prec<T>::diag(l), b + (size_t)c0 * ld, (in ? in + (size_t)c0 * ld : nullptr),
                                              out + (size_t)c0 * ld, ld));
        } else {
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_block<T, D, (D == 1 ? 32 : 24)>), dim3(nb), dim3(h->cfg.block_rows), 0, h->stream, blk_begin,
                                              blk_ncolors, l.d_row_color, l.Ain.slice_ptr, l.ain_col16, Prec<T>::val(l.Ain), l.Aout.slice_ptr,
                                              l.Aout.col, Prec<T>::val(l.Aout), Prec<T>::diag(l), b + (size_t)c0 * ld, (in ? in + (size_t)c0 * ld : nullptr),
                                              out + (size_t)c0 * ld, ld));
        }
    }
}

template <class T>
void launch_block_sweeps(gmg_handle h, Level& l, int d, int iters, bool from_zero = false) {
    const int ld = l.n_pad;
    const int nb = l.ord.n_blocks();
    // from_zero: the iterate is the zero vector (the coarse correction's initial guess, multigrid_solver.cpp:1072-1073): the
    // first sweep gets no input vector -- it neither reads x nor gathers the off-block couplings (all zero) -- and the
    // caller skips the memset
    T* in = from_zero ? nullptr : Prec<T>::x(l); T* out = Prec<T>::tmp(l);
    const T* before_last = nullptr;      // the iterate the last sweep started from (nullptr: the zero vector)
    bool last_known = false;
    T* il = (T*)h->il_sweep_out;          // (enqueue_up: the last sweep also writes the level's x as an interleaved multi-vector)
    h->il_sweep_out = nullptr; h->il_sweep_done = false;
    int it0 = 0;
    const bool first_fused = from_zero && h->first_sweep_fused;
    h->first_sweep_fused = false;                  // (consumed by the level it was set for -- or void)
    if (first_fused) {                             // the restriction into this level already ran the first sweep (launch_restrict_sweep0): its result is in tmp
        before_last = nullptr; last_known = true;
        in = out; out = Prec<T>::x(l);
        it0 = 1;
    }
    for (int it = it0; it < iters; ++it) {
        const bool with_il = il && it == iters - 1 && l.use_ep && d > 1 && d <= 4;
        launch_block_sweep_range<T>(h, l, d, in, out, 0, nb, nullptr, nullptr, with_il ? il : nullptr);
        if (with_il) h->il_sweep_done = true;
        before_last = in; last_known = true;
        if (it == 0 && from_zero) { in = out; out = Prec<T>::x(l); }      // the result of sweep 1 is in tmp; ping-pong from there
        else std::swap(in, out);
    }
    if (in != Prec<T>::x(l)) (void)hipMemcpyAsync(Prec<T>::x(l), in, sizeof(T) * (size_t)ld * d, hipMemcpyDeviceToDevice, h->stream);
    // what residual_delta_ep needs: the last sweep went before_last -> x (the copy above, if any, does not touch before_last)
    h->sweep_prev_valid = last_known && before_last != (const T*)Prec<T>::x(l);
    h->sweep_prev = (const void*)before_last;
}

// r = b - A x of a blocked level with the unpadded block storage, straight after its block sweeps (h->sweep_prev: the iterate the
// last sweep started from): the sweep's explicit part applied to x_old - x_new (kernels.hip.hpp::residual_delta_ep)
// (begin_table / nb_list: an explicit list of blocks -- a rank's blocks of a partitioned level -- instead of all of them)
template <class T>
bool launch_residual_delta(gmg_handle h, Level& l, int d, T* r, const int* begin_table = nullptr, int nb_list = 0) {
    if (!l.use_ep || !h->sweep_prev_valid || h->cfg.smoother == GMG_SMOOTHER_JACOBI) return false;
    h->sweep_prev_valid = false;
    const int ld = l.n_pad, nb = begin_table ? nb_list : l.ord.n_blocks();
    if (nb <= 0) return true;
    const int grid = (nb + 7) / 8 * 8;
    const T* x_old = (const T*)h->sweep_prev;
    const T* x_new = Prec<T>::x(l);
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        if (ep_streams(l)) {
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::residual_delta_ep<T, D, true>), dim3(grid), dim3(64), gmgk::ep_lds_bytes<T>(0, l.ep_cap_e, 0), h->stream, begin_table,
                                              l.ee_ptr, l.ee_col, Prec<T>::eeval(l), (x_old ? x_old + (size_t)c0 * ld : nullptr), x_new + (size_t)c0 * ld,
                                              r + (size_t)c0 * ld, ld, nb));
        } else {
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::residual_delta_ep<T, D, false>), dim3(grid), dim3(64), gmgk::ep_lds_bytes<T>(0, l.ep_cap_e, 0), h->stream, begin_table,
                                              l.ee_ptr, l.ee_col, Prec<T>::eeval(l), (x_old ? x_old + (size_t)c0 * ld : nullptr), x_new + (size_t)c0 * ld,
                                              r + (size_t)c0 * ld, ld, nb));
        }
    }
    return true;
}

// true when smoothing level l from a zero iterate needs no materialised zero vector (block sweeps, at least one of them)
inline bool smooth_from_zero_ok(gmg_handle h, const Level& l, int iters) {
    return iters > 0 && h->cfg.smoother != GMG_SMOOTHER_JACOBI && l.ord.blocked;
}

template <class T = double>
void launch_smooth(gmg_handle h, Level& l, int d, int iters, bool from_zero = false) {
    if (iters <= 0) return;
    if (h->cfg.smoother == GMG_SMOOTHER_JACOBI) launch_jacobi_sweeps<T>(h, l, d, iters);
    else if (l.ord.blocked) launch_block_sweeps<T>(h, l, d, iters, from_zero);
    else launch_gs_sweeps<T>(h, l, d, iters);
}

// y = A x (mode 0) or y = b - A x (mode 1)
template <class T, int LPR>
void launch_spmv_lpr(gmg_handle h, Level& l, int d, int mode, const T* b, const T* x, T* y, int n_slices = -1, bool y_il = false) {
    const int ld = l.n_pad;
    if (n_slices < 0) n_slices = l.Aoff.n_slices;
    if (y_il && mode == 1 && LPR == 1 && d > 1 && d <= 4) {        // residual as an interleaved multi-vector (level 0, d > 1: what the restriction gathers from)
        DISPATCH_D(d, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::spmv_full<T, D, 1, 1, C16, (D > 1)>), dim3(grid_for(n_slices)), dim3(gmgk::kBlock), 0, h->stream,
                                         l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), b, x, y, ld, 0, n_slices, 1, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
        return;
    }
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        if (LPR == 1 && l.Aoff.c16_mode != 0) {           // level 0 with 16-bit column codes
            if (mode == 1) {
                DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::spmv_full<T, D, 1, 1, C16>), dim3(grid_for(n_slices)), dim3(gmgk::kBlock), 0, h->stream,
                                                  l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), b + (size_t)c0 * ld, x + (size_t)c0 * ld,
                                                  y + (size_t)c0 * ld, ld, 0, n_slices, 1, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
            } else {
                DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::spmv_full<T, D, 0, 1, C16>), dim3(grid_for(n_slices)), dim3(gmgk::kBlock), 0, h->stream,
                                                  l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), (const T*)nullptr, x + (size_t)c0 * ld,
                                                  y + (size_t)c0 * ld, ld, 0, n_slices, 1, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
            }
        } else if (mode == 1) {
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::spmv_full<T, D, 1, LPR>), dim3(grid_for(n_slices)), dim3(gmgk::kBlock), 0, h->stream,
                                              l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), b + (size_t)c0 * ld, x + (size_t)c0 * ld,
                                              y + (size_t)c0 * ld, ld, 0, n_slices, 1));
        } else {
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::spmv_full<T, D, 0, LPR>), dim3(grid_for(n_slices)), dim3(gmgk::kBlock), 0, h->stream,
                                              l.Aoff.slice_ptr, l.Aoff.col, Prec<T>::val(l.Aoff), Prec<T>::diag(l), (const T*)nullptr, x + (size_t)c0 * ld,
                                              y + (size_t)c0 * ld, ld, 0, n_slices, 1));
        }
    }
}
template <class T>
void launch_spmv(gmg_handle h, Level& l, int d, int mode, const T* b, const T* x, T* y, int n_slices = -1, bool y_il = false) {
    if (l.Aoff.lpr == 4) launch_spmv_lpr<T, 4>(h, l, d, mode, b, x, y, n_slices);
    else launch_spmv_lpr<T, 1>(h, l, d, mode, b, x, y, n_slices, y_il);
}

// coarse.b = U^T fine.r
template <class T, int LPR>
void launch_restrict_lpr(gmg_handle h, Level& fine, Level& coarse, int d, const T* src, T* dst, bool src_il = false) {
    if (src_il && d > 1 && d <= 4) {
        DISPATCH_D(d, DISPATCH_C16(fine.R.c16_sel(), hipLaunchKernelGGL((gmgk::transfer<T, D, 0, LPR, C16, (D > 1)>), dim3(grid_for(fine.R.n_slices)), dim3(gmgk::kBlock), 0,
                                         h->stream, fine.R.slice_ptr, fine.R.col, Prec<T>::val(fine.R), fine.R.row_of, src, fine.n_pad, dst, coarse.n_pad, 0, fine.R.n_slices, 1,
                                         fine.R.col16, fine.R.win_base, fine.R.c16_arg())));
        return;
    }
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        DISPATCH_D(dc, DISPATCH_C16(fine.R.c16_sel(), hipLaunchKernelGGL((gmgk::transfer<T, D, 0, LPR, C16>), dim3(grid_for(fine.R.n_slices)), dim3(gmgk::kBlock), 0,
                                          h->stream, fine.R.slice_ptr, fine.R.col, Prec<T>::val(fine.R), fine.R.row_of, src + (size_t)c0 * fine.n_pad, fine.n_pad,
                                          dst + (size_t)c0 * coarse.n_pad, coarse.n_pad, 0, fine.R.n_slices, 1, fine.R.col16, fine.R.win_base, fine.R.c16_arg())));
    }
}
template <class T>
void launch_restrict(gmg_handle h, Level& fine, Level& coarse, int d, const T* src, T* dst, bool src_il = false) {
    if (fine.R.lpr == 4) launch_restrict_lpr<T, 4>(h, fine, coarse, d, src, dst, src_il);
    else launch_restrict_lpr<T, 1>(h, fine, coarse, d, src, dst, src_il);
}

// coarse.b = U^T fine.r AND the first pre-sweep of the coarse level from the zero iterate, in one launch (kernels.hip.hpp::restrict_sweep0), where the
// layouts allow it: the coarse level runs the entry-parallel block sweep on blocks b = rows 64 b .., its restriction has the quad layout with
// 64-row sorting windows (a workgroup's four slices = one block), and the level's pre-smoothing starts from zero.  The coarse level's
// launch_block_sweeps then starts with its second sweep (h->first_sweep_fused).
// which fused form the restriction into `coarse` takes: 0 none, 1 restrict_sweep0 (entry-parallel sweep), 2 gs_block4<.., FR = true> (quad layout)
template <class T>
int restrict_sweep0_kind(gmg_handle h, const Level& fine, const Level& coarse, int d, bool src_il) {
    if (!h->cfg.fuse_restrict_sweep || h->cfg.smoother == GMG_SMOOTHER_JACOBI || h->cfg.pre_iters <= 0 || d > 4) return 0;
    if (!coarse.ord.blocked || fine.R.lpr != 4 || !(h->cfg.restrict_sigma == 0 || h->cfg.restrict_sigma == 64)) return 0;
    if (coarse.use_ep) return (coarse.n_pad == 64 * coarse.ord.n_blocks() && fine.R.n_slices == 4 * coarse.ord.n_blocks()) ? 1 : 0;      // (block b = rows 64 b .. 64 b + 63)
    // quad layout: a wave of the block sweep covers the 16 rows of one restriction slice; plain 32-bit restriction, column-major source
    if (!(coarse.use_bcsr && d > 1) && coarse.Ain.lpr == 4 && fine.R.c16_mode == 0 && !src_il && fine.R.n_slices * 16 == coarse.n_pad) return 2;
    return 0;
}
template <class T>
void launch_restrict_sweep0(gmg_handle h, Level& fine, Level& coarse, int d, const T* src, bool src_il, int kind) {
    const int nb = coarse.ord.n_blocks();
    if (kind == 2) {
        DISPATCH_D(d, hipLaunchKernelGGL((gmgk::gs_block4<T, D, 8, true>), dim3(nb), dim3(4 * h->cfg.block_rows), 0, h->stream, coarse.d_blk_begin, coarse.d_blk_ncolors,
                                         coarse.d_row_color, coarse.Ain.slice_ptr, coarse.ain_col16, Prec<T>::val(coarse.Ain), coarse.Aout.slice_ptr, coarse.Aout.col,
                                         Prec<T>::val(coarse.Aout), Prec<T>::diag(coarse), (const T*)nullptr, (const T*)nullptr, Prec<T>::tmp(coarse), coarse.n_pad,
                                         fine.R.slice_ptr, fine.R.col, Prec<T>::val(fine.R), fine.R.row_of, src, fine.n_pad, Prec<T>::b(coarse)));
        h->first_sweep_fused = true;
        return;
    }
    const int vgrid = (nb + 7) / 8 * 8;
    const size_t lds_sweep = gmgk::ep_lds_bytes<T>(d, 0, coarse.ep_cap_l);
    const size_t lds = lds_sweep + (size_t)d * 64 * sizeof(T);
    T* bdst = Prec<T>::b(coarse);
    T* xdst = Prec<T>::tmp(coarse);               // (the result of a from-zero first sweep lives in tmp: launch_block_sweeps)
#define GMG_RS0(XI_)                                                                                                                                        \
    DISPATCH_D(d, DISPATCH_C16(fine.R.c16_sel(), {                                                                                                           \
        if (ep_streams(coarse))                                                                                                                              \
            hipLaunchKernelGGL((gmgk::restrict_sweep0<T, D, true, C16, (XI_ && D > 1)>), dim3(vgrid), dim3(256), lds, h->stream, fine.R.slice_ptr, fine.R.col,   \
                               Prec<T>::val(fine.R), fine.R.row_of, src, fine.n_pad, fine.R.col16, fine.R.win_base, fine.R.c16_arg(), bdst, coarse.d_blk_ncolors,      \
                               coarse.d_row_color, coarse.ep_ptr, coarse.ep_col, Prec<T>::epval(coarse), Prec<T>::diag(coarse), xdst, coarse.n_pad, nb, vgrid,       \
                               (int)lds_sweep);                                                                                                              \
        else                                                                                                                                                 \
            hipLaunchKernelGGL((gmgk::restrict_sweep0<T, D, false, C16, (XI_ && D > 1)>), dim3(vgrid), dim3(256), lds, h->stream, fine.R.slice_ptr, fine.R.col,  \
                               Prec<T>::val(fine.R), fine.R.row_of, src, fine.n_pad, fine.R.col16, fine.R.win_base, fine.R.c16_arg(), bdst, coarse.d_blk_ncolors,      \
                               coarse.d_row_color, coarse.ep_ptr, coarse.ep_col, Prec<T>::epval(coarse), Prec<T>::diag(coarse), xdst, coarse.n_pad, nb, vgrid,       \
                               (int)lds_sweep);                                                                                                              \
    }))
    if (src_il && d > 1) { GMG_RS0(1); } else { GMG_RS0(0); }
#undef GMG_RS0
    h->first_sweep_fused = true;
}

// fine.x += U coarse.x   (U has <= 3 entries per row: always one lane per row)
template <class T>
void launch_prolong_add(gmg_handle h, Level& fine, Level& coarse, int d, const T* src, T* dst, bool src_il = false) {
    if (src_il && d > 1 && d <= 4) {
        DISPATCH_D(d, DISPATCH_C16(fine.P.c16_sel(), hipLaunchKernelGGL((gmgk::transfer<T, D, 1, 1, C16, (D > 1)>), dim3(grid_for(fine.P.n_slices)), dim3(gmgk::kBlock), 0,
                                         h->stream, fine.P.slice_ptr, fine.P.col, Prec<T>::val(fine.P), (const int*)nullptr, src, coarse.n_pad, dst, fine.n_pad, 0, fine.P.n_slices, 1,
                                         fine.P.col16, fine.P.win_base, fine.P.c16_arg())));
        return;
    }
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        DISPATCH_D(dc, DISPATCH_C16(fine.P.c16_sel(), hipLaunchKernelGGL((gmgk::transfer<T, D, 1, 1, C16>), dim3(grid_for(fine.P.n_slices)), dim3(gmgk::kBlock), 0,
                                          h->stream, fine.P.slice_ptr, fine.P.col, Prec<T>::val(fine.P), (const int*)nullptr, src + (size_t)c0 * coarse.n_pad,
                                          coarse.n_pad, dst + (size_t)c0 * fine.n_pad, fine.n_pad, 0, fine.P.n_slices, 1, fine.P.col16, fine.P.win_base, fine.P.c16_arg())));
    }
}

// Polled completion.  Slot 0: residual-norm sums in h_norm; slot 1: the coarsest right-hand side in h_pinned.
// The polled words and the data they announce are written by a kernel and read by the host while the stream is still
// busy: that needs fine-grained (coherent) pinned memory, requested explicitly -- what hipHostMallocDefault gives depends on
// the ROCm version and on HIP_HOST_COHERENT.
constexpr unsigned kPolledHostFlags = hipHostMallocCoherent | hipHostMallocMapped;
inline bool polled(gmg_handle h) { return !h->cfg.use_graph && h->poll && h->h_flag; }

int wait_flag(gmg_handle h, int slot) {
    const unsigned long long want = h->flag_seq[slot];
    auto t0 = clk::now();
    double next_query_ms = 20.0;
    for (unsigned spin = 1;; ++spin) {
        if (__atomic_load_n(h->h_flag + 8 * slot, __ATOMIC_ACQUIRE) == want) return GMG_OK;
        // safety net only (the runtime's query is not free and may touch the queue): a finished stream implies the data
        // is visible, flag or not; an error state must not spin for ever
        if ((spin & 4095u) == 0 && ms_since(t0) > next_query_ms) {
            hipError_t q = hipStreamQuery(h->stream);
            if (q == hipSuccess) {
                // the stream is idle, so the data is visible -- but the flag never showed up while it ran: this host does not
                // see device writes to pinned memory in flight.  Stop polling on this handle (copy + synchronise instead)
                if (__atomic_load_n(h->h_flag + 8 * slot, __ATOMIC_ACQUIRE) != want) { h->poll = false; h->timing["poll_disabled"] = 1.0; }
                return GMG_OK;
            }
            if (q != hipErrorNotReady) HIPCHK(q);
            next_query_ms += 20.0;
        }
        __builtin_ia32_pause();
    }
}

// the result of the last launch_norm / launch_residual_to_f32 is in h_norm when this returns
int wait_norm(gmg_handle h) {
    if (polled(h)) return wait_flag(h, 0);
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMG_OK;
}

// the reduction of one group of <= 4 columns: into h_norm + flag when polled, into d_norm (+ a copy later) otherwise
void launch_reduce(gmg_handle h, int nblk, int dc, int c0, bool last) {
    if (polled(h)) {
        gmgk::SolveWatch watch{nullptr, nullptr, nullptr, 0.0, 0, 0, 0, 0};
        if (h->watch_active && last && c0 == 0 && h->d_watch)         // (one group of columns: the kernel sees every sum the decision needs)
            watch = gmgk::SolveWatch{reinterpret_cast<int*>(h->d_watch + 1), h->h_flag + 1, h->d_watch, h->watch_tol, h->watch_mode, h->watch_type, dc, h->watch_cycles_done};
        hipLaunchKernelGGL(gmgk::reduce_partials, dim3(1), dim3(gmgk::kReduceBlock), 0, h->stream, h->d_partials, nblk, 2 * dc, h->h_norm + 2 * c0,
                           last ? h->h_flag : nullptr, last ? ++h->flag_seq[0] : 0ull, (int)EnvSwitches::get().publish_fenced, watch);
    }
    else
        hipLaunchKernelGGL(gmgk::reduce_partials, dim3(1), dim3(gmgk::kReduceBlock), 0, h->stream, h->d_partials, nblk, 2 * dc, h->d_norm + 2 * c0,
                           (unsigned long long*)nullptr, 0ull, 0);
}

// sums of w r^2 / w b^2 per column -> h_norm[2*d] (after wait_norm)
int launch_norm(gmg_handle h, int d, int type) {
    Level& l = h->lv[0];
    const double* w = type == 1 ? h->d_minv : (type == 2 ? h->d_mass : nullptr);
    // (rows of the last colour: already summed by the cycle's last colour launch when it was folded in, fold_norm)
    const int folded = h->fuse_norm_blocks;
    h->fuse_norm_blocks = 0;
    const int n_slices = folded ? l.ord.color_begin[l.ord.n_colors - 1] / 64 : l.Aoff.n_slices;
    const int nblk = norm_grid(n_slices);                // one slice per wave, like the residual SpMV; one partial per block of 16
    const bool poll = polled(h);
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::residual_norm_slices<D, 0, C16>), dim3(nblk), dim3(gmgk::kNormWaves * 64), 0, h->stream,
                                          l.Aoff.slice_ptr, l.Aoff.col, l.Aoff.val, l.diag, l.b + (size_t)c0 * l.n_pad, l.x + (size_t)c0 * l.n_pad, w,
                                          l.n_pad, n_slices, (float*)nullptr, h->d_partials, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
        launch_reduce(h, nblk + folded, dc, c0, c0 + 4 >= d);
    }
    if (!poll) HIPCHK(hipMemcpyAsync(h->h_norm, h->d_norm, sizeof(double) * 2 * d, hipMemcpyDeviceToHost, h->stream));
    return GMG_OK;
}

double norm_from_sums(const double* s, int d, int type) {
    if (type == 3) {
        double t = 0.0;
        for (int c = 0; c < d; ++c) t += s[2 * c];
        return std::sqrt(t);
    }
    double out = 0.0;
    for (int c = 0; c < d; ++c) {
        double v = type == 0 ? std::sqrt(s[2 * c]) / std::sqrt(s[2 * c + 1]) : std::sqrt(s[2 * c] / s[2 * c + 1]);
        if (c == 0 || v > out) out = v;
    }
    return out;
}

int ensure_vectors(gmg_handle h, int d) {
    if (d <= h->dcap) return GMG_OK;
    drop_graphs(h);
    unbind_level0(h);
    for (auto& l : h->lv) {
        for (double** p : {&l.x, &l.b, &l.r, &l.tmp}) {
            if (*p) { (void)dev_free(*p); *p = nullptr; }
            if (p == &l.tmp && h->cfg.smoother != GMG_SMOOTHER_JACOBI && !l.ord.blocked) continue;
            size_t bytes = sizeof(double) * (size_t)l.n_pad * d;
            HIPCHK(dev_malloc((void**)p, bytes));
            HIPCHK(hipMemsetAsync(*p, 0, bytes, h->stream));
        }
        for (float** p : {&l.x32, &l.b32, &l.r32, &l.tmp32}) {
            if (*p) { (void)dev_free(*p); *p = nullptr; }
            if (!h->cfg.inner_precision) continue;
            if (p == &l.tmp32 && h->cfg.smoother != GMG_SMOOTHER_JACOBI && !l.ord.blocked) continue;
            size_t bytes = sizeof(float) * (size_t)l.n_pad * d;
            HIPCHK(dev_malloc((void**)p, bytes));
            HIPCHK(hipMemsetAsync(*p, 0, bytes, h->stream));
        }
    }
    Level& c = h->lv[h->L];
    size_t need = (size_t)c.n_pad * d * 2;
    if (need > h->pinned_cap) {
        if (h->h_pinned) (void)sync_hipHostFree(h->h_pinned);
        HIPCHK(hipHostMalloc((void**)&h->h_pinned, sizeof(double) * need, kPolledHostFlags));
        h->pinned_cap = need;
    }
    if (h->h_norm) (void)sync_hipHostFree(h->h_norm);
    HIPCHK(hipHostMalloc((void**)&h->h_norm, sizeof(double) * 2 * d, kPolledHostFlags));
    if (!h->h_flag) {
        HIPCHK(hipHostMalloc((void**)&h->h_flag, 256, kPolledHostFlags));
        std::memset(h->h_flag, 0, 256);
    }

    if (h->d_norm) (void)dev_free(h->d_norm);
    HIPCHK(dev_malloc((void**)&h->d_norm, sizeof(double) * 2 * d));
    if (!h->d_watch) {
        HIPCHK(dev_malloc((void**)&h->d_watch, 16));
        HIPCHK(hipMemsetAsync(h->d_watch, 0, 16, h->stream));
    }
    h->dcap = d;
    h->loaded_d = 0;
    return GMG_OK;
}

int ensure_stage(gmg_handle h, size_t n_doubles) {
    if (n_doubles <= h->stage_cap) return GMG_OK;
    if (h->d_stage) (void)dev_free(h->d_stage);
    HIPCHK(dev_malloc((void**)&h->d_stage, sizeof(double) * n_doubles));
    h->stage_cap = n_doubles;
    return GMG_OK;
}

// Pinned, double-buffered host staging: the caller's (pageable) vectors are copied in with a few threads and moved by
// DMA at PCIe speed (a pageable hipMemcpy of 24 MB runs at a fraction of that: 3 vectors cost ~18 ms per solve at 3 M).
int ensure_host_stage(gmg_handle h, size_t n_doubles) {
    if (n_doubles <= h->h_stage_cap) return GMG_OK;
    for (int i = 0; i < 2; ++i) {
        if (h->h_stage[i]) (void)sync_hipHostFree(h->h_stage[i]);
        h->h_stage[i] = nullptr;
        HIPCHK(hipHostMalloc((void**)&h->h_stage[i], sizeof(double) * n_doubles, hipHostMallocDefault));
        if (!h->h_stage_ev[i]) HIPCHK(hipEventCreateWithFlags(&h->h_stage_ev[i], hipEventDisableTiming));
    }
    h->h_stage_cap = n_doubles;
    return GMG_OK;
}

inline void threaded_copy(double* dst, const double* src, size_t n, int threads) {
    const int T = (int)std::min<size_t>(std::max(1, std::min(threads, 16)), n / 65536 + 1);
    if (T <= 1) { std::memcpy(dst, src, sizeof(double) * n); return; }
    parallel_ranges(T, T, [&](int t0, int t1, int) {
        for (int t = t0; t < t1; ++t) {
            size_t lo = n * t / T, hi = n * (t + 1) / T;
            std::memcpy(dst + lo, src + lo, sizeof(double) * (hi - lo));
        }
    }, 1);
}

// Vectors cross PCIe in chunks so that the host's copy between the caller's (pageable) array and the pinned staging buffer overlaps
// the DMA of the neighbouring chunk: 24 MB took ~0.35 ms of copy + ~0.5 ms of DMA one after the other.
constexpr int kXferChunks = 6;
inline size_t xfer_chunk(size_t cnt) { return std::max<size_t>((cnt + kXferChunks - 1) / kXferChunks, (size_t)1 << 17); }

// The pinned staging holds ONE column (n doubles per buffer, two buffers): an n x d block crosses column by column, alternating the buffers, so
// a first n x 3 solve pays no page-locking of a bigger staging area (16-40 ms at 3 M vertices: the demos' call, core.cpp:68-72) and the
// host copy of column c + 1 overlaps the DMA of column c.
// host natural n x d  ->  device numbering (level k) buffer
int to_device(gmg_handle h, int k, const double* src, int d, double* dst) {
    Level& l = h->lv[k];
    const size_t n = (size_t)l.n, cnt = n * d;
    int rc = ensure_stage(h, cnt);
    if (rc) return rc;
    if ((rc = ensure_host_stage(h, n))) return rc;
    const size_t ch = xfer_chunk(n);
    for (int c = 0; c < d; ++c) {
        const int f = h->h_stage_flip;
        h->h_stage_flip ^= 1;
        HIPCHK(hipEventSynchronize(h->h_stage_ev[f]));          // the previous DMA out of this staging buffer is done
        for (size_t off = 0; off < n; off += ch) {
            const size_t len = std::min(ch, n - off);
            threaded_copy(h->h_stage[f] + off, src + c * n + off, len, h->cfg.host_threads);
            HIPCHK(hipMemcpyAsync(h->d_stage + c * n + off, h->h_stage[f] + off, sizeof(double) * len, hipMemcpyHostToDevice, h->stream));
        }
        HIPCHK(hipEventRecord(h->h_stage_ev[f], h->stream));
    }
    hipLaunchKernelGGL(gmgk::permute_in, dim3((l.n_pad + 255) / 256), dim3(256), 0, h->stream, h->d_stage, l.n, l.d_new2old, dst, l.n_pad, l.n_pad, d);
    // d_stage is reused by the next call: order is guaranteed by the single stream
    return GMG_OK;
}

int to_host(gmg_handle h, int k, const double* src, int d, double* dst) {
    Level& l = h->lv[k];
    const size_t n = (size_t)l.n, cnt = n * d;
    int rc = ensure_stage(h, cnt);
    if (rc) return rc;
    if ((rc = ensure_host_stage(h, n))) return rc;
    HIPCHK(hipEventSynchronize(h->h_stage_ev[0]));
    HIPCHK(hipEventSynchronize(h->h_stage_ev[1]));
    hipLaunchKernelGGL(gmgk::permute_out, dim3((l.n_pad + 255) / 256), dim3(256), 0, h->stream, src, l.n_pad, l.n_pad, l.d_new2old, h->d_stage, l.n, d);
    const size_t ch = xfer_chunk(n);
    constexpr int kEvPerBuf = (int)(sizeof(h->h_chunk_ev) / sizeof(h->h_chunk_ev[0])) / 2;
    static_assert(kEvPerBuf >= kXferChunks, "one event per chunk and staging buffer");
    auto issue = [&](int c) -> int {                             // column c on the wire, into buffer c & 1, an event behind every chunk
        const int f = c & 1;
        int nch = 0;
        for (size_t off = 0; off < n; off += ch, ++nch) {
            const size_t len = std::min(ch, n - off);
            hipEvent_t& ev = h->h_chunk_ev[f * kEvPerBuf + nch];
            if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIPCHK(hipMemcpyAsync(h->h_stage[f] + off, h->d_stage + c * n + off, sizeof(double) * len, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipEventRecord(ev, h->stream));
        }
        return GMG_OK;
    };
    if ((rc = issue(0))) return rc;
    for (int c = 0; c < d; ++c) {
        if (c + 1 < d && (rc = issue(c + 1))) return rc;         // (its buffer was copied out two columns ago)
        const int f = c & 1;
        int nch = 0;
        for (size_t off = 0; off < n; off += ch, ++nch) {
            const size_t len = std::min(ch, n - off);
            HIPCHK(hipEventSynchronize(h->h_chunk_ev[f * kEvPerBuf + nch]));
            threaded_copy(dst + c * n + off, h->h_stage[f] + off, len, h->cfg.host_threads);
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMG_OK;
}

// ---- V-cycle legs --------------------------------------------------------------------------------------

// gmg_profile_cycle: an event at every boundary between the legs of a cycle (2 L + 3 of them: before each level's way down, after the
// last one, at the start of the way up, after each level's way up, after the residual check)
inline void prof_mark(gmg_handle h) {
    if (!h->prof_on) return;
    if ((size_t)h->prof_n >= h->prof_ev.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return; h->prof_ev.push_back(e); }
    (void)hipEventRecord(h->prof_ev[h->prof_n++], h->stream);
}

// coarse.b = U_k^T r_k (:1069) into level k + 1 -- together with that level's first pre-sweep where the layouts allow the two in one launch
// (restrict_sweep0_kind; the level's launch_block_sweeps then starts with its second sweep: h->first_sweep_fused)
template <class T>
void restrict_into(gmg_handle h, int k, int d, bool il) {
    Level& l = h->lv[k];
    Level& c = h->lv[k + 1];
    const int fused = (k + 1 < h->L && smooth_from_zero_ok(h, c, h->cfg.pre_iters)) ? restrict_sweep0_kind<T>(h, l, c, d, il) : 0;
    if (fused) launch_restrict_sweep0<T>(h, l, c, d, Prec<T>::r(l), il, fused);
    else launch_restrict<T>(h, l, c, d, Prec<T>::r(l), Prec<T>::b(c), il);
}

template <class T = double>
void enqueue_down(gmg_handle h, int d, int k0 = 0) {
    const int L = h->L;
    if (k0 == 0) h->first_sweep_fused = false;     // (set by restrict_into for the level that follows; a caller that starts lower restricted into level k0 itself)
    for (int k = k0; k < L; ++k) {
        Level& l = h->lv[k];
        prof_mark(h);
        const bool from_zero = k > 0 && smooth_from_zero_ok(h, l, h->cfg.pre_iters);      // eps.setZero (:1072-1073) folded into the first sweep
        if (k > 0 && !from_zero) (void)hipMemsetAsync(Prec<T>::x(l), 0, sizeof(T) * (size_t)l.n_pad * d, h->stream);
        // level 0, fp64: the last colour launch of the pre-smoothing also writes the residual of its rows (fold_residual)
        constexpr bool no_fold = false;
        // d = 2 .. 4: the level-0 residual is written as an INTERLEAVED multi-vector (n x d row-major) -- only the restriction reads it, and a
        // gathered child then costs one cache line instead of d
        constexpr bool no_il = false;
        const bool il = k == 0 && d > 1 && d <= 4 && !no_il && l.Aoff.lpr == 1;
        h->il_r0 = il;
        if (k == 0 && sizeof(T) == 8 && !no_fold) h->fuse_res_out = h->lv[0].r;
        h->fuse_res_from = 0;
        h->sweep_prev_valid = false;
        launch_smooth<T>(h, l, d, h->cfg.pre_iters, from_zero);                                     // :1063
        h->fuse_res_out = nullptr;
        const int res_slices = (k == 0 && h->fuse_res_from > 0) ? h->fuse_res_from : -1;
        h->fuse_res_from = 0;
        // :1066.  Straight after block sweeps on the unpadded block storage the residual comes from the sweep's explicit part alone
        if (!(k > 0 && launch_residual_delta<T>(h, l, d, Prec<T>::r(l))))
            launch_spmv<T>(h, l, d, 1, Prec<T>::b(l), Prec<T>::x(l), Prec<T>::r(l), res_slices, il);
        // :1069 (+ the first pre-sweep of level k + 1 where the layouts allow the two in one launch)
        restrict_into<T>(h, k, d, il);
        h->il_r0 = false;
    }
    prof_mark(h);
}

template <class T = double>
void enqueue_up(gmg_handle h, int d, int k0 = 0) {
    prof_mark(h);
    for (int k = h->L - 1; k >= k0; --k) {
        Level& l = h->lv[k];
        // d = 2 .. 4: level 1's last post-sweep left a second, interleaved copy of its x in its (idle) residual vector: the prolongation into
        // level 0 gathers a parent's d values from one cache line
        const bool il = k == 0 && h->il_sweep_done;
        h->il_sweep_done = false;
        const T* src = il ? Prec<T>::r(h->lv[k + 1]) : Prec<T>::x(h->lv[k + 1]);
        launch_prolong_add<T>(h, l, h->lv[k + 1], d, src, Prec<T>::x(l), il);     // :1082
        constexpr bool no_il = false;
        if (k == 1 && k0 == 0 && d > 1 && d <= 4 && !no_il && h->cfg.post_iters > 0 && l.ord.blocked && l.use_ep && h->cfg.smoother != GMG_SMOOTHER_JACOBI)
            h->il_sweep_out = (void*)Prec<T>::r(l);
        launch_smooth<T>(h, l, d, h->cfg.post_iters);                                                // :1085
        h->il_sweep_out = nullptr;
        prof_mark(h);
    }
}

inline void launch_cvt(gmg_handle h, const double* src, float* dst, size_t n) {
    hipLaunchKernelGGL(gmgk::cvt_f64_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, dst, (int64_t)n);
}
inline void launch_cvt(gmg_handle h, const float* src, double* dst, size_t n) {
    hipLaunchKernelGGL(gmgk::cvt_f32_to_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, dst, (int64_t)n);
}

#ifndef GMG_SYMV_ROWS
#define GMG_SYMV_ROWS 0
#endif
inline int symv_env(const char* name) { const char* v = std::getenv(name); return v ? std::atoi(v) : 0; }
inline int symv_rows(int n, int d) {
    static const int forced = symv_env("GMG_SYMV_ROWS");      // (measurement: scripts/symv_sweep.py)
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    if (GMG_SYMV_ROWS > 0) return GMG_SYMV_ROWS;       // (A/B builds)
    // with 16-byte loads (dense_symv_v2; profiles/r06/symv_rows_loads_sweep.txt): one row per wave while the level is small (n_L = 1 929: 11.8 us at
    // d = 1, 13.5 at d = 3; 2 968: 16.3), two beyond (4 046 at d = 3: 32.6; 6 005: 57 at d = 3, 50 at d = 1); four rows only pay with 8-byte loads
    (void)d;
    return n >= 3000 ? 2 : 1;
}
// strides of 64 columns a lane loads per trip (dense_symv's U)
inline int symv_strides(int n, int d, int rows) {
    static const int forced = symv_env("GMG_SYMV_STRIDES");
    if (forced == 2 || forced == 4 || forced == 8) return forced;
    (void)n; (void)d; (void)rows;
    return 4;                                          // (2 / 4 / 8 measured: within the noise of each other except four rows x two strides at 4 046: +3 us)
}

// e = A_L^{-1} rc with the dense inverse (always applied in fp64; the fp32 cycle converts around it)
template <class T = double>
void enqueue_coarse_device(gmg_handle h, int d) {
    Level& c = h->lv[h->L];
    const size_t cnt = (size_t)c.n_pad * d;
    if (sizeof(T) == 4) launch_cvt(h, c.b32, c.b, cnt);
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        // rows per wave: the rows of a wave share the loads of the vectors (what bounds the product at d = 3 and at n_L beyond the L1's reach); a small
        // level keeps one row per wave -- it needs every wave it can get to cover the memory latency
        const int rows_per_wave = symv_rows(c.n, dc);
        const int per_block = gmgk::kWavesPerBlock * rows_per_wave;
        const dim3 grid((c.n + per_block - 1) / per_block);
        const int strides = symv_strides(c.n, dc, rows_per_wave);
#define GMG_SYMV_LAUNCH(R, U) DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::dense_symv<D, R, U>), grid, dim3(gmgk::kBlock), 0, h->stream, h->d_ainv, c.n, h->ainv_ld, c.b + (size_t)c0 * c.n_pad, c.x + (size_t)c0 * c.n_pad, c.n_pad))
#define GMG_SYMV_U(R) do { if (strides >= 8) { GMG_SYMV_LAUNCH(R, 8); } else if (strides >= 4) { GMG_SYMV_LAUNCH(R, 4); } else { GMG_SYMV_LAUNCH(R, 2); } } while (0)
        static const char* v2_env = std::getenv("GMG_SYMV_V2");
        const bool v2 = v2_env ? std::atoi(v2_env) != 0 : true;
        if (v2 && rows_per_wave <= 2 && (h->ainv_ld & 1) == 0 && (c.n_pad & 1) == 0) {
            if (rows_per_wave == 4) { DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::dense_symv_v2<D, 4, 2>), grid, dim3(gmgk::kBlock), 0, h->stream, h->d_ainv, c.n, h->ainv_ld, c.b + (size_t)c0 * c.n_pad, c.x + (size_t)c0 * c.n_pad, c.n_pad)); }
            else if (rows_per_wave == 2) { DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::dense_symv_v2<D, 2, 2>), grid, dim3(gmgk::kBlock), 0, h->stream, h->d_ainv, c.n, h->ainv_ld, c.b + (size_t)c0 * c.n_pad, c.x + (size_t)c0 * c.n_pad, c.n_pad)); }
            else { DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::dense_symv_v2<D, 1, 2>), grid, dim3(gmgk::kBlock), 0, h->stream, h->d_ainv, c.n, h->ainv_ld, c.b + (size_t)c0 * c.n_pad, c.x + (size_t)c0 * c.n_pad, c.n_pad)); }
            continue;
        }
        if (rows_per_wave == 4) GMG_SYMV_U(4);
        else if (rows_per_wave == 2) GMG_SYMV_U(2);
        else GMG_SYMV_U(1);
#undef GMG_SYMV_U
#undef GMG_SYMV_LAUNCH
    }
    if (sizeof(T) == 4) launch_cvt(h, c.x, c.x32, cnt);
}

// Host coarsest solve (:1075): rc to the host, LDL^T back-substitution per column (fp64), eps back.
// Polled handles split it in two so that the host's answer releases work that is ALREADY in the queue: coarse_host_begin enqueues
// the publication of rc, a stream wait on a host-written word (hipStreamWaitValue64) and the fetch of eps; the caller goes on
// enqueuing the way up (and the residual check); coarse_host_serve then waits for rc, solves, writes eps and the word.  Releasing
// queued work costs ~3 us after the host's store; launching it after the solve costs ~12 us for the first kernel and ~4 us for
// each of the next few (scripts/micro/wait_value.hip).
void coarse_host_solve(gmg_handle h, int d) {
    Level& c = h->lv[h->L];
    const size_t cnt = (size_t)c.n_pad * d;
    double* rc = h->h_pinned;
    double* e = h->h_pinned + cnt;
    auto t0 = clk::now();
    std::memset(e, 0, sizeof(double) * cnt);
    if (h->coarse_work.size() < (size_t)c.n * d) h->coarse_work.resize((size_t)c.n * d);
    h->coarse.solve_multi(rc, (size_t)c.n_pad, e, (size_t)c.n_pad, d, h->coarse_work.data(), h->coarse_helper.get());
    h->timing["coarse_host_ms"] += ms_since(t0);
}

// While one of these is alive the helper threads of the coarsest back-substitution spin (a hand-over costs ~0.2 us instead of a
// wake-up); outside they sleep and single solves run every part of the elimination tree on the calling thread.
// GMG_LDLT_THREADS = threads of a solve including the caller (default: as many as the factor has parts, at most 8); 1: no team.
struct HelperScope {
    gmg_handle h;
    // cols: right-hand sides of the solves to come (their (part, column) jobs are independent: up to 8 threads have work with d = 3)
    explicit HelperScope(gmg_handle hh, int cols = 1) : h(hh) {
        const int env_threads = EnvSwitches::get().ldlt_threads;
        // (every rank of a multi-GPU job keeps its team busy -- and this thread polls -- on the CPUs the job may use: a rank's team
        // is sized to its share of them (cpu_budget() divides by LOCAL_WORLD_SIZE) minus one CPU of slack, a throttled spinning
        // thread costs far more than it saves; ranks started without that variable are counted through the handle's world size)
        const int ranks = (h->dist_ready && EnvSwitches::get().local_world <= 1) ? std::max(1, h->world) : 1;
        const int share = cpu_budget() / ranks;
        int threads = std::min(std::min(h->coarse.parts() * std::max(1, std::min(cols, 4)), 8), share - 1);
        if (env_threads > 0) threads = std::min(threads, env_threads);
        if (h->coarse_device || threads < 2) { h = nullptr; return; }
        if (!h->coarse_helper || h->coarse_helper->helpers() != threads - 1) h->coarse_helper.reset(new SpinTeam(threads - 1));
        h->coarse_helper->stay_near_caller();
        h->coarse_helper->arm();
    }
    ~HelperScope() { if (h) h->coarse_helper->disarm(); }
    HelperScope(const HelperScope&) = delete;
    HelperScope& operator=(const HelperScope&) = delete;
};

// the host part of a pending gate (no-op without one).  The word is written even when waiting for rc failed: the stream must
// not stay blocked.
int coarse_host_serve(gmg_handle h) {
    if (!h->coarse_pending) return GMG_OK;
    h->coarse_pending = false;
    if (!h->coarse_warm) { h->coarse_warm_sink += h->coarse.warm(); h->coarse_warm = true; }      // (the device is busy with the way down)
    const int w = wait_flag(h, 1);
    if (w == GMG_OK) coarse_host_solve(h, h->coarse_pending_d);
    __atomic_store_n(h->h_flag + 16, h->flag_seq[2], __ATOMIC_RELEASE);
    if (h->gate_shared) { h->gate_shared = false; --tl_gates_held; gate_mutex().unlock_shared(); }
    return w;
}

}  // namespace
// an exception crossed a cycle: open a pending gate without an answer (the stream must not stay parked, the shared lock not stay held)
int gate_unwound(gmg_handle h) {
    if (h->coarse_pending) {
        h->coarse_pending = false;
        if (h->h_flag) __atomic_store_n(h->h_flag + 16, h->flag_seq[2], __ATOMIC_RELEASE);
    }
    if (h->gate_shared) { h->gate_shared = false; --tl_gates_held; gate_mutex().unlock_shared(); }
    return GMG_OK;
}
namespace {

template <class T = double>
int coarse_host_begin(gmg_handle h, int d) {
    Level& c = h->lv[h->L];
    const size_t cnt = (size_t)c.n_pad * d;
    double* rc = h->h_pinned;
    double* e = h->h_pinned + cnt;
    if (sizeof(T) == 4) launch_cvt(h, c.b32, c.b, cnt);          // tiny (n_L doubles): convert on the device, ship fp64
    if (polled(h)) {
        hipLaunchKernelGGL(gmgk::publish_to_host, dim3(1), dim3(gmgk::kBlock), 0, h->stream, c.b, rc, (int)cnt, h->h_flag + 8, ++h->flag_seq[1], (int)EnvSwitches::get().publish_fenced);
        const bool gate_off = h->cfg.stream_gate == 0;      // gmg_config::stream_gate
        // The gate is only used once this handle has SEEN a published right-hand side arrive while its stream was still busy: wait_flag's
        // safety net (an idle stream implies visible data) cannot fire behind a gate the host itself has to open, so on a host that does
        // not see in-flight device writes a gated first contact would spin for ever.  The first coarse solve of a handle is ungated.
        if (h->gate_ok && h->gate_proven && !gate_off) {
            // (shared lock first: from here to coarse_host_serve no thread of this process starts a device-wide synchronisation, see gate_mutex)
            if (!h->gate_shared) { gate_mutex().lock_shared(); h->gate_shared = true; ++tl_gates_held; }
            if (hipStreamWaitValue64(h->stream, h->h_flag + 16, ++h->flag_seq[2], hipStreamWaitValueGte, ~0ull) == hipSuccess) {
                // (round 4: letting the prolongation out of the coarsest level gather the answer straight from the pinned host buffer instead of this
                // copy kernel -- one launch less behind the host -- measured 0.6608 vs 0.6582 ms per cycle: the gathers over PCIe cost what the copy costs)
                hipLaunchKernelGGL(gmgk::fetch_from_host, dim3((unsigned)std::min<size_t>(8, (cnt + gmgk::kBlock - 1) / gmgk::kBlock)), dim3(gmgk::kBlock), 0, h->stream,
                                   (const double*)e, c.x, (int)cnt);
                if (sizeof(T) == 4) launch_cvt(h, c.x, c.x32, cnt);
                h->coarse_pending = true; h->coarse_pending_d = d;
                return GMG_OK;
            }
            (void)hipGetLastError();
            if (h->gate_shared) { h->gate_shared = false; --tl_gates_held; gate_mutex().unlock_shared(); }
            h->gate_ok = false; --h->flag_seq[2]; h->timing["gate_disabled"] = 1.0;
        }
        int w = wait_flag(h, 1);
        if (w) return w;
        if (h->poll) h->gate_proven = true;        // the flag showed up by itself (wait_flag clears h->poll otherwise)
        if (!polled(h)) HIPCHK(hipStreamSynchronize(h->stream));
    } else {
        HIPCHK(hipMemcpyAsync(rc, c.b, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    coarse_host_solve(h, d);
    HIPCHK(hipMemcpyAsync(c.x, e, sizeof(double) * cnt, hipMemcpyHostToDevice, h->stream));
    if (sizeof(T) == 4) launch_cvt(h, c.x, c.x32, cnt);
    return GMG_OK;
}

// both halves back to back (callers that have nothing to enqueue in between)
template <class T = double>
int coarse_host_roundtrip(gmg_handle h, int d) {
    int rc = coarse_host_begin<T>(h, d);
    return rc ? rc : coarse_host_serve(h);
}

// Mixed precision: fp64 residual of the current iterate -> fp32 right-hand side of the inner cycle, plus the norm sums
// of that same residual (h_norm after the copy + sync).  type < 0: weights of type 0.
int launch_residual_to_f32(gmg_handle h, int d, int type) {
    Level& l = h->lv[0];
    const double* w = type == 1 ? h->d_minv : (type == 2 ? h->d_mass : nullptr);
    const int nblk = norm_grid(l.Aoff.n_slices);
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::residual_norm_slices<D, 1, C16>), dim3(nblk), dim3(gmgk::kNormWaves * 64), 0, h->stream,
                                          l.Aoff.slice_ptr, l.Aoff.col, l.Aoff.val, l.diag, l.b + (size_t)c0 * l.n_pad, l.x + (size_t)c0 * l.n_pad, w, l.n_pad,
                                          l.Aoff.n_slices, l.b32 + (size_t)c0 * l.n_pad, h->d_partials, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
        launch_reduce(h, nblk, dc, c0, c0 + 4 >= d);
    }
    if (!polled(h)) HIPCHK(hipMemcpyAsync(h->h_norm, h->d_norm, sizeof(double) * 2 * d, hipMemcpyDeviceToHost, h->stream));
    return GMG_OK;
}

enum { G_DOWN = 1, G_UP = 2, G_FULL = 3 };

template <class F>
int run_graph(gmg_handle h, int key, F&& enqueue) {
    if (!h->cfg.use_graph) { enqueue(); return GMG_OK; }
    auto it = h->graphs.find(key);
    if (it == h->graphs.end()) {
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        auto tc = clk::now();
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        enqueue();
        HIPCHK(hipStreamEndCapture(h->stream, &g));
        h->timing["graph_capture_ms"] += ms_since(tc);
        tc = clk::now();
        HIPCHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        h->timing["graph_instantiate_ms"] += ms_since(tc);
        (void)hipGraphDestroy(g);
        it = h->graphs.emplace(key, ge).first;
    }
    HIPCHK(hipGraphLaunch(it->second, h->stream));
    return GMG_OK;
}

// One V-cycle on the resident problem; norm_type >= 0 also enqueues the residual check of that type
// (result in h_norm after the stream is synchronised by the caller).
template <class T>
int vcycle_legs(gmg_handle h, int d, int norm_type, int key_salt) {
    int rc;
    const int nt = norm_type < 0 ? 9 : norm_type;
    constexpr bool mixed = sizeof(T) == 4;
    // the residual check's sums over the rows of the last colour come out of the last colour launch of the post-smoothing
    constexpr bool no_fold = false;
    const bool foldable = !mixed && !no_fold && norm_type >= 0 && h->cfg.post_iters > 0 && h->cfg.smoother == GMG_SMOOTHER_MULTICOLOR_GS && !h->lv[0].ord.blocked;
    // mixed precision: the fp32 cycle starts from a zero guess on the defect b32 = b - A x (already in place), its
    // result is added to the fp64 iterate, and the new defect + its norms are formed in one fp64 pass
    auto head = [&] { if (mixed) (void)hipMemsetAsync(h->lv[0].x32, 0, sizeof(float) * (size_t)h->lv[0].n_pad * d, h->stream); };
    auto tail = [&](int& err) {
        if (mixed) {
            const size_t cnt = (size_t)h->lv[0].n_pad * d;
            hipLaunchKernelGGL(gmgk::add_correction, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->lv[0].x32, h->lv[0].x, (int64_t)cnt);
            err = launch_residual_to_f32(h, d, norm_type);
        } else if (norm_type >= 0) err = launch_norm(h, d, norm_type);
        prof_mark(h);
    };
    if (h->coarse_device) {
        int err = GMG_OK;
        rc = run_graph(h, key_salt + G_FULL * 10000 + d * 10 + nt, [&] {
            head();
            enqueue_down<T>(h, d);
            enqueue_coarse_device<T>(h, d);
            h->fuse_norm_type = foldable ? norm_type : -1;
            enqueue_up<T>(h, d);
            h->fuse_norm_type = -1;
            tail(err);
        });
        return rc ? rc : err;
    }
    if ((rc = run_graph(h, key_salt + G_DOWN * 10000 + d * 10, [&] { head(); enqueue_down<T>(h, d); }))) return rc;
    if ((rc = coarse_host_begin<T>(h, d))) return rc;              // (polled: the way up is enqueued behind a gate the host opens in _serve)
    int err = GMG_OK;
    rc = run_graph(h, key_salt + G_UP * 10000 + d * 10 + nt, [&] {
        h->fuse_norm_type = foldable ? norm_type : -1;
        enqueue_up<T>(h, d);
        h->fuse_norm_type = -1;
        tail(err);
    });
    const int served = coarse_host_serve(h);
    return rc ? rc : (served ? served : err);
}

int vcycle_resident(gmg_handle h, int d, int norm_type) {
    if (h->cfg.inner_precision) return vcycle_legs<float>(h, d, norm_type, 100000);
    return vcycle_legs<double>(h, d, norm_type, 0);
}

int check_level(gmg_handle h, int k, bool allow_coarsest) {
    if (!h->system_ready) return fail(h, GMG_ERR_STATE, "no system set (call gmg_set_system first)");
    if (k < 0 || k > h->L || (!allow_coarsest && k == h->L)) return fail(h, GMG_ERR_INVALID, "level index out of range");
    return GMG_OK;
}

// entry points that read whole level operators: not on a handle that holds one rank's rows of a partitioned system
int check_whole_system(gmg_handle h) {
    if (h->partitioned) return fail(h, GMG_ERR_STATE, "this handle holds one rank's rows of a partitioned system (gmg_dist_partition): only the gmg_p2p_* / gmg_dist_* entry points run on it");
    return GMG_OK;
}

int check_norm_type(gmg_handle h, int type) {
    if (type < 0 || type > 3) return fail(h, GMG_ERR_INVALID, "residual norm type must be 0..3");
    if ((type == 1 || type == 2) && !h->d_mass) return fail(h, GMG_ERR_STATE, "mass matrix not set (gmg_set_mass) but an M-weighted norm was requested");
    return GMG_OK;
}

}  // namespace
