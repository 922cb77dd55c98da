// engine.hip -- libgravomg_hip.so: the C-ABI of include/gravomg_hip.h over the device engine.
//
// One handle = one HIP device, one stream.  The hierarchy (all A_k as SELL-64 + diagonal, all U_k / U_k^T) is built on
// the device once per system (gmg_set_system); a V-cycle is a fixed launch sequence on that stream (smooth -> residual
// -> restrict ... coarse solve ... prolong-add -> smooth), optionally captured into hipGraphs.  The coarsest operator is
// factored on the host (supernodal LDL^T, host_ldlt.hpp); per cycle it is applied as a dense inverse on the device (built on
// the device from that factor; gmg_config::coarse_mode = GMG_COARSE_AUTO, the default) or back-substituted on the host.
//
// One translation unit, in four files: engine_state.hip.hpp (memory pool, level / handle structures, helpers),
// engine_setup.hip.hpp (device-side layout construction, Galerkin products), engine_cycle.hip.hpp (launch helpers,
// V-cycle legs) and this file (the extern "C" entry points, incl. the setup pipeline of gmg_set_system).
//
// Reference call sites this replaces: gravomg/src/multigrid_solver.cpp:1059-1088 (V-cycle),
// :1194-1226 (smoother), :1228-1277 (norms), :1387-1419 (solve loop).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <functional>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <vector>

#if !defined(__HIP_DEVICE_COMPILE__)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#endif

#include "../../include/gravomg_hip.h"
#include "../../include/gravomg_hip_internal.h"
#include "host_hierarchy.hpp"
#include "host_ldlt.hpp"
#include "host_plan.hpp"
#include "host_sparse.hpp"
#include "kernels.hip.hpp"
#include "setup_kernels.hip.hpp"
#include "hierarchy_kernels.hip.hpp"

using namespace gmg;
using clk = std::chrono::steady_clock;
static inline double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

#if !defined(__HIP_DEVICE_COMPILE__)
// Debugging aid: GMG_SEGV_BACKTRACE=1 makes a fatal signal inside the process print the native call stack of the faulting thread
// (symbol + offset; `addr2line -e libgravomg_hip.so` resolves them) before the default action takes over.
namespace {
struct sigaction gmg_prev_action[32];
void gmg_fatal_signal(int sig) {
    void* frames[48];
    const int nf = backtrace(frames, 48);          // (libgcc's unwinder was loaded at install time: no first-use allocation in here)
    const char msg[] = "[gmg] fatal signal, native stack of the faulting thread:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, nf, 2);
    // hand over to whoever was installed before us (the host application's handler, or the default action)
    if (sig > 0 && sig < 32) sigaction(sig, &gmg_prev_action[sig], nullptr); else signal(sig, SIG_DFL);
    raise(sig);
}
struct GmgSignalAid {
    GmgSignalAid() {
        if (!EnvSwitches::get().segv_backtrace) return;
        void* warm[4];
        (void)backtrace(warm, 4);                  // first use loads libgcc_s (dlopen + malloc): not something to do inside a signal handler
        struct sigaction sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sa_handler = gmg_fatal_signal;
        for (int sig : {SIGSEGV, SIGBUS, SIGABRT, SIGFPE}) sigaction(sig, &sa, &gmg_prev_action[sig]);
    }
} gmg_signal_aid;
}  // namespace
#endif

#include "engine_state.hip.hpp"
#include "engine_setup.hip.hpp"
#include "engine_cycle.hip.hpp"
#include "engine_part.hip.hpp"

// =========================================================================================================
// No exception leaves the C-ABI: std::bad_alloc (a 3 M-vertex set-up allocates hundreds of MB on the host), a failed
// thread start, ... become a status code + gmg_last_error.
// (a cycle that was unwound between the two halves of its coarsest solve leaves a gate in its stream and the process-wide shared lock that
// goes with it: the handler opens both -- gate_unwound)
int gate_unwound(gmg_handle h);
#define GMG_CATCH_H                                                                                             \
    catch (const std::exception& e_) { if (h) (void)gate_unwound(h); return h ? fail(h, GMG_ERR_STATE, std::string("exception: ") + e_.what()) : GMG_ERR_STATE; } \
    catch (...) { if (h) (void)gate_unwound(h); return h ? fail(h, GMG_ERR_STATE, "unknown exception") : GMG_ERR_STATE; }
#define GMG_CATCH_0                                              \
    catch (...) { return GMG_ERR_STATE; }

void p2p_release_handle(gmg_handle h);      // engine_dist.hip.hpp

// The order in which the numeric Galerkin pass of level 1 can follow the upload of A_0's values (set_system_impl): coarse row p needs the rows of
// A_0 its children are (U_0's column p, ascending), so the coarse rows are bucketed by their largest child (64 fine rows per bucket, counting
// sort) -- in a locally numbered mesh a tenth of them becomes computable with every tenth of the upload.
static void drop_rap_order(gmg_handle h) {
    h->rap_need.clear(); h->rap_need2.clear();
    if (h->d_rap_order) { (void)sync_hipFree(h->d_rap_order); h->d_rap_order = nullptr; }
    if (h->d_rap_order2) { (void)sync_hipFree(h->d_rap_order2); h->d_rap_order2 = nullptr; }
}
static bool ensure_rap_order(gmg_handle h) {
    if (!h->rap_need.empty() && h->d_rap_order) return true;
    drop_rap_order(h);
    const Compressed& U0 = h->U[0];
    const int nc = U0.n_outer, nf = U0.n_inner;
    if (nc <= 0 || nf <= 0) return false;
    const int nb = (nf + 63) / 64 + 1;                    // (bucket 0: coarse rows without children)
    std::vector<int> start((size_t)nb + 1, 0), order((size_t)nc), bucket((size_t)nc);
    // (the largest child: the maximum over the column -- a caller of the raw C-ABI may hand over columns whose row indices are not ascending)
    for (int p = 0; p < nc; ++p) {
        int big = -1;
        for (int e = U0.ptr[p]; e < U0.ptr[p + 1]; ++e) big = std::max(big, U0.idx[e]);
        bucket[p] = big >= 0 ? (big >> 6) + 1 : 0;
        ++start[(size_t)bucket[p] + 1];
    }
    for (int b = 0; b < nb; ++b) start[b + 1] += start[b];
    h->rap_need.resize((size_t)nc);
    for (int p = 0; p < nc; ++p) { const int at = start[bucket[p]]++; order[at] = p; h->rap_need[at] = bucket[p] > 0 ? (bucket[p] - 1) * 64 + 63 : -1; }
    if (hipMalloc((void**)&h->d_rap_order, sizeof(int) * (size_t)nc) != hipSuccess) { (void)hipGetLastError(); h->d_rap_order = nullptr; h->rap_need.clear(); return false; }
    if (hipMemcpy(h->d_rap_order, order.data(), sizeof(int) * (size_t)nc, hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); drop_rap_order(h); return false; }
    // level 2 (optional: without it the level's pass simply runs after level 1's): need of row q = the largest need among its children on level 1
    if (h->L >= 3 && h->U[1].n_inner == nc && h->U[1].n_outer > 0) {
        const Compressed& U1 = h->U[1];
        const int nc2 = U1.n_outer;
        std::vector<int> need1((size_t)nc);                     // by level-1 row (natural numbering)
        for (int at = 0; at < nc; ++at) need1[order[at]] = h->rap_need[at];
        std::vector<std::pair<int, int>> rows((size_t)nc2);
        for (int q = 0; q < nc2; ++q) {
            int m = -1;
            for (int e = U1.ptr[q]; e < U1.ptr[q + 1]; ++e) m = std::max(m, need1[U1.idx[e]]);
            rows[q] = {m, q};
        }
        std::sort(rows.begin(), rows.end());
        std::vector<int> order2((size_t)nc2);
        h->rap_need2.resize((size_t)nc2);
        for (int i = 0; i < nc2; ++i) { h->rap_need2[i] = rows[i].first; order2[i] = rows[i].second; }
        if (hipMalloc((void**)&h->d_rap_order2, sizeof(int) * (size_t)nc2) != hipSuccess || hipMemcpy(h->d_rap_order2, order2.data(), sizeof(int) * (size_t)nc2, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            if (h->d_rap_order2) { (void)sync_hipFree(h->d_rap_order2); h->d_rap_order2 = nullptr; }
            h->rap_need2.clear();
        }
    }
    return true;
}

extern "C" {

int gmg_config_size(void) { return (int)sizeof(gmg_config); }

int gmg_config_default(gmg_config* cfg) try {
    if (!cfg) return GMG_ERR_INVALID;
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->device = 0;
    cfg->smoother = GMG_SMOOTHER_MULTICOLOR_GS;
    cfg->jacobi_omega = 0.67;
    cfg->pre_iters = 2;       // gravomg_bindings/src/gravomg/core.py:10
    cfg->post_iters = 2;
    cfg->coarse_mode = GMG_COARSE_AUTO;
    cfg->use_graph = 0;      // measured: the cycle is not launch-bound (eager == graph per cycle) and instantiating costs ~5 ms per system
    cfg->sigma = 0;          // measured: no length sorting inside colour classes beats every window size (irregular meshes; profiles/README.md)
    cfg->row_align = 64;
    cfg->block_rows = 64;
    cfg->block_from_level = 1;
    cfg->block_lanes = 0;
    cfg->device_setup = 1;
    cfg->device_rap = 1;
    cfg->reorder_fine = 2;
    cfg->inner_precision = 0;
    cfg->block_csr = 1;
    cfg->host_threads = 0;
    cfg->verbose = 0;
    cfg->block_ep = 1;
    cfg->dist_shard_levels = 2;
    cfg->fine_col16 = 1;
    cfg->stream_gate = 1;
    cfg->prepare_structure = 1;
    cfg->fuse_restrict_sweep = 1;
    cfg->speculate_head = 1;
    cfg->uniform_slices = 1;
    cfg->color_ahead = 1;
    cfg->dist_exchange = 0;
    cfg->block_fine = 1;      // level 0 blocked too where it pays and is safe (long rows, Stieltjes matrix): see gmg_config
    cfg->restrict_sigma = 64;
    cfg->gs_omega = 1.35;     // measured (profiles/r02/a_iteration_ab.json, f_iteration_ab_omega_scan.json): 7 -> 4 V-cycles to 1e-4 on the 3 M Poisson
                              // problem at the same cost per cycle; centre of the 1.3 - 1.4 plateau on six workloads
    return GMG_OK;
} GMG_CATCH_0

int gmg_host_threads(void) try { return hw_threads(); } GMG_CATCH_0

int gmg_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} GMG_CATCH_0

int gmg_create(const gmg_config* cfg, gmg_handle* out) try {
    if (!out) return GMG_ERR_INVALID;
    gmg_config c;
    if (cfg) c = *cfg; else gmg_config_default(&c);
    if (c.sigma < 0 || c.sigma % 64 || c.restrict_sigma < 0 || c.restrict_sigma % 64 || c.row_align <= 0 || c.row_align % 64 || c.pre_iters < 0 || c.post_iters < 0 ||
        c.reorder_fine < 0 || c.reorder_fine > 2 || c.inner_precision < 0 || c.inner_precision > 1 || c.block_rows < 0 || c.block_rows > gmgk::kBlockRows || c.block_rows % 64 || c.block_from_level < 0 || !(c.gs_omega > 0.0 && c.gs_omega < 2.0) || c.dist_shard_levels < 1 || c.dist_shard_levels > 2 || c.block_fine < 0 || c.block_fine > 1 || c.dist_exchange < 0 || c.dist_exchange > 2 ||
        (c.block_lanes != 0 && c.block_lanes != 1 && c.block_lanes != 4) || (c.block_lanes != 1 && c.block_rows > gmgk::kQuadBlockRows)) return GMG_ERR_INVALID;
    gmg_handle h = new gmg_solver_s();
    h->cfg = c;
    if (h->cfg.host_threads <= 0) h->cfg.host_threads = hw_threads();
    int ndev = gmg_device_count();
    if (ndev > 0 && c.device >= 0 && c.device < ndev && hipSetDevice(c.device) == hipSuccess &&
        hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess) {
        h->has_device = true;
        h->own_stream = h->stream;
        { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device) == hipSuccess && cus > 0) h->n_cus = cus; else (void)hipGetLastError(); }
        (void)hipEventCreate(&h->ev0);
        (void)hipEventCreate(&h->ev1);
        if (hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess) { h->aux_stream = nullptr; (void)hipGetLastError(); }
        if (hipEventCreateWithFlags(&h->aux_ev, hipEventDisableTiming) != hipSuccess) { h->aux_ev = nullptr; (void)hipGetLastError(); }
        if (hipMalloc((void**)&h->d_aux_err, sizeof(int)) != hipSuccess) { h->d_aux_err = nullptr; (void)hipGetLastError(); }
        h->poll = EnvSwitches::get().poll;
        (void)ensure_bounce(h);         // 2 x 16 MB pinned, once per handle (page-locking is not free: not inside gmg_set_system)
    }
    *out = h;
    return GMG_OK;     // host-only entry points work without a device; device ones report GMG_ERR_NO_DEVICE
} GMG_CATCH_0

void gmg_destroy(gmg_handle h) {
    if (!h) return;
    PoolScope pool_scope_(&h->pool);
    if (h->has_device) {
        (void)hipSetDevice(h->cfg.device);
        (void)hipStreamSynchronize(h->stream);
        p2p_release_handle(h);
        drop_system(h);
        drop_device_transfers(h);
        for (double** p : {&h->d_stage, &h->d_partials, &h->d_norm, &h->d_watch}) if (*p) (void)dev_free(*p);
        if (h->h_pinned) (void)sync_hipHostFree(h->h_pinned);
        for (int i = 0; i < 2; ++i) { if (h->h_stage[i]) (void)sync_hipHostFree(h->h_stage[i]); if (h->h_stage_ev[i]) (void)hipEventDestroy(h->h_stage_ev[i]); }
        for (hipEvent_t& e : h->h_chunk_ev) if (e) (void)hipEventDestroy(e);
        if (h->h_norm) (void)sync_hipHostFree(h->h_norm);
        if (h->h_flag) (void)sync_hipHostFree(h->h_flag);
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
        if (h->aux_ev) (void)hipEventDestroy(h->aux_ev);
        if (h->aux_stream) { (void)hipStreamSynchronize(h->aux_stream); (void)sync_hipStreamDestroy(h->aux_stream); }
        if (h->d_aux_err) (void)sync_hipFree(h->d_aux_err);
        drop_rap_order(h);
        for (int i = 0; i < 2; ++i) { if (h->bounce[i]) (void)sync_hipHostFree(h->bounce[i]); if (h->bounce_ev[i]) (void)hipEventDestroy(h->bounce_ev[i]); }
        for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
        (void)sync_hipStreamDestroy(h->own_stream);
        h->pool.trim();
    }
    delete h;
}

const char* gmg_last_error(gmg_handle h) { return h ? h->err.c_str() : "null handle"; }

int gmg_set_num_levels(gmg_handle h, int L) try {
    if (!h || L < 0 || L > 64) return h ? fail(h, GMG_ERR_INVALID, "invalid level count") : GMG_ERR_INVALID;
    PoolScope pool_scope_(&h->pool);
    if (h->has_device) { drop_system(h); drop_device_transfers(h); }
    h->patches.clear(); h->patches_ready = false;
    h->bfs_order.clear();
    h->fine_graph.reset();
    drop_rap_order(h);
    h->L = L;
    h->ord_cache_valid = false;
    h->U.assign(L, Compressed());
    h->U_set.assign(L, 0);
    return GMG_OK;
} GMG_CATCH_H

int gmg_set_prolongation(gmg_handle h, int k, int n_fine, int n_coarse, const int* colptr, const int* rowidx, const double* val) try {
    if (!h) return GMG_ERR_INVALID;
    PoolScope pool_scope_(&h->pool);
    if (h->L < 0) return fail(h, GMG_ERR_STATE, "call gmg_set_num_levels first");
    if (k < 0 || k >= h->L || n_fine <= 0 || n_coarse <= 0 || !colptr || !rowidx || !val) return fail(h, GMG_ERR_INVALID, "bad prolongation arguments");
    for (int j = 0; j < n_coarse; ++j) if (colptr[j + 1] < colptr[j]) return fail(h, GMG_ERR_INVALID, "colptr not monotone");
    {
        std::atomic<bool> bad{false};
        parallel_ranges(colptr[n_coarse], h->cfg.host_threads, [&](int lo, int hi, int) {
            for (int p = lo; p < hi; ++p) if (rowidx[p] < 0 || rowidx[p] >= n_fine) { bad = true; return; }
        }, 1 << 18);
        if (bad) return fail(h, GMG_ERR_INVALID, "row index out of range in U");
    }
    if (h->has_device) { drop_system(h); drop_device_transfers(h); }
    h->patches.clear(); h->patches_ready = false;
    {   // (default-initialised vectors filled on all cores: 108 MB of first touches at 3 M vertices)
        Compressed& u = h->U[k];
        u.n_outer = n_coarse; u.n_inner = n_fine;
        u.ptr.assign(colptr, colptr + n_coarse + 1);
        u.idx.resize((size_t)colptr[n_coarse]); u.val.resize((size_t)colptr[n_coarse]);
        threaded_copy_bytes(u.idx.data(), rowidx, sizeof(int) * u.idx.size(), h->cfg.host_threads);
        threaded_copy_bytes(u.val.data(), val, sizeof(double) * u.val.size(), h->cfg.host_threads);
    }
    h->U_set[k] = 1;
    h->ord_cache_valid = false;
    if (k == 0) drop_rap_order(h);
    return GMG_OK;
} GMG_CATCH_H

// h->mass (natural numbering) -> device numbering of level 0 (d_mass, d_minv); needs a system (the ordering)
static int upload_mass(gmg_handle h) {
    const int n = (int)h->mass.size();
    {
        // device numbering (padding rows get weight 1: they carry r = b = 0)
        Level& l = h->lv[0];
        if (l.n != n) return fail(h, GMG_ERR_INVALID, "mass size does not match the system");
        int rc = ensure_stage(h, (size_t)n);
        if (rc) return rc;
        for (double** p : {&h->d_mass, &h->d_minv}) if (!*p) HIPCHK(dev_malloc((void**)p, sizeof(double) * l.n_pad));
        if ((rc = h2d(h, h->d_stage, h->mass.data(), sizeof(double) * (size_t)n))) return rc;
        hipLaunchKernelGGL(gmgk::permute_mass, dim3((l.n_pad + 255) / 256), dim3(256), 0, h->stream, h->d_stage, l.d_new2old, l.n_pad, h->d_mass, h->d_minv);
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return GMG_OK;
}

int gmg_set_mass(gmg_handle h, int n, const double* mass_diag) try {
    if (!h || n <= 0 || !mass_diag) return h ? fail(h, GMG_ERR_INVALID, "bad mass arguments") : GMG_ERR_INVALID;
    PoolScope pool_scope_(&h->pool);
    // a live system (or the prepared structure of one) fixes n: a mass of another size could only leave the M-weighted norms on stale weights
    if ((h->system_ready || h->placeholder_ready) && !h->lv.empty() && h->lv[0].n != n)
        return fail(h, GMG_ERR_INVALID, "mass size does not match the system");
    h->mass.assign(mass_diag, mass_diag + n);
    // (with a system -- or the prepared structure of one: the ordering that permutes the mass exists -- it goes to the device now)
    if (h->has_device && (h->system_ready || h->placeholder_ready) && !h->lv.empty() && h->lv[0].n == n && h->lv[0].d_new2old) { h->mass_dirty = false; return upload_mass(h); }
    h->mass_dirty = true;
    return GMG_OK;
} GMG_CATCH_H

// fp32 twins of the value arrays (mixed precision); `alloc`: (re)allocate them, otherwise they exist with the right sizes
static int refresh_fp32_twins(gmg_handle h, bool alloc) {
    const int L = h->L;
    auto twin = [&](DevSell& m) -> int {
        if (!m.val || m.stored <= 0) return GMG_OK;
        if (alloc || !m.val32) {
            if (m.val32) { (void)dev_free(m.val32); m.val32 = nullptr; }
            HIPCHK(dev_malloc((void**)&m.val32, sizeof(float) * (size_t)m.stored));
        }
        launch_cvt(h, m.val, m.val32, (size_t)m.stored);
        return GMG_OK;
    };
    for (int k = 0; k < L; ++k) {
        Level& l = h->lv[k];
        int rc;
        if ((rc = twin(l.Aoff)) || (rc = twin(l.Ain)) || (rc = twin(l.Aout)) || (rc = twin(l.P)) || (rc = twin(l.R))) return rc;
        if (l.use_bcsr) {
            if (alloc || !l.bc_val32) {
                if (l.bc_val32) { (void)dev_free(l.bc_val32); l.bc_val32 = nullptr; }
                HIPCHK(dev_malloc((void**)&l.bc_val32, sizeof(float) * (size_t)std::max<int64_t>(l.bc_nnz, 1)));
            }
            launch_cvt(h, l.bc_val, l.bc_val32, (size_t)l.bc_nnz);
        }
        if (l.use_ep) {
            if (alloc || !l.ep_val32 || !l.ee_val32) {
                for (float** q : {&l.ep_val32, &l.ee_val32}) { if (*q) (void)dev_free(*q); *q = nullptr; }
                HIPCHK(dev_malloc((void**)&l.ep_val32, sizeof(float) * (size_t)std::max<int64_t>(l.ep_nnz, 1)));
                HIPCHK(dev_malloc((void**)&l.ee_val32, sizeof(float) * (size_t)std::max<int64_t>(l.ee_nnz, 1)));
            }
            launch_cvt(h, l.ep_val, l.ep_val32, (size_t)l.ep_nnz);
            launch_cvt(h, l.ee_val, l.ee_val32, (size_t)l.ee_nnz);
        }
        if (alloc || !l.diag32) {
            if (l.diag32) { (void)dev_free(l.diag32); l.diag32 = nullptr; }
            HIPCHK(dev_malloc((void**)&l.diag32, sizeof(float) * (size_t)l.n_pad));
        }
        launch_cvt(h, l.diag, l.diag32, (size_t)l.n_pad);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMG_OK;
}

// Where the coarsest solve of this system runs (gmg_config::coarse_mode): GMG_COARSE_AUTO puts it on the device while the dense inverse is small
// enough to be read once per cycle for less than the host round trip costs (n_L <= kCoarseDeviceMax: 512 MB, ~0.1 ms; the reference's
// lower_bound = 1000 / ratio = 8 keep n_L below 8 000).
constexpr int kCoarseDeviceMax = 8192;
static bool want_coarse_device(gmg_handle h, int n_coarse) {
    if (h->cfg.coarse_mode == GMG_COARSE_DEVICE_INVERSE) return true;
    return h->cfg.coarse_mode == GMG_COARSE_AUTO && n_coarse <= kCoarseDeviceMax;
}

// Dense inverse of the coarsest operator, built ON THE DEVICE from the host's sparse factor (setup_kernels.hip.hpp::coarse_inverse_tiles): the factor
// goes up in the device's chunk layout (a few MB), one launch carries every 64-column tile of the identity through it, a second one mirrors the
// lower triangle, a third one moves it from the factor's numbering into the level's.  (Until round 6 the host solved n_L right-hand sides one by
// one: 188 - 226 ms of set-up at n_L = 6 005.)
static int build_coarse_inverse_device(gmg_handle h) {
    auto t0 = clk::now();
    const int nl = h->coarse.n;
    SupernodalLDLT::DeviceFactor E;
    h->coarse.export_device_factor(E);
    h->timing["coarse_inverse_export_ms"] = ms_since(t0);
    // tile width: enough workgroups for the chip's compute units (121 tiles of 16 columns at n_L = 1 929, 376 at 6 005: two 1024-thread
    // workgroups per compute unit are resident), wider tiles only where there would be more tiles than that
    const int width = nl <= 8192 ? 16 : (nl <= 16384 ? 32 : 64);
    std::vector<int> tile_ptr_h, tile_q_h;
    E.tile_paths(width, tile_ptr_h, tile_q_h);
    const std::vector<int> lev_big_h = E.big_per_level(gmgs::kInvBigRows);
    // chunk records in the kernel's two orders (setup_kernels.hip.hpp::InvFactor)
    auto record = [&](int q) { return make_int4(E.q_col0[(size_t)q] | (E.q_w[(size_t)q] << 24), E.q_rptr[(size_t)q], E.q_rptr[(size_t)q + 1], q); };
    std::vector<int4> lev_meta_h(E.lev_q.size()), tile_meta_h(tile_q_h.size());
    for (size_t k = 0; k < E.lev_q.size(); ++k) lev_meta_h[k] = record(E.lev_q[k]);
    for (size_t t = 0; t < tile_q_h.size(); ++t) tile_meta_h[t] = record(tile_q_h[t]);
    DevTmp<int4> lev_meta, tile_meta;
    DevTmp<int> rows, lev_ptr, lev_big, tile_ptr;
    DevTmp<double> vals, tri, dinv;
    int rc;
    auto up_i = [&](DevTmp<int>& d, const std::vector<int>& v) -> int { int r = d.alloc(h, std::max<size_t>(v.size(), 1)); if (r) return r; return v.empty() ? GMG_OK : h2d(h, d.p, v.data(), sizeof(int) * v.size()); };
    auto up_4 = [&](DevTmp<int4>& d, const std::vector<int4>& v) -> int { int r = d.alloc(h, std::max<size_t>(v.size(), 1)); if (r) return r; return v.empty() ? GMG_OK : h2d(h, d.p, v.data(), sizeof(int4) * v.size()); };
    auto up_d = [&](DevTmp<double>& d, const std::vector<double>& v) -> int { int r = d.alloc(h, std::max<size_t>(v.size(), 1)); if (r) return r; return v.empty() ? GMG_OK : h2d(h, d.p, v.data(), sizeof(double) * v.size()); };
    if ((rc = up_4(lev_meta, lev_meta_h)) || (rc = up_4(tile_meta, tile_meta_h)) || (rc = up_i(rows, E.rows)) || (rc = up_i(lev_ptr, E.lev_ptr)) || (rc = up_i(lev_big, lev_big_h)) ||
        (rc = up_i(tile_ptr, tile_ptr_h)) || (rc = up_d(vals, E.vals)) || (rc = up_d(tri, E.tri)) || (rc = up_d(dinv, E.dinv)))
        return rc;
    std::vector<int> inv_h((size_t)nl);
    for (int i = 0; i < nl; ++i) inv_h[(size_t)E.perm[(size_t)i]] = i;
    DevTmp<int> perm_d, inv_d;
    if ((rc = up_i(perm_d, E.perm)) || (rc = up_i(inv_d, inv_h))) return rc;
    const size_t bytes = sizeof(double) * (size_t)nl * nl;
    DevTmp<double> X;                                       // the inverse in the factor's numbering
    if ((rc = X.alloc(h, std::max<size_t>((size_t)nl * nl, 1)))) return rc;
    const int lda = (nl + 7) / 8 * 8;
    if (h->d_ainv && h->ainv_n != nl) { (void)dev_free(h->d_ainv); h->d_ainv = nullptr; }
    if (!h->d_ainv) HIPCHK(dev_malloc((void**)&h->d_ainv, std::max<size_t>(sizeof(double) * (size_t)nl * lda, 8)));
    h->ainv_n = nl; h->ainv_ld = lda;
    HIPCHK(hipMemsetAsync(X.p, 0, bytes, h->stream));
    gmgs::InvFactor F;
    F.n = nl; F.nq = E.nq; F.nlev = E.nlev;
    F.lev_meta = lev_meta.p; F.tile_meta = tile_meta.p; F.rows = rows.p; F.lev_ptr = lev_ptr.p; F.lev_big = lev_big.p; F.tile_ptr = tile_ptr.p;
    F.vals = vals.p; F.tri = tri.p; F.dinv = dinv.p;
    static_assert(gmgs::kInvChunk == SupernodalLDLT::kChunk, "chunk width of the exported factor");
    const int nt = (nl + width - 1) / width, nm = (nl + 63) / 64;
    h->timing["coarse_inverse_upload_ms"] = ms_since(t0) - h->timing["coarse_inverse_export_ms"];
    if (nl > 0) {
        (void)hipEventRecord(h->ev0, h->stream);
        const dim3 block(64 * gmgs::kInvWaves);
        if (width == 16) hipLaunchKernelGGL(gmgs::coarse_inverse_tiles<16>, dim3(nt), block, 0, h->stream, F, X.p);
        else if (width == 32) hipLaunchKernelGGL(gmgs::coarse_inverse_tiles<32>, dim3(nt), block, 0, h->stream, F, X.p);
        else hipLaunchKernelGGL(gmgs::coarse_inverse_tiles<64>, dim3(nt), block, 0, h->stream, F, X.p);
        (void)hipEventRecord(h->ev1, h->stream);
        hipLaunchKernelGGL(gmgs::mirror_lower_to_upper, dim3(nm, nm), dim3(256), 0, h->stream, X.p, nl);
        // ... and into the level's numbering: the product kernel then reads its vectors contiguously
        if (nl <= 8192) hipLaunchKernelGGL(gmgs::permute_symmetric, dim3(nl), dim3(256), sizeof(double) * (size_t)nl, h->stream, (const double*)X.p, (const int*)perm_d.p, (const int*)inv_d.p, nl, h->d_ainv, lda);
        else hipLaunchKernelGGL(gmgs::permute_symmetric_scatter, dim3(nl), dim3(256), 0, h->stream, (const double*)X.p, (const int*)perm_d.p, nl, h->d_ainv, lda);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));        // (the temporaries above go back to the pool; the host copy E dies here)
    if (nl > 0) { float ms = 0.f; if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->timing["coarse_inverse_tiles_ms"] = ms; }
    h->timing["coarse_inverse_ms"] = ms_since(t0);
    h->timing["coarse_inverse_levels"] = E.nlev;
    h->timing["coarse_inverse_chunks"] = E.nq;
    return GMG_OK;
}

// gmg_set_system for a matrix with the sparsity pattern of the live system: values only.  Returns 1 when it cannot be
// done in place (nothing has been changed then, except values that the full path overwrites anyway).
// values_uploaded: the caller has already put `val` into the resident A_0 (the speculative upload of set_system_impl)
// l1_rows_done: ... and has queued the numeric Galerkin pass of the first l1_rows_done rows of level 1 behind it (flag: h->d_aux_err)
static int refresh_system_values(gmg_handle h, int n, const double* val, clk::time_point t_all, bool values_uploaded = false, int l1_rows_done = 0, int l2_rows_done = 0) {
    const int L = h->L;
    auto mark = [&](const std::string& what) { h->timing["t_" + what] = ms_since(t_all); };
    for (int k = 0; k <= L; ++k) if (!h->lv[k].dA.ptr || !h->lv[k].dA.idx || !h->lv[k].dA.val) return 1;
    for (int k = 0; k < L; ++k) if (!h->lv[k].d_old2new || !h->lv[k].diag) return 1;
    if (!h->lv[L].hostA_pattern) return 1;
    for (auto it = h->timing.begin(); it != h->timing.end();) it = it->first.rfind("t_", 0) == 0 ? h->timing.erase(it) : std::next(it);
    mark("pattern_key");
    int rc;
    DevTmp<int> d_err;
    if ((rc = d_err.alloc(h, 1))) return rc;
    HIPCHK(hipMemsetAsync(d_err.p, 0, sizeof(int), h->stream));
    h->loaded_d = 0;
    for (int k = 0; k <= L; ++k) h->lv[k].hostA_values = false;          // host copies (if any) keep their pattern only
    if (!values_uploaded && (rc = h2d(h, h->lv[0].dA.val, val, sizeof(double) * (size_t)h->lv[0].nnz))) return rc;
    mark("upload_A0");
    auto t0 = clk::now();
    for (int k = 1; k <= L; ++k) {
        Level& lk = h->lv[k];
        if ((rc = device_rap(h, h->lv[k - 1].dA, h->dU[k - 1], h->dE3[k - 1], lk.dA, lk.A, false, k == L, &lk.nnz, d_err.p, true, k == 1 ? l1_rows_done : (k == 2 ? l2_rows_done : 0)))) return rc < 0 ? rc : GMG_ERR_STATE;
        if (k == L) lk.hostA_values = true;
        mark("rap_l" + std::to_string(k));
    }
    h->timing["reduction"] = ms_since(t0);
    double ms_factor = 0;
    std::future<bool> factor_done = std::async(std::launch::async, [&] {
        auto t = clk::now();
        bool ok = h->coarse.factor(h->lv[L].A, true);
        h->coarse_warm = false;
        ms_factor = ms_since(t);
        return ok;
    });
    auto tl = clk::now();
    for (int k = 0; k < L && rc == GMG_OK; ++k) rc = device_refill_level(h, k, d_err.p);
    int herr = 0, herr_aux = 0;
    if (rc == GMG_OK) {
        (void)hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, h->stream);
        if (l1_rows_done > 0) (void)hipMemcpyAsync(&herr_aux, h->d_aux_err, sizeof(int), hipMemcpyDeviceToHost, h->stream);
        (void)hipStreamSynchronize(h->stream);
        if (herr == 0) herr = herr_aux;
    }
    h->timing["setup_rap_rows_pipelined"] = l1_rows_done;
    h->timing["setup_rap_rows_pipelined_l2"] = l2_rows_done;
    h->timing["setup_device_layout"] = ms_since(tl);
    mark("device_layout");
    const bool factor_ok = factor_done.get();
    mark("factor_joined");
    if (rc != GMG_OK) return rc;
    if (herr == 2) return fail(h, GMG_ERR_NUMERIC, "system matrix has a missing or zero diagonal entry");
    if (herr != 0) { h->refill_ready = false; return fail(h, GMG_ERR_STATE, "value refresh failed on the device"); }
    if (!factor_ok) return fail(h, GMG_ERR_NUMERIC, "coarsest operator is singular (LDL^T hit a zero pivot)");
    h->coarse_device = want_coarse_device(h, h->lv[L].A.n_outer);
    h->timing["coarse_on_device"] = h->coarse_device ? 1.0 : 0.0;
    if (h->coarse_device && (rc = build_coarse_inverse_device(h))) return rc;
    if (h->cfg.inner_precision && (rc = refresh_fp32_twins(h, false))) return rc;
    if (!h->mass.empty() && (h->mass_dirty || !h->d_mass)) {          // a mass set while only the placeholder structure stood (prepare_structure)
        if ((int)h->mass.size() != n) return fail(h, GMG_ERR_INVALID, "mass size does not match the system");
        if ((rc = upload_mass(h))) return rc;
        h->mass_dirty = false;
    }
    h->timing["coarsest_solve"] = ms_factor;
    h->timing["setup_ordering_cached"] = 1.0;
    h->timing["setup_values_only"] = 1.0;
    h->timing["setup_ordering"] = 0.0; h->timing["setup_sell"] = 0.0; h->timing["setup_wait_ordering"] = 0.0;
    for (int k = 0; k <= L; ++k) h->timing["setup_ordering_l" + std::to_string(k)] = 0.0;
    mark("mass_done");
    h->timing["upload"] = ms_since(t_all) - h->timing["reduction"];
    h->timing["setup_total"] = ms_since(t_all);
    h->timing["coarse_host_ms"] = 0.0;
    return GMG_OK;
}

static int set_system_impl(gmg_handle h, int n, const int* colptr, const int* rowidx, const double* val) {
    NEED_DEVICE();
    if (h->L <= 0) return fail(h, GMG_ERR_STATE, "hierarchy has no transfer levels (U is empty)");
    for (int k = 0; k < h->L; ++k) if (!h->U_set[k]) return fail(h, GMG_ERR_STATE, "prolongation matrix missing for level " + std::to_string(k));
    if (n <= 0 || !colptr || !rowidx || !val) return fail(h, GMG_ERR_INVALID, "bad system arguments");
    if (h->U[0].n_inner != n) return fail(h, GMG_ERR_INVALID, "system size does not match U[0]");
    for (int k = 0; k + 1 < h->L; ++k)
        if (h->U[k].n_outer != h->U[k + 1].n_inner) return fail(h, GMG_ERR_INVALID, "U[k] / U[k+1] shapes do not chain");
    auto t_all = clk::now();
    HIPCHK(hipSetDevice(h->cfg.device));
    const int L = h->L;
    Compressed canon;       // only filled when the caller's storage is unsorted or has duplicates
    // A system of the live system's size is most likely the live pattern with new values (the demos' new tau per frame; the first system after
    // the structure was prepared): its values go up to the resident A_0 AHEAD of the verdict, while the worker threads inspect and digest the
    // pattern (3-4 ms at 3 M vertices, as long as the upload itself).  Should the pattern be another one after all, nothing is lost but the
    // live system, which the full set-up replaces anyway.
    bool speculative_upload = false;
    int l1_rows_done = 0;             // coarse rows of level 1 whose numeric Galerkin pass was queued behind the chunks of that upload
    int l2_rows_done = 0;             // ... and of level 2 behind those
    uint64_t pat_key[2] = {0, 0};
    bool have_key = false;
    int rc_spec = 1;                  // result of the refresh that ran ahead of the verdict (spec_done)
    bool spec_done = false;
    // A handle WITHOUT a live system has nothing to refresh: the cold set-up's first device step -- A_0 in natural numbering, 250 MB at 3 M
    // vertices -- goes up beside the inspection instead of after it (into a matrix of its own: the levels are rebuilt further down)
    // Colouring ahead of the layout decisions: when this call cannot be a values-only refresh (no live structure, or another entry count) and level 0
    // will most likely be the colour-major one in the caller's order, the greedy colouring -- 10-13 ms on one core at 3 M vertices, what a cold
    // set-up waits for longest -- starts as soon as the inspection of the caller's arrays has its verdict, while this thread is still busy sending
    // A_0 to the device (host_plan.hpp::greedy_coloring_ahead: range-checked all the same).
    // Used by the level-0 ordering task if the decisions below come out that way (canonical arrays, no renumbering, not blocked, no
    // cached ordering); stopped and joined otherwise, and in any case before this function returns (it reads the caller's arrays).
    struct AheadColoring {
        std::atomic<int> stop{0};
        PreColoring pre;
        std::future<int> fut;
        const int* ptr = nullptr;
        void cancel() { stop.store(1); }
        ~AheadColoring() { stop.store(1); if (fut.valid()) fut.wait(); }
    } ahead;
    bool want_ahead = false;
    {
        const bool live0 = h->system_ready || h->placeholder_ready;
        const bool refresh_possible = live0 && h->live_key_valid && (int)h->lv.size() == L + 1 && h->lv[0].n == n && (int64_t)colptr[n] == h->lv[0].nnz;
        if (!refresh_possible && !h->ord_cache_valid && h->cfg.smoother == GMG_SMOOTHER_MULTICOLOR_GS && n >= (1 << 17) && h->cfg.color_ahead &&
            !(h->cfg.block_rows > 0 && h->cfg.block_from_level <= 0)) {
            ahead.ptr = colptr;
            const int64_t nnz_claimed = colptr[n];
            want_ahead = true;
            (void)nnz_claimed;
        }
    }
    DevCsr early_A0;
    bool early_upload = false;
    struct EarlyGuard { DevCsr& d; ~EarlyGuard() { free_csr(d); } } early_guard{early_A0};
    {
        std::shared_future<int> inspected = std::async(std::launch::async, [&] { return inspect_pattern(n, n, colptr, rowidx, h->cfg.host_threads); }).share();
        if (want_ahead) {
            // (behind the inspection's verdict, which takes a millisecond or two -- not behind the upload this thread makes meanwhile: started at
            // entry, beside the inspection's threads, the loop ran at half its speed)
            const int64_t nnz_claimed = colptr[n];
            ahead.fut = std::async(std::launch::async, [&ahead, inspected, n, colptr, rowidx, nnz_claimed] {
                if (inspected.get() != 0) return -2;
                return greedy_coloring_ahead(n, colptr, rowidx, nnz_claimed, ahead.pre.c8, ahead.stop);
            });
        }
        std::future<void> keyed;
        if ((h->system_ready || h->placeholder_ready) && h->live_key_valid && h->refill_ready && (int)h->lv.size() == L + 1 && h->lv[0].n == n && colptr[0] == 0 &&
            (int64_t)colptr[n] == h->lv[0].nnz && h->lv[0].dA.val) {
            keyed = std::async(std::launch::async, [&] { pattern_key(n, colptr, rowidx, h->cfg.host_threads, pat_key); });
            h->loaded_d = 0;
            // ... and the numeric Galerkin pass of level 1 -- the longest kernel of the refresh -- follows the values chunk by chunk on a second
            // stream, over the coarse rows in the order in which their inputs arrive (ensure_rap_order).  A randomly numbered input needs the
            // last chunk for nearly every row: everything then runs after the upload, as before.
            const bool pipeline = L >= 2 && h->aux_stream && h->aux_ev && h->d_aux_err && h->dU_ready && (int)h->dU.size() == L && h->dU[0].ptr && h->dE3[0].cnt &&
                                  h->lv[1].dA.ptr && h->lv[1].dA.idx && h->lv[1].dA.val && h->lv[1].dA.n_outer == h->U[0].n_outer && !h->dU_flagged && ensure_rap_order(h);
            const bool pipeline2 = pipeline && L >= 3 && !h->rap_need2.empty() && h->d_rap_order2 && h->dU[1].ptr && h->dE3[1].cnt && h->lv[2].dA.ptr && h->lv[2].dA.idx && h->lv[2].dA.val &&
                                   h->lv[2].dA.n_outer == h->U[1].n_outer && (int)h->rap_need2.size() == h->U[1].n_outer;
            const size_t val_bytes = sizeof(double) * (size_t)h->lv[0].nnz;
            std::function<void(size_t, hipEvent_t)> after_chunk = [&](size_t bytes_done, hipEvent_t arrived) {
                const int nc = h->U[0].n_outer;
                const int64_t entries = (int64_t)(bytes_done / sizeof(double));
                // complete rows of A_0 (a colptr that is not ascending -- the inspection is still running -- gives some row count: the rows computed from
                // it are recomputed by the full set-up that follows a failed inspection)
                const int rows = (int)(std::upper_bound(colptr, colptr + n + 1, (int)std::min<int64_t>(entries, colptr[n])) - colptr) - 1;
                const bool last = bytes_done >= val_bytes;
                int p_hi = last ? nc : (int)(std::upper_bound(h->rap_need.begin(), h->rap_need.end(), rows - 1) - h->rap_need.begin());
                if (p_hi - l1_rows_done < 32768 && !last) return;
                if (p_hi > l1_rows_done) {
                    (void)hipStreamWaitEvent(h->aux_stream, arrived, 0);
                    const DevCsr &dA0 = h->lv[0].dA, &dU0 = h->dU[0], &dC = h->lv[1].dA;
                    const DevEll3& e3 = h->dE3[0];
                    hipLaunchKernelGGL(gmgs::rap_rows<2>, dim3(p_hi - l1_rows_done), dim3(64), 0, h->aux_stream, dA0.ptr, dA0.idx, dA0.val, dU0.ptr, dU0.idx, dU0.val, e3.cnt, e3.col, e3.val, p_hi,
                                       (const int*)dC.ptr, (int*)nullptr, dC.idx, dC.val, h->d_aux_err, l1_rows_done, (const int*)h->d_rap_order);
                    l1_rows_done = p_hi;
                }
                // level 2 behind it, on the same stream: the rows whose children on level 1 are all among the rows launched so far (the pieces
                // of level 1 end at a boundary of `need`: every row that needs no more than fine row rows - 1 has been launched)
                if (pipeline2) {
                    const int nc2 = h->U[1].n_outer;
                    const int q_hi = last ? nc2 : (int)(std::upper_bound(h->rap_need2.begin(), h->rap_need2.end(), rows - 1) - h->rap_need2.begin());
                    if (q_hi > l2_rows_done && (last || q_hi - l2_rows_done >= 4096)) {
                        const DevCsr &dA1 = h->lv[1].dA, &dU1 = h->dU[1], &dC2 = h->lv[2].dA;
                        const DevEll3& e31 = h->dE3[1];
                        hipLaunchKernelGGL(gmgs::rap_rows<2>, dim3(q_hi - l2_rows_done), dim3(64), 0, h->aux_stream, dA1.ptr, dA1.idx, dA1.val, dU1.ptr, dU1.idx, dU1.val, e31.cnt, e31.col, e31.val, q_hi,
                                           (const int*)dC2.ptr, (int*)nullptr, dC2.idx, dC2.val, h->d_aux_err, l2_rows_done, (const int*)h->d_rap_order2);
                        l2_rows_done = q_hi;
                    }
                }
            };
            if (pipeline) (void)hipMemsetAsync(h->d_aux_err, 0, sizeof(int), h->aux_stream);
            const int rc_up = h2d(h, h->lv[0].dA.val, val, val_bytes, pipeline ? &after_chunk : nullptr);
            if (pipeline) { (void)hipEventRecord(h->aux_ev, h->aux_stream); (void)hipStreamWaitEvent(h->stream, h->aux_ev, 0); }
            speculative_upload = true;
            // ... and so does the rest of the refresh (Galerkin chain, layout refills, numeric LDL^T): none of it reads the pattern the threads are
            // still inspecting, all of it is overwritten by the full set-up should the verdict be "another pattern"
            if (rc_up == GMG_OK) { rc_spec = refresh_system_values(h, n, val, t_all, true, l1_rows_done, l2_rows_done); spec_done = true; }
            keyed.get();
            have_key = true;
            if (rc_up != GMG_OK) { (void)inspected.get(); h->system_ready = false; h->placeholder_ready = false; return rc_up; }
        }
        if (!speculative_upload && !(h->system_ready || h->placeholder_ready) && h->cfg.device_setup && h->cfg.device_rap && h->cfg.smoother == GMG_SMOOTHER_MULTICOLOR_GS &&
            colptr[0] == 0 && colptr[n] >= n) {
            const int rc_up = upload_csr_raw(h, early_A0, n, colptr, rowidx, val);
            if (rc_up != GMG_OK) { (void)inspected.get(); return rc_up; }
            early_upload = true;
        }
        const int what = inspected.get();
        if (what == 2) { if (speculative_upload) { h->system_ready = false; h->placeholder_ready = false; } return fail(h, GMG_ERR_INVALID, "index out of range in LHS"); }
        if (what == 1) {
            canon = canonical_copy(n, n, colptr, rowidx, val, h->cfg.host_threads);
            colptr = canon.ptr.data(); rowidx = canon.idx.data(); val = canon.val.data();
            have_key = false;
            ahead.cancel();                         // (coloured from rows that are not ascending: no result)
            if (early_upload) { free_csr(early_A0); early_upload = false; }      // (it went up in the caller's storage order)
            // (the values went up in the caller's storage order, the resident pattern is canonical: the live system is void, and this
            // matrix takes the full set-up from its canonical copy)
            if (speculative_upload) { speculative_upload = false; spec_done = false; h->system_ready = false; h->placeholder_ready = false; h->refill_ready = false; }
        }
    }
    // (live: a system is set -- or the structure of one was prepared on placeholder values when the hierarchy was finalized, prepare_structure)
    const bool live = h->system_ready || h->placeholder_ready;
    if (live && h->live_key_valid && h->refill_ready && (int)h->lv.size() == L + 1 && h->lv[0].n == n) {
        // Same sparsity pattern as the live system (and the same hierarchy: refill_ready dies with it)?  Then every
        // structure on the device stands and only values move: LHS values up, numeric Galerkin passes, value refill of
        // the layouts, numeric LDL^T.  (The demos' usage: lhs = M + tau * S with a new tau per frame.)
        if (!have_key) { pattern_key(n, colptr, rowidx, h->cfg.host_threads, pat_key); have_key = true; }
        // (a level 0 that gmg_config::block_fine blocked stays blocked only while the new values pass its sign test)
        const bool keeps_fine_blocks = !(h->lv[0].ord.blocked && h->cfg.block_from_level >= 1) || stieltjes_signs(n, colptr, rowidx, val, h->cfg.host_threads);
        if (pat_key[0] == h->live_key[0] && pat_key[1] == h->live_key[1] && colptr[n] == h->lv[0].nnz && keeps_fine_blocks) {
            const bool from_placeholder = h->placeholder_ready && !h->system_ready;
            int rc = spec_done ? rc_spec : refresh_system_values(h, n, val, t_all);
            if (rc != GMG_OK && rc != 1) { h->system_ready = false; h->placeholder_ready = false; h->refill_ready = false; }      // half-refreshed values: no solves on them
            if (rc == GMG_OK) {
                h->system_ready = true; h->placeholder_ready = false; h->timing["setup_structure_prepared"] = from_placeholder ? 1.0 : 0.0;
                h->timing["t_verdict"] = h->timing["setup_total"] = ms_since(t_all);
                h->timing["upload"] = h->timing["setup_total"] - h->timing["reduction"];
            }
            if (rc != 1) return rc;                 // 1: could not be done in place -> the full path below rebuilds everything
        }
    }
    // (the resident values were overwritten ahead of the verdict and the pattern turned out to be another one: the live system is gone -- the
    // full path below drops it anyway; a failure on the way must not leave a system that solves with foreign values)
    if (speculative_upload) {
        h->system_ready = false; h->placeholder_ready = false;
        if (spec_done && rc_spec != GMG_OK) h->err.clear();      // (the refresh ran on a matrix of another pattern: its complaint is about that combination, not about this call)
    }
    if (live && h->live_key_valid && (int)h->lv.size() == L + 1) {
        h->ord_cache.resize(L + 1);
        for (int k = 0; k <= L; ++k) h->ord_cache[k] = std::move(h->lv[k].ord);
        h->ord_cache_key[0] = h->live_key[0]; h->ord_cache_key[1] = h->live_key[1];
        h->ord_cache_valid = true;
    }
    h->live_key_valid = false;
    drop_system(h);
    h->lv.resize(L + 1);
    // Host setup as a small task graph (everything below the RAP chain is independent per level):
    //   main thread : A_1 .. A_L by Galerkin products (multigrid_solver.cpp:1387-1392)
    //   per level k : device ordering of level k as soon as A_k exists, then its operator layout (SELL)
    //   level L     : LDL^T factorisation of A_L (:1401) (+ dense inverse for GMG_COARSE_DEVICE_INVERSE)
    //   per level k : transfer layouts P_k, R_k once the orderings of levels k and k+1 exist
    // The uploads follow on the calling thread once their inputs are ready.
    auto mark = [&](const std::string& what) { h->timing["t_" + what] = ms_since(t_all); };   // setup timeline (ms since entry)
    h->timing["setup_wait_ordering"] = 0.0; h->timing["setup_device_layout"] = 0.0;
    const bool mc = h->cfg.smoother == GMG_SMOOTHER_MULTICOLOR_GS;
    bool device_setup = h->cfg.device_setup != 0;
    const bool part = h->part_world > 1;      // gmg_dist_partition: lay out and keep this rank's rows of levels 0 / 1 only
    if (part && !(device_setup && h->cfg.device_rap && mc)) return fail(h, GMG_ERR_UNSUPPORTED, "a partitioned set-up needs device_setup = 1, device_rap = 1 and the multicolour smoother");
    h->partitioned = false;
    h->pool.reset_peak();
    if (device_setup) {
        int rc = ensure_device_transfers(h);        // no-op when gmg_use_hierarchy (or an earlier system) made them
        if (rc) return rc;
        if (h->dU_flagged) device_setup = false;    // prolongation rows with more than 3 entries: host planner and host RAP
        if (part && !device_setup) return fail(h, GMG_ERR_UNSUPPORTED, "a partitioned set-up has no host fallback (a prolongation row has more than 3 entries)");
    }
    struct LevelStage {
        SellHost sa, sin, sout, sp, sr;
        BlockCsrHost bc, bin;
        bool use_bcsr = false, use_ep = false;
        std::vector<unsigned short> ep16;
        std::vector<double> dg;
        std::vector<unsigned short> c16;
        std::string err;
        bool ok = true;
        double ms_order = 0, ms_sell = 0;
    };
    std::vector<LevelStage> stage(L + 1);
    std::vector<std::shared_future<void>> ord_done(L + 1);
    std::vector<std::future<void>> op_done(L), tr_done(L);
    std::future<bool> factor_done;
    double ms_factor = 0;
    // Host copy of the LHS (kept for gmg_get_level_operator, the level-0 ordering and the host fallbacks): 250 MB at
    // 3 M vertices, made in the background while the device works from the caller's arrays.
    // Host copy of the LHS: only where a host stage needs it (host RAP, host planner, block ordering of level 0); the
    // default path works from the caller's arrays and the device copy, and gmg_get_level_operator fetches on demand.
    double ms_lhs_copied = 0;
    const bool need_host_A0 = !device_setup || !h->cfg.device_rap;
    h->lv[0].n = n; h->lv[0].nnz = colptr[n];
    std::shared_future<void> lhs_copied;
    if (need_host_A0) {
        lhs_copied = std::async(std::launch::async, [&] { h->lv[0].A.assign(n, n, colptr, rowidx, val); ms_lhs_copied = ms_since(t_all); }).share();
        h->lv[0].hostA_pattern = h->lv[0].hostA_values = true;      // valid once lhs_copied is ready (every reader waits on it)
    }
    auto wait_lhs = [&] { if (lhs_copied.valid()) lhs_copied.wait(); };
    // pinned staging for one right-hand side (the solve's b / x transfers): page-locking costs milliseconds, do it now
    std::future<int> stage_ready = std::async(std::launch::async, [h, n] { (void)hipSetDevice(h->cfg.device); return ensure_host_stage(h, (size_t)n); });
    if (!have_key) pattern_key(n, colptr, rowidx, h->cfg.host_threads, pat_key);
    mark("pattern_key");
    // Level 0 as a blocked level (one launch per sweep instead of one per colour): asked for (block_from_level = 0), or chosen here
    // (gmg_config::block_fine) for an operator whose multicolour sweep would be a dozen small launches -- long rows -- and for which the
    // block-hybrid sweep is known to converge: positive diagonal, no positive off-diagonal entry (with the symmetric positive definite
    // system the method presumes, a Stieltjes matrix: D + in-block lower part is a regular splitting).  kNN graph Laplacians qualify;
    // meshes keep the colour-major sweep (4-7 colours, over-relaxed), Bilaplacians fail the sign test.
    const bool blocked0 = fine_level_blocked(h, n, colptr, rowidx, val);
    h->timing["fine_level_blocked"] = blocked0 ? 1.0 : 0.0;
    const bool ord_hit = h->ord_cache_valid && (int)h->ord_cache.size() == L + 1 && pat_key[0] == h->ord_cache_key[0] && pat_key[1] == h->ord_cache_key[1] &&
                         h->ord_cache[0].blocked == blocked0;
    h->timing["setup_ordering_cached"] = ord_hit ? 1.0 : 0.0;
    if (ord_hit || blocked0 || !mc || ahead.ptr != colptr) ahead.cancel();
    h->timing["setup_values_only"] = 0.0;
    h->ord_cache_valid = false;       // a hit moves the cached orderings into the levels; the next call moves them back
    std::shared_future<void> patches_done;      // hierarchies set level by level (gmg_set_prolongation): grown now, in the background
    if (!h->patches_ready && !ord_hit) patches_done = std::async(std::launch::async, [h] { build_patches(h); }).share();
    bool reorder0 = false, permuted0 = false;      // level-0 locality renumbering (decided below, before level 0 is spawned)
    std::atomic<int> colored_ahead{0};             // the level-0 ordering took the colouring made ahead of the verdict (AheadColoring)
    std::function<void(int)> spawn_level_ops;
    auto spawn_level = [&](int k) {
        ord_done[k] = std::async(std::launch::async, [&, k] {
            auto t = clk::now();
            Level& lk = h->lv[k];
            const bool blocked = mc && k < L && h->cfg.block_rows > 0 && (k >= h->cfg.block_from_level || (k == 0 && blocked0));
            if (ord_hit) lk.ord = std::move(h->ord_cache[k]);          // same pattern + same hierarchy => same orderings
            else if (k == L) lk.ord = identity_ordering(lk.n);
            else if (blocked) {
                if (patches_done.valid()) patches_done.wait();
                // level 0: blocks = runs of block_rows points of the hierarchy's cluster order (level0_patches), coloured from the caller's arrays
                // (row_align = 64 P: the block count is padded to a multiple of P, so that P ranks own whole blocks -- engine_dist.hip.hpp::p2p_smooth)
                if (k == 0) lk.ord = make_block_ordering(PatternView{n, colptr, rowidx}, h->cfg.block_rows, level0_patches(h, n), std::max(1, h->cfg.row_align / 64));
                else lk.ord = make_block_ordering(lk.A, h->cfg.block_rows, k < (int)h->patches.size() ? &h->patches[k] : nullptr);
            }
            else if (k == 0) {
                if (reorder0 && patches_done.valid()) patches_done.wait();
                const std::vector<int>* base = reorder0 && (int)base_order(h).size() == n ? &base_order(h) : nullptr;
                if (permuted0 && base) {
                    // colour the LHS pattern in cluster order (made on the device, see device_permute_pattern), then map back
                    LevelOrdering c = make_ordering(PatternView{n, h->reo_ptr.data(), h->reo_idx.data()}, mc, h->cfg.row_align, h->cfg.sigma, 0);
                    const int T = std::min(h->cfg.host_threads, 32);
                    parallel_ranges(c.n_pad, T, [&](int lo, int hi, int) { for (int r = lo; r < hi; ++r) if (c.new2old[r] >= 0) c.new2old[r] = (*base)[c.new2old[r]]; });
                    parallel_ranges(c.n_pad, T, [&](int lo, int hi, int) { for (int r = lo; r < hi; ++r) if (c.new2old[r] >= 0) c.old2new[c.new2old[r]] = r; });
                    c.reordered = true;
                    lk.ord = std::move(c);
                } else {
                    PreColoring* pre = nullptr;
                    if (!reorder0 && !base && mc && ahead.fut.valid() && ahead.ptr == colptr && !ahead.stop.load()) {
                        ahead.pre.n_colors = ahead.fut.get();
                        if (ahead.pre.n_colors >= 0) pre = &ahead.pre;
                    } else ahead.cancel();
                    colored_ahead.store(pre ? 1 : 0);
                    lk.ord = make_ordering(PatternView{n, colptr, rowidx}, mc, h->cfg.row_align, h->cfg.sigma, reorder0 ? 1 : 0, base, /*idx_sorted=*/true, pre);   // the caller's arrays (canonical: checked / canonicalised at entry)
                }
            }
            else lk.ord = make_ordering(lk.A, mc, h->cfg.row_align, h->cfg.sigma);
            lk.n_pad = lk.ord.n_pad;
            stage[k].ms_order = ms_since(t);
        }).share();
        if (k == L || device_setup) return;
        spawn_level_ops(k);
    };
    spawn_level_ops = [&](int k) {
        op_done[k] = std::async(std::launch::async, [&, k] {
            ord_done[k].wait();
            if (k == 0) wait_lhs();      // level 0 is ordered from the caller's arrays, but laid out from the host copy: that copy must be complete
            auto t = clk::now();
            Level& lk = h->lv[k];
            LevelStage& st = stage[k];
            // lanes per row on a blocked level: the quad layout pays where the level is latency-bound (few wavefronts);
            // a big level is throughput-bound and keeps one lane per row (single-wave blocks, no cross-wave barriers).
            // Level 0 always keeps one lane per row: the residual-norm kernels read its operator in that layout.
            const int lanes_auto = lk.n < kQuadLevelRows ? 4 : 1;
            const int lpr = (lk.ord.blocked && k > 0) ? (h->cfg.block_lanes ? h->cfg.block_lanes : lanes_auto) : 1;
            if (lk.ord.n_colors > 255) { st.ok = false; st.err = "more than 255 colours"; return; }
            if (!build_operator_sell(lk.A, lk.ord, lpr, st.sa, st.dg, st.err)) { st.ok = false; return; }
            if (lk.ord.blocked && wants_block_ep(h, lpr)) {
                build_operator_blockcsr(lk.A, lk.ord, st.bc, 3);       // "explicit" part
                build_operator_blockcsr(lk.A, lk.ord, st.bin, 4);      // "lower" part
                st.use_ep = st.bc.max_block_entries <= kEpMaxBlockEntries && st.bin.max_block_entries <= kEpMaxBlockLower;
                if (st.use_ep) {
                    st.ep16.resize(st.bin.col.size());
                    for (size_t i = 0; i < st.ep16.size(); ++i) st.ep16[i] = (unsigned short)st.bin.col[i];
                }
            }
            if (lk.ord.blocked && !st.use_ep && wants_block_csr(h, lpr)) {
                build_operator_blockcsr(lk.A, lk.ord, st.bc);
                st.use_bcsr = st.bc.max_block_entries <= kBcsrMaxBlockEntries;
            }
            if (lk.ord.blocked && !st.use_ep) {
                build_operator_sell_split(lk.A, lk.ord, st.sin, st.sout, lpr);
                st.c16.resize(st.sin.col.size());
                parallel_ranges((int)st.sin.col.size(), h->cfg.host_threads, [&](int lo, int hi, int) { for (int i = lo; i < hi; ++i) st.c16[i] = (unsigned short)st.sin.col[i]; });
            }
            st.ms_sell += ms_since(t);
        });
    };
    auto spawn_transfer = [&](int k) {
        if (device_setup) return;
        tr_done[k] = std::async(std::launch::async, [&, k] {
            ord_done[k].wait();
            ord_done[k + 1].wait();
            auto t = clk::now();
            Level& lk = h->lv[k];
            Compressed Urows = transpose_parallel(h->U[k]);                            // outer = fine rows
            stage[k].sp = build_transfer_sell(Urows, lk.ord, h->lv[k + 1].ord, 0);
            stage[k].sr = build_transfer_sell(h->U[k], h->lv[k + 1].ord, lk.ord, h->cfg.restrict_sigma > 0 ? h->cfg.restrict_sigma : 0,
                                              h->cfg.block_lanes == 1 ? 1 : 4);       // outer = coarse rows (~18 entries each)
            stage[k].ms_sell += ms_since(t);
        });
    };
    auto join_tasks = [&] {     // never leave with tasks still referencing this frame
        wait_lhs();
        if (patches_done.valid()) patches_done.wait();
        if (stage_ready.valid()) (void)stage_ready.get();
        for (int j = 0; j <= L; ++j) if (ord_done[j].valid()) ord_done[j].wait();
        for (int j = 0; j < L; ++j) { if (op_done[j].valid()) op_done[j].wait(); if (tr_done[j].valid()) tr_done[j].wait(); }
    };
    // Whatever way this frame is left (an exception of a host stage included), no task may outlive the locals it
    // references: the guard is declared after all of them, so it runs first.
    struct JoinGuard { std::function<void()> f; ~JoinGuard() { try { f(); } catch (...) {} } } join_guard{join_tasks};
    auto t0 = clk::now();
    // Level 0 of a badly numbered input (random-order scans, point clouds) is renumbered for locality.  With the
    // hierarchy's cluster order at hand the LHS pattern is permuted on the device first, so that the (sequential) greedy
    // colouring runs on a locally ordered graph; that needs the LHS on the device before the ordering task starts.
    reorder0 = mc && !ord_hit && !blocked0 && wants_locality_reorder(PatternView{n, colptr, rowidx}, h->cfg.reorder_fine);
    if (reorder0) ahead.cancel();                  // (the level is coloured along another visit order: the loop started on the caller's order is of no use)
    bool A0_uploaded = false;
    if (early_upload && device_setup && h->cfg.device_rap) { free_csr(h->lv[0].dA); h->lv[0].dA = early_A0; early_A0 = DevCsr(); A0_uploaded = true; }
    // (h->cluster_order is only read once the patches are ready: build_patches may still be writing it)
    const bool have_bfs = (int)h->bfs_order.size() == n, have_cluster = h->patches_ready && (int)h->cluster_order.size() == n;
    h->base_order_choice = (reorder0 && h->patches_ready && have_bfs && !have_cluster) ? 1 : 0;
    if (reorder0 && have_bfs && have_cluster && !(device_setup && h->cfg.device_rap)) {
        // host-planner path: the same decision from the host twin of the device score (choose_base_order)
        unsigned long long sc[2], sb[2];
        order_gather_score_host(PatternView{n, colptr, rowidx}, h->cluster_order, 4096, sc);
        order_gather_score_host(PatternView{n, colptr, rowidx}, h->bfs_order, 4096, sb);
        h->base_order_choice = (sc[1] && sb[1] && (double)sb[0] / (double)sb[1] < (double)sc[0] / (double)sc[1]) ? 1 : 0;
        h->timing["base_order_score_cluster"] = sc[1] ? (double)sc[0] / (double)sc[1] : 0.0;
        h->timing["base_order_score_bfs"] = sb[1] ? (double)sb[0] / (double)sb[1] : 0.0;
        h->timing["base_order_choice"] = h->base_order_choice;
    }
    if (reorder0 && device_setup && h->cfg.device_rap && (have_cluster || (h->patches_ready && have_bfs))) {
        int rc = A0_uploaded ? GMG_OK : upload_csr_raw(h, h->lv[0].dA, n, colptr, rowidx, val);
        if (rc == GMG_OK) { A0_uploaded = true; rc = choose_base_order(h, h->lv[0].dA, n); }
        if (rc == GMG_OK) rc = device_permute_pattern(h, h->lv[0].dA, n, colptr[n]);
        if (rc != GMG_OK) { join_tasks(); return rc; }
        permuted0 = true;
        mark("permuted_pattern");
    }
    spawn_level(0);
    auto host_level_from_A = [&](int k) { Level& l = h->lv[k]; l.n = l.A.n_outer; l.nnz = l.A.nnz(); l.hostA_pattern = l.hostA_values = true; };
    // The device keeps A_k (Level::dA) and U_k (h->dU, h->dE3, built once per hierarchy) in natural numbering: inputs of
    // the device RAP and of the device layout builder, and the source of the on-demand host copies.
    DevTmp<int> d_rap_err;
    bool device_rap_ok = device_setup && h->cfg.device_rap != 0;
    if (device_setup) {
        int rc = d_rap_err.alloc(h, 1);
        if (rc == GMG_OK) rc = hipMemsetAsync(d_rap_err.p, 0, sizeof(int), h->stream) == hipSuccess ? GMG_OK : GMG_ERR_HIP;
        if (rc != GMG_OK) { join_tasks(); return rc; }
    }
    if (device_rap_ok) {
        int rc = A0_uploaded ? GMG_OK : upload_csr_raw(h, h->lv[0].dA, n, colptr, rowidx, val);
        mark("upload_A0");
        int k = 1;
        for (; k <= L && rc == GMG_OK; ++k) {
            Level& lk = h->lv[k];
            const bool want_pattern = !ord_hit && k < L, want_values = k == L;
            rc = device_rap(h, h->lv[k - 1].dA, h->dU[k - 1], h->dE3[k - 1], lk.dA, lk.A, want_pattern, want_values, &lk.nnz, d_rap_err.p);
            if (rc != GMG_OK) break;
            lk.n = lk.dA.n_outer;
            lk.hostA_pattern = want_pattern || want_values; lk.hostA_values = want_values;
            spawn_level(k);
            mark("rap_l" + std::to_string(k));
        }
        if (rc != GMG_OK && rc != 1) { join_tasks(); return rc; }
        if (rc == 1) {
            // a coarse row with more distinct columns than the device hash set holds (or a U row with > 3 entries):
            // finish the chain with the host implementation
            device_rap_ok = false;
            wait_lhs();
            if ((rc = ensure_host_A(h, k - 1, true))) { join_tasks(); return rc; }
            for (; k <= L; ++k) { h->lv[k].A = galerkin_rap(h->lv[k - 1].A, h->U[k - 1], h->cfg.host_threads); host_level_from_A(k); spawn_level(k); }
        }
    } else {
        wait_lhs();
        for (int k = 1; k <= L; ++k) {
            h->lv[k].A = galerkin_rap(h->lv[k - 1].A, h->U[k - 1], h->cfg.host_threads);
            host_level_from_A(k);
            spawn_level(k);
            spawn_transfer(k - 1);
        }
    }
    h->timing["reduction"] = ms_since(t0);
    factor_done = std::async(std::launch::async, [&] {
        auto t = clk::now();
        bool ok = h->coarse.factor(h->lv[L].A, ord_hit);
        h->coarse_warm = false;
        ms_factor = ms_since(t);
        return ok;
    });
    int factor_state = -1;                          // (the future is read once: by the early inverse below, or at the end)
    auto factor_result = [&]() -> bool { if (factor_state < 0) factor_state = factor_done.get() ? 1 : 0; return factor_state == 1; };
    bool inverse_built = false;
    auto tl = clk::now();
    double ms_h2d = 0;
    int rc_all = GMG_OK;
    std::string err_all;
    if (device_setup) {
        // -- layouts built on the device from the raw matrices + orderings (setup_kernels.hip.hpp)
        DevTmp<int> d_err;
        int herr = 0;
        if ((rc_all = d_err.alloc(h, 1)) == GMG_OK) {
            (void)hipMemsetAsync(d_err.p, 0, sizeof(int), h->stream);
            // the coarse levels first: their orderings are short jobs, while level 0's (a sequential greedy colouring of
            // the whole mesh) is the longest host task of the set-up and may still be running
            auto ordering_of = [&](int k) {
                auto tw = clk::now();
                // (an exception of the ordering task becomes an error code here: unwinding past the other tasks' futures
                // would free what they still reference)
                try { ord_done[k].get(); } catch (const std::exception& e) { rc_all = GMG_ERR_STATE; err_all = std::string("ordering of level ") + std::to_string(k) + ": " + e.what(); return; }
                h->timing["setup_wait_ordering"] += ms_since(tw);
                mark("ordering_ready_l" + std::to_string(k));
                if (h->lv[k].ord.n_colors > 255) { rc_all = GMG_ERR_UNSUPPORTED; err_all = "more than 255 colours on level " + std::to_string(k); return; }
                rc_all = upload(h, &h->lv[k].d_new2old, h->lv[k].ord.new2old);
            };
            double ms_layout = 0;
            for (int k = L; k >= 1 && rc_all == GMG_OK; --k) ordering_of(k);
            // A partitioned set-up (gmg_dist_partition) lays out this rank's rows of levels 0 / 1 only: it needs the partition plan -- hence
            // both orderings -- before the first layout; everybody else's rows are masked out of the row maps the builders read
            int *d_mask0 = nullptr, *d_mask1 = nullptr;
            struct MaskGuard { int*& a; int*& b; ~MaskGuard() { if (a) (void)dev_free(a); if (b) (void)dev_free(b); } } mask_guard{d_mask0, d_mask1};
            bool shard1 = false;
            if (part && rc_all == GMG_OK) {
                ordering_of(0);
                auto tp = clk::now();
                const LevelOrdering& o0 = h->lv[0].ord;
                if (rc_all == GMG_OK && o0.blocked) { rc_all = GMG_ERR_STATE; err_all = "a partitioned set-up needs the colour-major level 0 (block_from_level >= 1)"; }
                for (int c = 0; c < o0.n_colors && rc_all == GMG_OK; ++c)
                    if ((o0.color_begin[c + 1] - o0.color_begin[c]) % (64 * h->part_world)) { rc_all = GMG_ERR_STATE; err_all = "colour classes are not aligned to 64*world rows: create the handle with row_align = 64*world"; }
                if (rc_all == GMG_OK) {
                    shard1 = plan_can_shard_level1(h, h->part_world, false);
                    const bool reuse = h->plan && h->plan->key[0] == pat_key[0] && h->plan->key[1] == pat_key[1] && h->plan->rank == h->part_rank &&
                                       h->plan->world == h->part_world && h->plan->shard1 == shard1 && h->plan->n_colors == o0.n_colors;
                    if (!reuse) {
                        if (shard1) rc_all = ensure_host_A(h, 1, false);
                        auto plan = std::make_shared<DistPlan>();
                        if (rc_all == GMG_OK) rc_all = build_dist_plan(h, *plan, h->part_rank, h->part_world, PatternView{n, colptr, rowidx}, shard1 ? &h->lv[1].A : nullptr, shard1);
                        plan->key[0] = pat_key[0]; plan->key[1] = pat_key[1];
                        if (rc_all == GMG_OK) h->plan = plan;
                    }
                    h->timing["dist_plan_cached"] = reuse ? 1.0 : 0.0;
                }
                if (rc_all == GMG_OK) rc_all = make_row_masks(h, *h->plan, &d_mask0, &d_mask1);
                h->timing["dist_plan_ms"] = ms_since(tp);
            }
            auto tlay = clk::now();
            for (int k = 1; k < L && rc_all == GMG_OK; ++k) rc_all = device_layout_level(h, k, d_err.p, (k == 1 && shard1) ? d_mask1 : nullptr, nullptr);
            if (part && shard1 && rc_all == GMG_OK && !h->lv[1].use_ep) { rc_all = GMG_ERR_UNSUPPORTED; err_all = "level 1 cannot run the entry-parallel block sweep (a block is too large for its LDS buffers): create the handle with dist_shard_levels = 1"; }
            ms_layout += ms_since(tlay);
            // The device has nothing to do until the ordering of level 0 arrives (a sequential colouring on the host): when the coarsest factor is
            // there first, the dense inverse of the coarsest operator is built in that gap instead of at the end of the call
            if (rc_all == GMG_OK && !part && !h->preparing_structure && want_coarse_device(h, h->lv[L].A.n_outer)) {
                while (ord_done[0].wait_for(std::chrono::seconds(0)) != std::future_status::ready &&
                       factor_done.valid() && factor_done.wait_for(std::chrono::microseconds(200)) != std::future_status::ready) {}
                if (factor_state >= 0 || (factor_done.valid() && factor_done.wait_for(std::chrono::seconds(0)) == std::future_status::ready)) {
                    if (factor_result()) {
                        h->coarse_device = true;
                        rc_all = build_coarse_inverse_device(h);
                        inverse_built = rc_all == GMG_OK;
                        mark("coarse_inverse_early");
                    }
                }
            }
            if (rc_all == GMG_OK && !part) ordering_of(0);
            tlay = clk::now();
            if (rc_all == GMG_OK) rc_all = device_layout_level(h, 0, d_err.p, part ? d_mask0 : nullptr, (part && shard1) ? d_mask1 : nullptr);
            ms_layout += ms_since(tlay);
            h->timing["setup_device_layout"] = ms_layout;
            mark("device_layout");
            if (rc_all == GMG_OK) {
                (void)hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, h->stream);
                (void)hipStreamSynchronize(h->stream);
                if (herr == 2) { rc_all = GMG_ERR_NUMERIC; err_all = "system matrix has a missing or zero diagonal entry"; }
                else if (herr != 0 && part) { rc_all = GMG_ERR_UNSUPPORTED; err_all = "rows too long for the device layout builder: a partitioned set-up has no host fallback"; }
                else if (herr != 0) {
                    // rows too long for the device builder: redo the layout with the host planner
                    device_setup = false;
                    wait_lhs();
                    for (int k = 0; k < L && rc_all == GMG_OK; ++k) rc_all = ensure_host_A(h, k, true);
                    for (int k = 0; k < L && rc_all == GMG_OK; ++k) spawn_level_ops(k);
                    for (int k = 0; k < L && rc_all == GMG_OK; ++k) spawn_transfer(k);
                }
            }
        }
        ms_h2d = ms_since(tl);
    }
    // -- host-planned layouts: uploads in the order the stages complete (level 0 first: the largest, ready early)
    for (int k = 0; k <= L && rc_all == GMG_OK && !device_setup; ++k) {
        Level& l = h->lv[k];
        int rc;
        try { ord_done[k].get(); } catch (const std::exception& e) { rc_all = GMG_ERR_STATE; err_all = std::string("ordering of level ") + std::to_string(k) + ": " + e.what(); break; }
        auto tu = clk::now();
        if ((rc = upload(h, &l.d_new2old, l.ord.new2old))) { rc_all = rc; break; }
        ms_h2d += ms_since(tu);
        if (k == L) break;
        op_done[k].get();
        LevelStage& st = stage[k];
        if (!st.ok) { rc_all = GMG_ERR_NUMERIC; err_all = "level " + std::to_string(k) + ": " + st.err; break; }
        tu = clk::now();
        if ((rc = upload_sell(h, l.Aoff, st.sa)) || (rc = upload(h, &l.diag, st.dg))) { rc_all = rc; break; }
        if (l.ord.blocked && st.use_ep) {
            l.use_ep = true;
            l.ee_nnz = st.bc.ptr[l.n_pad]; l.ep_nnz = st.bin.ptr[l.n_pad];
            l.ep_cap_e = (st.bc.max_block_entries + 63) / 64 * 64; l.ep_cap_l = std::max(st.bin.max_block_entries, 1);
            if ((rc = upload(h, &l.ee_ptr, st.bc.ptr)) || (rc = upload(h, &l.ee_col, st.bc.col)) || (rc = upload(h, &l.ee_val, st.bc.val)) ||
                (rc = upload(h, &l.ep_ptr, st.bin.ptr)) || (rc = upload(h, &l.ep_col, st.ep16)) || (rc = upload(h, &l.ep_val, st.bin.val)) ||
                (rc = upload(h, &l.d_blk_begin, l.ord.blk_begin)) || (rc = upload(h, &l.d_blk_ncolors, l.ord.blk_ncolors)) ||
                (rc = upload(h, &l.d_row_color, l.ord.row_color))) { rc_all = rc; break; }
        } else if (l.ord.blocked && st.use_bcsr) {
            l.use_bcsr = true;
            l.bc_cap = (st.bc.max_block_entries + 63) / 64 * 64;
            l.bc_nnz = st.bc.ptr[l.n_pad];
            if ((rc = upload_sell(h, l.Ain, st.sin)) || (rc = upload_sell(h, l.Aout, st.sout)) || (rc = upload(h, &l.ain_col16, st.c16)) ||
                (rc = upload(h, &l.bc_ptr, st.bc.ptr)) || (rc = upload(h, &l.bc_mid, st.bc.mid)) || (rc = upload(h, &l.bc_col, st.bc.col)) ||
                (rc = upload(h, &l.bc_val, st.bc.val)) || (rc = upload(h, &l.d_blk_begin, l.ord.blk_begin)) ||
                (rc = upload(h, &l.d_blk_ncolors, l.ord.blk_ncolors)) || (rc = upload(h, &l.d_row_color, l.ord.row_color))) { rc_all = rc; break; }
        } else if (l.ord.blocked) {
            if ((rc = upload_sell(h, l.Ain, st.sin)) || (rc = upload_sell(h, l.Aout, st.sout)) || (rc = upload(h, &l.ain_col16, st.c16)) ||
                (rc = upload(h, &l.d_blk_begin, l.ord.blk_begin)) || (rc = upload(h, &l.d_blk_ncolors, l.ord.blk_ncolors)) ||
                (rc = upload(h, &l.d_row_color, l.ord.row_color))) { rc_all = rc; break; }
        }
        ms_h2d += ms_since(tu);
    }
    for (int k = 0; k < L && rc_all == GMG_OK && !device_setup; ++k) {
        Level& l = h->lv[k];
        int rc;
        tr_done[k].get();
        auto tu = clk::now();
        if ((rc = upload_sell(h, l.P, stage[k].sp)) || (rc = upload_sell(h, l.R, stage[k].sr))) { rc_all = rc; break; }
        ms_h2d += ms_since(tu);
    }
    join_tasks();
    h->timing["t_lhs_copied"] = ms_lhs_copied;
    h->timing["setup_colored_ahead"] = colored_ahead.load();
    mark("tasks_joined");
    const bool factor_ok = factor_result();
    mark("factor_joined");
    (void)hipStreamSynchronize(h->stream);      // staged host arrays die at scope end
    if (rc_all != GMG_OK) return err_all.empty() ? rc_all : fail(h, rc_all, err_all);
    if (!factor_ok) return fail(h, GMG_ERR_NUMERIC, "coarsest operator is singular (LDL^T hit a zero pivot)");
    h->coarse_work.assign((size_t)h->lv[L].A.n_outer * 4, 0.0);       // grown by coarse_host_roundtrip for more than 4 columns
    h->timing["coarsest_solve"] = ms_factor;
    h->timing["setup_ordering"] = 0.0; h->timing["setup_sell"] = 0.0;
    for (int k = 0; k <= L; ++k) h->timing["setup_ordering_l" + std::to_string(k)] = stage[k].ms_order;
    for (int k = 0; k <= L; ++k) { h->timing["setup_ordering"] = std::max(h->timing["setup_ordering"], stage[k].ms_order); h->timing["setup_sell"] = std::max(h->timing["setup_sell"], stage[k].ms_sell); }
    h->timing["setup_h2d"] = ms_h2d;
    (void)tl;
    {
        int nblk = std::max(kNormBlocks, grid_for((h->lv[0].n_pad + 63) / 64));        // one partial per four level-0 slices
        if (nblk > h->partial_blocks) {
            if (h->d_partials) (void)dev_free(h->d_partials);
            HIPCHK(dev_malloc((void**)&h->d_partials, sizeof(double) * (size_t)nblk * 8));
            h->partial_blocks = nblk;
        }
    }
    h->coarse_device = want_coarse_device(h, h->lv[L].A.n_outer);
    h->timing["coarse_on_device"] = h->coarse_device ? 1.0 : 0.0;
    if (h->coarse_device && !h->preparing_structure && !inverse_built) {
        int rc = build_coarse_inverse_device(h);
        if (rc) return rc;
    }
    if (h->cfg.inner_precision) {
        // fp32 twins of every value array (same layout): the inner V-cycle of the mixed-precision iteration
        int rc = refresh_fp32_twins(h, true);
        if (rc) return rc;
    }
    if (!h->mass.empty()) {
        if ((int)h->mass.size() != n) return fail(h, GMG_ERR_INVALID, "mass size does not match the system");
        int rc = upload_mass(h);
        if (rc) return rc;
        h->mass_dirty = false;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->timing["device_bytes_peak"] = (double)h->pool.peak_live_bytes;
    if (part) {
        // This rank's rows of levels 0 / 1 are laid out; what the set-up needed in full -- A_0, A_1 and U_0 in natural numbering (inputs of the
        // Galerkin chain and of the layout builders) and the column maps of the two levels -- goes back to the pool.  A later system pays for
        // them again (no values-only refresh on a partitioned handle); the orderings and the plan stay cached under the pattern digest.
        const bool s1 = h->plan && h->plan->shard1;
        free_csr(h->lv[0].dA);
        if (s1) free_csr(h->lv[1].dA);
        if (!h->dU.empty()) { free_csr(h->dU[0]); free_ell3(h->dE3[0]); }
        for (int k = 0; k <= (s1 ? 1 : 0); ++k) {
            Level& lk = h->lv[k];
            if (lk.d_old2new) { (void)dev_free(lk.d_old2new); lk.d_old2new = nullptr; }
            if (lk.d_blk_of_row) { (void)dev_free(lk.d_blk_of_row); lk.d_blk_of_row = nullptr; }
        }
        h->pool.trim_large((size_t)4 << 20);   // (the big temporaries go back to the device, not to this handle's pool)
        h->partitioned = true;
    }
    h->timing["device_bytes"] = (double)h->pool.live_bytes;
    // only now is there a system: a failure above leaves the handle without one (no solves on a half-built state)
    h->live_key[0] = pat_key[0]; h->live_key[1] = pat_key[1];
    h->live_key_valid = true;
    h->ord_cache_valid = false;       // (moved into the levels on a hit; refilled from them by the next call)
    h->system_ready = true;
    h->refill_ready = device_setup && device_rap_ok && h->cfg.device_setup != 0 && !part;
    mark("mass_done");
    h->timing["upload"] = ms_since(t_all) - h->timing["reduction"];      // everything of the setup that is not the RAP chain
    h->timing["setup_total"] = ms_since(t_all);                          // wall time of this call (the coarsest factorisation overlaps)
    h->timing["coarse_host_ms"] = 0.0;
    h->timing["setup_structure_prepared"] = 0.0;
    return GMG_OK;
}

int gmg_set_system(gmg_handle h, int n, const int* colptr, const int* rowidx, const double* val) try {
    return set_system_impl(h, n, colptr, rowidx, val);
} GMG_CATCH_H

// gmg_finalize_hierarchy with a fine graph (gmg_set_fine_graph / gmg_use_hierarchy): the complete set-up for a PLACEHOLDER matrix with the
// graph's pattern -- diagonally dominant (row i: its entry count on the diagonal, -1 elsewhere: symmetric positive definite, and of the sign
// structure of the graph Laplacians the hierarchy is built for, so that the rules that look at values -- gmg_config::block_fine -- decide as
// they will for the real system; a system that makes them decide otherwise takes the cold path).  Everything structural then stands on the
// device: orderings and colourings of all levels, SELL / block layouts, 16-bit column codes, the patterns of the Galerkin operators, the
// symbolic LDL^T.  The handle holds no system afterwards (solves are refused until gmg_set_system), but a gmg_set_system whose pattern
// digest equals the prepared one is a values-only refresh: values up, numeric Galerkin passes, layout refill, numeric LDL^T -- the part of
// the reference's solve() preamble (multigrid_solver.cpp:1387-1401) that depends on the matrix, and nothing else.
static int prepare_structure(gmg_handle h) {
    const FineGraph& g = *h->fine_graph;
    const int n = g.n;
    auto t0 = clk::now();
    {   // The systems to come are symmetric (tau M + S): a point graph that is not (the kNN table of a point cloud: j among i's neighbours, i not among
        // j's) is not their pattern -- its placeholder set-up would be paid here and the first real system would take the cold path all the same
        // (round-5 advice).  One threaded pass, a binary search per entry (the rows are sorted).
        std::atomic<bool> symmetric{true};
        parallel_ranges(n, h->cfg.host_threads, [&](int lo, int hi, int) {
            for (int i = lo; i < hi && symmetric.load(std::memory_order_relaxed); ++i)
                for (int p = g.ptr[i]; p < g.ptr[i + 1]; ++p) {
                    const int j = g.idx[p];
                    if (j == i) continue;
                    if (!std::binary_search(g.idx.data() + g.ptr[j], g.idx.data() + g.ptr[j + 1], i)) { symmetric.store(false, std::memory_order_relaxed); break; }
                }
        }, 1 << 14);
        h->timing["structure_prepare_symmetric_graph"] = symmetric.load() ? 1.0 : 0.0;
        if (!symmetric.load()) { h->timing["structure_prepare_ms"] = ms_since(t0); return GMG_OK; }
    }
    RawVec<double> val;
    val.resize((size_t)g.ptr[n]);
    parallel_ranges(n, h->cfg.host_threads, [&](int lo, int hi, int) {
        for (int i = lo; i < hi; ++i) {
            const double diag = (double)(g.ptr[i + 1] - g.ptr[i]);
            for (int p = g.ptr[i]; p < g.ptr[i + 1]; ++p) val[p] = g.idx[p] == i ? diag : -1.0;
        }
    }, 1 << 14);
    std::vector<double> user_mass;
    user_mass.swap(h->mass);                         // (the placeholder set-up needs no mass; whatever the caller set waits for the real system)
    struct Flag { bool& f; explicit Flag(bool& x) : f(x) { f = true; } ~Flag() { f = false; } };
    int rc;
    {
        Flag preparing(h->preparing_structure);     // (no dense coarse inverse of placeholder values: the refresh with the real ones builds it)
        rc = set_system_impl(h, n, g.ptr.data(), g.idx.data(), val.data());
    }
    h->mass.swap(user_mass);
    if (rc != GMG_OK) { h->placeholder_ready = false; return rc; }
    h->system_ready = false;                         // nothing to solve with: the values are placeholders
    if (h->part_world > 1) {
        // a partitioned handle cannot refresh values in place (it keeps no whole operator): what carries over to the real system is what
        // depends on the pattern alone and lives on the host -- the orderings of all levels and the partition plan, both under the digest
        h->ord_cache.resize(h->L + 1);
        for (int k = 0; k <= h->L; ++k) h->ord_cache[k] = std::move(h->lv[k].ord);
        h->ord_cache_key[0] = h->live_key[0]; h->ord_cache_key[1] = h->live_key[1];
        h->ord_cache_valid = true;
        h->live_key_valid = false;
        drop_system(h);
        h->pool.trim_large((size_t)4 << 20);
    }
    h->placeholder_ready = h->refill_ready && h->live_key_valid;
    // (the level vectors of a one-column problem, so that the first solve does not pay their allocation either; a wider block re-allocates)
    if (h->placeholder_ready) { (void)ensure_vectors(h, 1); (void)ensure_rap_order(h); }
    h->mass_dirty = !h->mass.empty();
    h->timing["structure_prepare_ms"] = ms_since(t0);
    return GMG_OK;
}

int gmg_num_levels(gmg_handle h) { return h ? h->L : GMG_ERR_INVALID; }

int gmg_level_info(gmg_handle h, int k, int* n, int64_t* nnz, int* n_colors, int* n_pad) try {
    if (!h) return GMG_ERR_INVALID;
    int rc = check_level(h, k, true);
    if (rc) return rc;
    Level& l = h->lv[k];
    if (n) *n = l.n;
    if (nnz) *nnz = l.nnz;
    if (n_colors) *n_colors = l.ord.n_colors;
    if (n_pad) *n_pad = l.n_pad;
    return GMG_OK;
} GMG_CATCH_H

int gmg_get_level_operator(gmg_handle h, int k, int* colptr, int* rowidx, double* val) try {
    if (!h) return GMG_ERR_INVALID;
    int rc = check_level(h, k, true);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    {
        PoolScope pool_scope_(&h->pool);
        if ((rc = ensure_host_A(h, k, true))) return rc;
    }
    const Compressed& A = h->lv[k].A;
    if (colptr) std::memcpy(colptr, A.ptr.data(), sizeof(int) * (A.n_outer + 1));
    if (rowidx) std::memcpy(rowidx, A.idx.data(), sizeof(int) * A.nnz());
    if (val) std::memcpy(val, A.val.data(), sizeof(double) * A.nnz());
    return GMG_OK;
} GMG_CATCH_H

int gmg_get_level_ordering(gmg_handle h, int k, int* new2old, int* color_begin) try {
    if (!h) return GMG_ERR_INVALID;
    int rc = check_level(h, k, true);
    if (rc) return rc;
    const LevelOrdering& o = h->lv[k].ord;
    if (new2old) std::memcpy(new2old, o.new2old.data(), sizeof(int) * o.n_pad);
    if (color_begin) {      // colour-major levels: n_colors + 1 entries; blocked levels: {0, n_pad} (one range)
        for (int c = 0; c <= o.n_colors; ++c) color_begin[c] = c < (int)o.color_begin.size() ? o.color_begin[c] : o.n_pad;
    }
    return GMG_OK;
} GMG_CATCH_H

int gmg_get_level_blocks(gmg_handle h, int k, int* n_blocks, int* blk_begin, unsigned char* row_color) try {
    if (!h) return GMG_ERR_INVALID;
    int rc = check_level(h, k, true);
    if (rc) return rc;
    const LevelOrdering& o = h->lv[k].ord;
    if (n_blocks) *n_blocks = o.blocked ? o.n_blocks() : 0;
    if (o.blocked && blk_begin) std::memcpy(blk_begin, o.blk_begin.data(), sizeof(int) * o.blk_begin.size());
    if (o.blocked && row_color) std::memcpy(row_color, o.row_color.data(), o.row_color.size());
    return GMG_OK;
} GMG_CATCH_H

// Debug / test access to the device-resident layouts: which = 0 A (off-diagonal), 1 A_in, 2 A_out, 3 P (U), 4 R (U^T).
static DevSell* pick_sell(gmg_handle h, int k, int which) {
    Level& l = h->lv[k];
    switch (which) { case 0: return &l.Aoff; case 1: return &l.Ain; case 2: return &l.Aout; case 3: return &l.P; case 4: return &l.R; default: return nullptr; }
}

int gmg_debug_sell_info(gmg_handle h, int k, int which, int64_t* info) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if (which == 5) {      // block-CSR of a big blocked level: reported as n_pad "slices" of one row (lpr 64: one row_of entry per row)
        Level& l = h->lv[k];
        if (!info) return fail(h, GMG_ERR_INVALID, "bad arguments");
        info[0] = l.use_bcsr ? l.n_pad : 0; info[1] = 64; info[2] = l.use_bcsr ? l.bc_nnz : 0; info[3] = l.use_bcsr ? 1 : 0;
        return GMG_OK;
    }
    if (which == 6 || which == 7) {      // unpadded block sweep: 6 = "lower" part (local columns), 7 = "explicit" part (device columns)
        Level& l = h->lv[k];
        if (!info) return fail(h, GMG_ERR_INVALID, "bad arguments");
        info[0] = l.use_ep ? l.n_pad : 0; info[1] = 64; info[2] = l.use_ep ? (which == 6 ? l.ep_nnz : l.ee_nnz) : 0; info[3] = l.use_ep ? 1 : 0;
        return GMG_OK;
    }
    DevSell* s = pick_sell(h, k, which);
    if (!s || !info) return fail(h, GMG_ERR_INVALID, "bad arguments");
    info[0] = s->n_slices; info[1] = s->lpr; info[2] = s->stored; info[3] = s->row_of ? 1 : 0;
    return GMG_OK;
} GMG_CATCH_H

int gmg_debug_sell_copy(gmg_handle h, int k, int which, int64_t* slice_ptr, int* col, double* val, int* row_of, double* diag) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (which == 5) {      // slice_ptr <- row_ptr (n_pad + 1), row_of <- row_mid (n_pad)
        Level& lb = h->lv[k];
        if (!lb.use_bcsr) return GMG_OK;
        HIPCHK(hipStreamSynchronize(h->stream));
        std::vector<int> tmp((size_t)lb.n_pad + 1);
        HIPCHK(hipMemcpy(tmp.data(), lb.bc_ptr, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost));
        if (slice_ptr) for (size_t i = 0; i < tmp.size(); ++i) slice_ptr[i] = tmp[i];
        if (row_of) HIPCHK(hipMemcpy(row_of, lb.bc_mid, sizeof(int) * (size_t)lb.n_pad, hipMemcpyDeviceToHost));
        if (col) HIPCHK(hipMemcpy(col, lb.bc_col, sizeof(int) * (size_t)lb.bc_nnz, hipMemcpyDeviceToHost));
        if (val) HIPCHK(hipMemcpy(val, lb.bc_val, sizeof(double) * (size_t)lb.bc_nnz, hipMemcpyDeviceToHost));
        return GMG_OK;
    }
    if (which == 6 || which == 7) {      // slice_ptr <- row pointers (n_pad + 1), col <- columns, row_of[0] <- LDS capacity of that part
        Level& lb = h->lv[k];
        if (!lb.use_ep) return GMG_OK;
        HIPCHK(hipStreamSynchronize(h->stream));
        const int64_t nnz = which == 6 ? lb.ep_nnz : lb.ee_nnz;
        std::vector<int> tmp((size_t)lb.n_pad + 1);
        HIPCHK(hipMemcpy(tmp.data(), which == 6 ? lb.ep_ptr : lb.ee_ptr, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost));
        if (slice_ptr) for (size_t i = 0; i < tmp.size(); ++i) slice_ptr[i] = tmp[i];
        if (row_of) { std::memset(row_of, 0, sizeof(int) * (size_t)lb.n_pad); row_of[0] = which == 6 ? lb.ep_cap_l : lb.ep_cap_e; }
        if (col && which == 6) {
            std::vector<unsigned short> c16((size_t)nnz);
            HIPCHK(hipMemcpy(c16.data(), lb.ep_col, sizeof(unsigned short) * c16.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < c16.size(); ++i) col[i] = c16[i];
        } else if (col) HIPCHK(hipMemcpy(col, lb.ee_col, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost));
        if (val) HIPCHK(hipMemcpy(val, which == 6 ? lb.ep_val : lb.ee_val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost));
        return GMG_OK;
    }
    DevSell* s = pick_sell(h, k, which);
    if (!s) return fail(h, GMG_ERR_INVALID, "bad arguments");
    Level& l = h->lv[k];
    HIPCHK(hipStreamSynchronize(h->stream));
    if (slice_ptr && s->slice_ptr) HIPCHK(hipMemcpy(slice_ptr, s->slice_ptr, sizeof(int64_t) * (s->n_slices + 1), hipMemcpyDeviceToHost));
    if (val && s->val) HIPCHK(hipMemcpy(val, s->val, sizeof(double) * s->stored, hipMemcpyDeviceToHost));
    if (col) {
        if (which == 1 && l.ain_col16) {
            std::vector<unsigned short> c16((size_t)s->stored);
            HIPCHK(hipMemcpy(c16.data(), l.ain_col16, sizeof(unsigned short) * s->stored, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < s->stored; ++i) col[i] = c16[i];
        } else if (s->col) HIPCHK(hipMemcpy(col, s->col, sizeof(int) * s->stored, hipMemcpyDeviceToHost));
    }
    if (row_of && s->row_of) HIPCHK(hipMemcpy(row_of, s->row_of, sizeof(int) * (size_t)s->n_slices * (64 / s->lpr), hipMemcpyDeviceToHost));
    if (diag && which == 0 && l.diag) HIPCHK(hipMemcpy(diag, l.diag, sizeof(double) * l.n_pad, hipMemcpyDeviceToHost));
    return GMG_OK;
} GMG_CATCH_H

int gmg_get_timing(gmg_handle h, const char* key, double* out) try {
    if (!h || !key || !out) return GMG_ERR_INVALID;
    if (std::string(key) == "device_bytes_now") { *out = (double)h->pool.live_bytes; return GMG_OK; }      // device memory the handle holds at this moment (pool blocks in use)
    auto it = h->timing.find(key);
    if (it == h->timing.end()) return fail(h, GMG_ERR_INVALID, std::string("unknown timing key: ") + key);
    *out = it->second;
    return GMG_OK;
} GMG_CATCH_H

// ---- operators -------------------------------------------------------------------------------------------

int gmg_smooth(gmg_handle h, int k, const double* b, double* x, int d, int iters) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!b || !x || d <= 0 || iters < 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[k];
    if ((rc = to_device(h, k, b, d, l.b))) return rc;
    if ((rc = to_device(h, k, x, d, l.x))) return rc;
    launch_smooth<double>(h, l, d, iters);
    h->loaded_d = 0;
    return to_host(h, k, l.x, d, x);
} GMG_CATCH_H

int gmg_smooth_residual(gmg_handle h, int k, const double* b, double* x, int d, int iters, int from_zero, double* r) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!b || !x || !r || d <= 0 || iters < 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[k];
    if ((rc = to_device(h, k, b, d, l.b))) return rc;
    const bool zero = from_zero != 0 && k > 0 && smooth_from_zero_ok(h, l, iters);
    if (from_zero) HIPCHK(hipMemsetAsync(l.x, 0, sizeof(double) * (size_t)l.n_pad * d, h->stream));
    else if ((rc = to_device(h, k, x, d, l.x))) return rc;
    h->sweep_prev_valid = false;
    h->first_sweep_fused = false;                  // (a single level's sweeps: nothing ran ahead of them)
    launch_smooth<double>(h, l, d, iters, zero);
    const bool delta = k > 0 && launch_residual_delta<double>(h, l, d, l.r);        // exactly what enqueue_down does
    if (!delta) launch_spmv<double>(h, l, d, 1, l.b, l.x, l.r);
    h->timing["residual_from_sweep"] = delta ? 1.0 : 0.0;
    h->loaded_d = 0;
    if ((rc = to_host(h, k, l.x, d, x))) return rc;
    return to_host(h, k, l.r, d, r);
} GMG_CATCH_H

int gmg_residual(gmg_handle h, int k, const double* b, const double* x, int d, double* r) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!b || !x || !r || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[k];
    if ((rc = to_device(h, k, b, d, l.b))) return rc;
    if ((rc = to_device(h, k, x, d, l.x))) return rc;
    launch_spmv<double>(h, l, d, 1, l.b, l.x, l.r);
    h->loaded_d = 0;
    return to_host(h, k, l.r, d, r);
} GMG_CATCH_H

int gmg_spmv(gmg_handle h, int k, const double* x, int d, double* y) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!x || !y || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[k];
    if ((rc = to_device(h, k, x, d, l.x))) return rc;
    launch_spmv<double>(h, l, d, 0, nullptr, l.x, l.r);
    h->loaded_d = 0;
    return to_host(h, k, l.r, d, y);
} GMG_CATCH_H

int gmg_restrict(gmg_handle h, int k, const double* r, int d, double* rc_out) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!r || !rc_out || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[k];
    if ((rc = to_device(h, k, r, d, l.r))) return rc;
    launch_restrict<double>(h, l, h->lv[k + 1], d, l.r, h->lv[k + 1].b);
    h->loaded_d = 0;
    return to_host(h, k + 1, h->lv[k + 1].b, d, rc_out);
} GMG_CATCH_H

int gmg_prolong_add(gmg_handle h, int k, const double* e, int d, double* x) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!e || !x || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[k];
    if ((rc = to_device(h, k + 1, e, d, h->lv[k + 1].x))) return rc;
    if ((rc = to_device(h, k, x, d, l.x))) return rc;
    launch_prolong_add<double>(h, l, h->lv[k + 1], d, h->lv[k + 1].x, l.x);
    h->loaded_d = 0;
    return to_host(h, k, l.x, d, x);
} GMG_CATCH_H

int gmg_coarse_solve(gmg_handle h, const double* rc_in, int d, double* e) try {
    NEED_DEVICE();
    int rc = check_level(h, h->L, true);
    if (rc) return rc;
    if (!rc_in || !e || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& c = h->lv[h->L];
    if ((rc = to_device(h, h->L, rc_in, d, c.b))) return rc;
    if (h->coarse_device) enqueue_coarse_device<double>(h, d);
    else if ((rc = coarse_host_roundtrip<double>(h, d))) return rc;
    h->loaded_d = 0;
    return to_host(h, h->L, c.x, d, e);
} GMG_CATCH_H

int gmg_residual_norm(gmg_handle h, const double* b, const double* x, int d, int type, double* out) try {
    NEED_DEVICE();
    int rc = check_level(h, 0, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!b || !x || !out || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = check_norm_type(h, type))) return rc;
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[0];
    if ((rc = to_device(h, 0, b, d, l.b))) return rc;
    if ((rc = to_device(h, 0, x, d, l.x))) return rc;
    if ((rc = launch_norm(h, d, type))) return rc;
    if ((rc = wait_norm(h))) return rc;
    h->loaded_d = 0;
    *out = norm_from_sums(h->h_norm, d, type);
    return GMG_OK;
} GMG_CATCH_H

// ---- resident problem: load / run / fetch ------------------------------------------------------------------

int gmg_load_problem(gmg_handle h, const double* b, const double* x0, int d) try {
    NEED_DEVICE();
    int rc = check_level(h, 0, false);
    if (rc) return rc;
    if (!b || !x0 || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    auto tl = clk::now();
    if ((rc = ensure_vectors(h, d))) return rc;
    h->timing["load_vectors"] = ms_since(tl); tl = clk::now();
    Level& l = h->lv[0];
    // The reference's Python API always starts from x0 = rhs (gravomg_bindings/src/cpp/core.cpp:69): when the initial guess IS
    // the right-hand side (same buffer, or the same content -- one threaded comparison, far cheaper than a second trip over
    // PCIe), it is copied on the device instead of being uploaded again.  The comparison runs beside the upload of b.
    bool same = x0 == b;
    std::future<bool> compared;
    if (!same) {
        const size_t cnt = (size_t)l.n * d;
        const int threads = std::min(h->cfg.host_threads, 16);
        compared = std::async(std::launch::async, [b, x0, cnt, threads] {
            std::atomic<bool> differ{false};
            parallel_ranges((int)std::min<size_t>(cnt >> 12, 1 << 20) + 1, threads, [&](int lo, int hi, int) {
                const size_t a = (size_t)lo << 12, e = std::min(cnt, (size_t)hi << 12);
                if (a < e && !differ.load(std::memory_order_relaxed) && std::memcmp(b + a, x0 + a, sizeof(double) * (e - a)) != 0) differ = true;
            }, 2);
            return !differ.load();
        });
    }
    rc = to_device(h, 0, b, d, l.b);
    if (compared.valid()) same = compared.get();          // (joined before any return: the task reads the caller's arrays)
    if (rc) return rc;
    h->timing["load_b"] = ms_since(tl); tl = clk::now();
    if (same) HIPCHK(hipMemcpyAsync(l.x, l.b, sizeof(double) * (size_t)l.n_pad * d, hipMemcpyDeviceToDevice, h->stream));
    else if ((rc = to_device(h, 0, x0, d, l.x))) return rc;
    h->timing["load_x"] = ms_since(tl); tl = clk::now();
    if (h->cfg.inner_precision && (rc = launch_residual_to_f32(h, d, -1))) return rc;    // defect of the initial guess -> b32
    HIPCHK(hipStreamSynchronize(h->stream));
    h->timing["load_sync"] = ms_since(tl);
    h->loaded_d = d;
    return GMG_OK;
} GMG_CATCH_H

namespace {
// arms the device-side decision of the residual checks enqueued inside its scope (launch_reduce) and clears it, with a head that was not
// taken up, on every way out
struct WatchScope {
    gmg_handle h; bool on;
    WatchScope(gmg_handle h_, bool on_, int mode, double tol, int type) : h(h_), on(on_) {
        h->head_enqueued = false;
        if (!on) return;
        h->timing["heads_enqueued"] += 0.0; h->timing["head_decision_differs"] += 0.0;
        h->watch_active = true; h->watch_mode = mode; h->watch_tol = tol; h->watch_type = type; h->watch_cycles_done = 0;
    }
    ~WatchScope() { h->watch_active = false; h->head_enqueued = false; }
};
}  // namespace

int gmg_run_cycles(gmg_handle h, int n_cycles, int stop_type, double* residues) try {
    NEED_DEVICE();
    if (h->loaded_d <= 0) return fail(h, GMG_ERR_STATE, "no problem loaded (gmg_load_problem)");
    if (n_cycles < 0) return fail(h, GMG_ERR_INVALID, "bad cycle count");
    int rc;
    if ((rc = check_whole_system(h))) return rc;
    if (stop_type >= 0 && (rc = check_norm_type(h, stop_type))) return rc;
    const int d = h->loaded_d;
    HelperScope helper_scope(h, d);
    // (a fixed number of cycles: the first colour launch of the next one goes into the stream before the host has seen this one's norm,
    // as in the solve loop -- there the check's reduction decides on the device whether that launch does anything, solve_common)
    WatchScope watch(h, stop_type >= 0 && head_eligible(h, d), 0, 0.0, stop_type);
    for (int i = 0; i < n_cycles; ++i) {
        h->watch_cycles_done = i + 1;
        if ((rc = vcycle_resident(h, d, stop_type))) return rc;
        if (stop_type >= 0) {
            if (watch.on && i + 1 < n_cycles) enqueue_head(h, d);
            if ((rc = wait_norm(h))) return rc;
            if (residues) residues[i] = norm_from_sums(h->h_norm, d, stop_type);
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return GMG_OK;
} GMG_CATCH_H

// Where the time of a cycle goes, leg by leg: `reps` V-cycles + residual checks on the resident problem with an event at every leg boundary.
// ms_out[k], k < L: level k (way down: pre-smoothing, residual, restriction; way up: prolongation, post-smoothing); ms_out[L]: the coarsest
// solve (publication of the right-hand side, host back-substitution, fetch: GPU idle time included); ms_out[L + 1]: the residual check.
// The events cost a little (the sum is reported against the unprofiled cycle by the caller).
int gmg_profile_cycle(gmg_handle h, int stop_type, int reps, double* ms_out, int n_out) try {
    NEED_DEVICE();
    if (h->loaded_d <= 0) return fail(h, GMG_ERR_STATE, "no problem loaded (gmg_load_problem)");
    const int L = h->L;
    if (!ms_out || n_out < L + 2 || reps <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments (ms_out needs levels + 2 entries)");
    if (h->cfg.use_graph) return fail(h, GMG_ERR_UNSUPPORTED, "leg profiling needs stream launches (use_graph = 0)");
    int rc;
    if ((rc = check_whole_system(h))) return rc;
    if ((rc = check_norm_type(h, stop_type))) return rc;
    const int d = h->loaded_d;
    HelperScope helper_scope(h, d);
    std::vector<double> acc((size_t)L + 2, 0.0);
    for (int i = 0; i < reps; ++i) {
        h->prof_on = true; h->prof_n = 0;
        rc = vcycle_resident(h, d, stop_type);
        h->prof_on = false;
        if (rc) return rc;
        if ((rc = wait_norm(h))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->prof_n != 2 * L + 3) return fail(h, GMG_ERR_STATE, "unexpected number of leg boundaries");
        auto span = [&](int a, int b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, h->prof_ev[a], h->prof_ev[b]); return (double)ms; };
        for (int k = 0; k < L; ++k) acc[k] += span(k, k + 1) + span(L + 1 + (L - 1 - k), L + 2 + (L - 1 - k));
        acc[L] += span(L, L + 1);
        acc[L + 1] += span(2 * L + 1, 2 * L + 2);
    }
    for (int k = 0; k < L + 2; ++k) ms_out[k] = acc[k] / reps;
    return GMG_OK;
} GMG_CATCH_H

int gmg_fetch_solution(gmg_handle h, double* x) try {
    NEED_DEVICE();
    if (h->loaded_d <= 0) return fail(h, GMG_ERR_STATE, "no problem loaded (gmg_load_problem)");
    if (!x) return fail(h, GMG_ERR_INVALID, "bad arguments");
    return to_host(h, 0, h->lv[0].x, h->loaded_d, x);
} GMG_CATCH_H

int gmg_vcycle(gmg_handle h, const double* b, double* x, int d) try {
    int rc = gmg_load_problem(h, b, x, d);
    if (rc) return rc;
    if ((rc = gmg_run_cycles(h, 1, -1, nullptr))) return rc;
    return gmg_fetch_solution(h, x);
} GMG_CATCH_H

// x0: the initial guess (may be rhs itself: then it is copied on the device, not uploaded); x: receives the last iterate
static int solve_common(gmg_handle h, const double* rhs, const double* x0, double* x, int d, double tol, int stop_type, int max_iter, int* iters_out,
                        double* residue_out, double* conv) {
    NEED_DEVICE();
    int rc;
    if (!rhs || !x0 || !x) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = check_whole_system(h))) return rc;
    if ((rc = check_norm_type(h, stop_type))) return rc;
    if (max_iter < 1) max_iter = 1;      // do { } while: at least one cycle (multigrid_solver.cpp:1411-1417)
    auto t_all = clk::now();
    HelperScope helper_scope(h, d);
    if ((rc = gmg_load_problem(h, rhs, x0, d))) return rc;
    h->timing["solve_load"] = ms_since(t_all);
    h->timing["coarse_host_ms"] = 0.0;
    auto t0 = clk::now();
    double residue = 0.0, first_residue = 0.0, least_residue = 0.0;
    int it = 0;
    bool blown = false, go_on = false;
    // Head of the next cycle (gmg_config::speculate_head): the check's reduction takes the decision below on the device too (same sums, same
    // correctly rounded arithmetic: gmgk::reduce_partials / SolveWatch) and the first colour launch of the next cycle is enqueued behind it at
    // once -- it returns without touching x when the iteration has stopped.  The ~6 us the host needs to see the norm and to get a launch to
    // the device are hidden behind that launch.  The host follows the device's word (one decision, not two).
    WatchScope watch(h, head_eligible(h, d), 1, tol, stop_type);
    do {
        h->watch_cycles_done = it + 1;
        if ((rc = vcycle_resident(h, d, stop_type))) return rc;
        const bool head = watch.on && it + 2 <= max_iter;          // (a cycle after this one is allowed)
        if (head) enqueue_head(h, d);
        if ((rc = wait_norm(h))) return rc;
        residue = norm_from_sums(h->h_norm, d, stop_type);
        if (it == 0) first_residue = least_residue = residue;
        if (residue < least_residue) least_residue = residue;
        if (conv) { conv[2 * it] = ms_since(t0); conv[2 * it + 1] = residue; }
        ++it;
        if (h->cfg.verbose) std::printf("%d,%f,%.14f \n", it, ms_since(t0), residue);
        // no way back from here (the reference would spin to max_iter on NaNs): stop, the caller is told below
        blown = !std::isfinite(residue) || (it >= 3 && residue > 1e4 * least_residue);
        go_on = residue > tol && it < max_iter && !blown;
        if (head) {
            const bool device_go = __atomic_load_n(h->h_flag + 1, __ATOMIC_ACQUIRE) != 0;
            if (device_go != go_on) h->timing["head_decision_differs"] += 1.0;      // (never seen: the two sides compute the same bits)
            go_on = device_go;
            if (!go_on) h->head_enqueued = false;                  // that launch found the word cleared and returned
        }
    } while (go_on);
    h->timing["cycles"] = ms_since(t0);
    // Not contracting: the iteration ended above the tolerance with a residue that is not finite or larger than after the first cycle.
    // The parallel smoothers are not the reference's lexicographic Gauss-Seidel (block sweeps on the Galerkin levels take the
    // couplings between blocks from the previous sweep; nothing guarantees their convergence for every SPD matrix), so the caller
    // is told -- return value GMG_DIVERGED and timing key "diverged" -- and can retry on a handle with block_rows = 0, gs_omega = 1:
    // Gauss-Seidel in colour order on every level, convergent for every SPD matrix (MultigridSolver::solve does).  x receives the
    // last iterate either way, as in the reference (multigrid_solver.cpp:1408-1419 never looks at the trend).
    bool diverged = !(residue <= tol) && (blown || (it > 1 && residue > first_residue));
    h->timing["diverged"] = diverged ? 1.0 : 0.0;
    h->timing["blown_up"] = blown ? 1.0 : 0.0;           // stopped early: residue not finite or 1e4 x the smallest seen
    auto t_f = clk::now();
    if ((rc = gmg_fetch_solution(h, x))) return rc;
    h->timing["solve_fetch"] = ms_since(t_f);
    h->timing["iterations"] = it;
    h->timing["residue"] = residue;
    h->timing["solve_call"] = ms_since(t_all);
    h->timing["solver_total"] = h->timing["setup_total"] + h->timing["solve_call"];
    if (iters_out) *iters_out = it;
    if (residue_out) *residue_out = residue;
    return diverged ? GMG_DIVERGED : GMG_OK;
}

int gmg_solve(gmg_handle h, const double* rhs, double* x, int d, double tol, int stop_type, int max_iter, int* iters_out,
              double* residue_out, double* conv) try {
    return solve_common(h, rhs, x, x, d, tol, stop_type, max_iter, iters_out, residue_out, conv);
} GMG_CATCH_H

// The reference's binding always starts from x0 = rhs (gravomg_bindings/src/cpp/core.cpp:69): this entry point says so, and the
// caller neither fills x with a copy of rhs nor pays for the comparison gmg_solve makes to find that out.  x is output only.
int gmg_solve_x0_rhs(gmg_handle h, const double* rhs, double* x, int d, double tol, int stop_type, int max_iter, int* iters_out,
                     double* residue_out, double* conv) try {
    return solve_common(h, rhs, rhs, x, d, tol, stop_type, max_iter, iters_out, residue_out, conv);
} GMG_CATCH_H

// ---- multi-GPU: one process per GPU, level 0 row-partitioned per colour, levels >= 1 replicated ---------------
// The caller (gravo_mg_amd/dist.py) owns the level-0 vectors and performs the exchanges (RCCL all-gather of the
// colour segment of x after every colour); these entry points only launch this rank's share of the work.

int gmg_set_stream(gmg_handle h, void* hip_stream) try {
    NEED_DEVICE();
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graphs(h);
    h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
    return GMG_OK;
} GMG_CATCH_H

int gmg_dist_partition(gmg_handle h, int rank, int world) try {
    if (!h) return GMG_ERR_INVALID;
    PoolScope pool_scope_(&h->pool);
    if (world < 1 || rank < 0 || rank >= world) return fail(h, GMG_ERR_INVALID, "bad rank / world size");
    if (world > 1 && h->cfg.row_align % (64 * world)) return fail(h, GMG_ERR_STATE, "create the handle with row_align = 64 * world (colour classes are cut into `world` pieces of whole slices)");
    if (rank == h->part_rank && world == h->part_world) return GMG_OK;
    if (h->has_device && (h->system_ready || h->placeholder_ready)) { drop_system(h); h->live_key_valid = false; }      // laid out for another partition
    h->part_rank = rank; h->part_world = world;
    h->plan.reset();
    return GMG_OK;
} GMG_CATCH_H

int gmg_dist_setup(gmg_handle h, int rank, int world) try {
    NEED_DEVICE();
    int rc = check_level(h, 0, false);
    if (rc) return rc;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, GMG_ERR_INVALID, "bad rank / world size");
    if (h->partitioned && (rank != h->part_rank || world != h->part_world)) return fail(h, GMG_ERR_STATE, "the system was laid out for another rank / world size (gmg_dist_partition)");
    const LevelOrdering& o = h->lv[0].ord;
    if (h->cfg.smoother != GMG_SMOOTHER_MULTICOLOR_GS) return fail(h, GMG_ERR_STATE, "the distributed path needs the multicolour / block-hybrid smoothers (gmg_config::smoother)");
    // a blocked level 0 (gmg_config::block_fine: kNN operators) is ONE class of rows cut into `world` runs of whole 64-row blocks
    if (o.blocked && (h->cfg.block_rows != 64 || !h->lv[0].use_ep)) return fail(h, GMG_ERR_STATE, "a blocked level 0 is partitioned only as 64-row blocks of the entry-parallel sweep (block_rows = 64, block_ep = 1, one lane per row): set block_fine = 0 or block_lanes = 1");
    for (int c = 0; c < dist_classes(o); ++c)
        if ((o.color_begin[c + 1] - o.color_begin[c]) % (64 * world)) return fail(h, GMG_ERR_STATE, "colour classes are not aligned to 64*world rows: create the handle with row_align = 64*world");
    h->rank = rank; h->world = world; h->dist_ready = true;
    return GMG_OK;
} GMG_CATCH_H

int gmg_dist_bind(gmg_handle h, double* x0, double* b0, double* r0, int d) try {
    NEED_DEVICE();
    if (!h->dist_ready) return fail(h, GMG_ERR_STATE, "call gmg_dist_setup first");
    if (!x0 || !b0 || !r0 || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    int rc = ensure_vectors(h, d);
    if (rc) return rc;
    drop_graphs(h);
    Level& l = h->lv[0];
    if (!h->bound) { h->own_x0 = l.x; h->own_b0 = l.b; h->own_r0 = l.r; }
    l.x = x0; l.b = b0; l.r = r0;
    h->bound = true;
    h->loaded_d = d;
    return GMG_OK;
} GMG_CATCH_H

namespace {
inline void own_range(gmg_handle h, int c, int& sb, int& se) {
    const LevelOrdering& o = h->lv[0].ord;
    if (h->dist_all_rows) { sb = o.color_begin[c] / 64; se = o.color_begin[c + 1] / 64; return; }     // the *_all entry points
    const int chunk = (o.color_begin[c + 1] - o.color_begin[c]) / 64 / h->world;
    sb = o.color_begin[c] / 64 + h->rank * chunk;
    se = sb + chunk;
}
int dist_ready(gmg_handle h) {
    if (!h->dist_ready || !h->bound || h->loaded_d <= 0) return fail(h, GMG_ERR_STATE, "distributed state not set (gmg_dist_setup + gmg_dist_bind)");
    return GMG_OK;
}
}  // namespace

// One colour of one Gauss-Seidel sweep on this rank's rows of level 0.  plain_rows (hybrid smoother, engine_dist.hip.hpp::p2p_smooth): one 64-bit word per
// slice, bit set = the row takes omega = 1.
static int dist_smooth_color_impl(gmg_handle h, int c, const unsigned long long* plain_rows) {
    NEED_DEVICE();
    int rc = dist_ready(h);
    if (rc) return rc;
    Level& l = h->lv[0];
    if (c < 0 || c >= dist_classes(l.ord)) return fail(h, GMG_ERR_INVALID, "colour out of range");
    if (l.ord.blocked) return fail(h, GMG_ERR_STATE, "level 0 is blocked: its sweep is a block sweep (gmg_p2p_cycles), not a colour sweep");
    int sb, se;
    own_range(h, c, sb, se);
    const int d = h->loaded_d, ld = l.n_pad;
    if (se > sb)
        for (int c0 = 0; c0 < d; c0 += 4) {
            int dc = std::min(4, d - c0);
            if (l.Aoff.c16_mode != 0 && plain_rows) {
                DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::gs_color<double, D, C16 + 1, 1>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream, l.Aoff.slice_ptr,
                                                  l.Aoff.col, l.Aoff.val, l.diag, l.b + (size_t)c0 * ld, l.x + (size_t)c0 * ld, ld, sb, se, 1, h->cfg.gs_omega,
                                                  l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg(), plain_rows)));
            } else if (plain_rows) {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_color<double, D, 1, 1>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream, l.Aoff.slice_ptr,
                                                  l.Aoff.col, l.Aoff.val, l.diag, l.b + (size_t)c0 * ld, l.x + (size_t)c0 * ld, ld, sb, se, 1, h->cfg.gs_omega,
                                                  (const unsigned*)nullptr, (const int*)nullptr, 0, plain_rows));
            } else if (l.Aoff.c16_mode != 0) {       // (c16_sel: a rank whose share of the fine operators fits its memory-side cache reads them with ordinary loads)
                DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::gs_color<double, D, C16 + 1>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream, l.Aoff.slice_ptr,
                                                  l.Aoff.col, l.Aoff.val, l.diag, l.b + (size_t)c0 * ld, l.x + (size_t)c0 * ld, ld, sb, se, 1, h->cfg.gs_omega,
                                                  l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
            } else {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::gs_color<double, D, 1>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream, l.Aoff.slice_ptr,
                                                  l.Aoff.col, l.Aoff.val, l.diag, l.b + (size_t)c0 * ld, l.x + (size_t)c0 * ld, ld, sb, se, 1, h->cfg.gs_omega));
            }
        }
    return GMG_OK;
}
int gmg_dist_smooth_color(gmg_handle h, int c) try { return dist_smooth_color_impl(h, c, nullptr); } GMG_CATCH_H

// r0[own rows] = b0 - A x0
int gmg_dist_residual_own(gmg_handle h) try {
    NEED_DEVICE();
    int rc = dist_ready(h);
    if (rc) return rc;
    Level& l = h->lv[0];
    const int d = h->loaded_d, ld = l.n_pad;
    for (int c = 0; c < dist_classes(l.ord); ++c) {
        int sb, se;
        own_range(h, c, sb, se);
        if (se <= sb) continue;
        for (int c0 = 0; c0 < d; c0 += 4) {
            int dc = std::min(4, d - c0);
            DISPATCH_D(dc, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::spmv_full<double, D, 1, 1, C16>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0,
                                              h->stream, l.Aoff.slice_ptr, l.Aoff.col, l.Aoff.val, l.diag, l.b + (size_t)c0 * ld, l.x + (size_t)c0 * ld,
                                              l.r + (size_t)c0 * ld, ld, sb, se, 1, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg())));
        }
    }
    return GMG_OK;
} GMG_CATCH_H

// Replicated coarse part: b1 = U0^T r0 (needs the complete r0), levels 1..L-1 down, coarsest solve, back up to level 1.
// _enqueue leaves the host half of the coarsest solve pending (coarse_host_serve) so that the caller can queue more work first.
static int dist_coarse_cycle_enqueue(gmg_handle h) {
    int rc = dist_ready(h);
    if (rc) return rc;
    const int d = h->loaded_d;
    h->first_sweep_fused = false;
    restrict_into<double>(h, 0, d, false);          // (+ level 1's first sweep where the layouts allow it)
    enqueue_down<double>(h, d, 1);
    if (h->coarse_device) enqueue_coarse_device<double>(h, d);
    else if ((rc = coarse_host_begin<double>(h, d))) return rc;
    enqueue_up<double>(h, d, 1);
    return GMG_OK;
}
int gmg_dist_coarse_cycle(gmg_handle h) try {
    NEED_DEVICE();
    int rc = dist_coarse_cycle_enqueue(h);
    const int served = coarse_host_serve(h);
    return rc ? rc : served;
} GMG_CATCH_H

// x0[own rows] += U0 x1
int gmg_dist_prolong_own(gmg_handle h) try {
    NEED_DEVICE();
    int rc = dist_ready(h);
    if (rc) return rc;
    Level& l = h->lv[0];
    Level& cl = h->lv[1];
    const int d = h->loaded_d;
    for (int c = 0; c < dist_classes(l.ord); ++c) {
        int sb, se;
        own_range(h, c, sb, se);
        if (se <= sb) continue;
        for (int c0 = 0; c0 < d; c0 += 4) {
            int dc = std::min(4, d - c0);
            DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::transfer<double, D, 1, 1>), dim3(grid_for(se - sb)), dim3(gmgk::kBlock), 0, h->stream, l.P.slice_ptr, l.P.col,
                                              l.P.val, (const int*)nullptr, cl.x + (size_t)c0 * cl.n_pad, cl.n_pad, l.x + (size_t)c0 * l.n_pad, l.n_pad,
                                              sb, se, 1));
        }
    }
    return GMG_OK;
} GMG_CATCH_H

// this rank's share of sum w r^2 / sum w b^2 per column -> h->d_norm[2 * d] (on the stream; no synchronisation)
static int dist_norm_launch(gmg_handle h, int type) {
    int rc = dist_ready(h);
    if (rc) return rc;
    if ((rc = check_norm_type(h, type))) return rc;
    Level& l = h->lv[0];
    const int d = h->loaded_d;
    const double* w = type == 1 ? h->d_minv : (type == 2 ? h->d_mass : nullptr);
    const int nc = dist_classes(l.ord);
    const int nblk = std::max(1, kNormBlocks / std::max(nc, 1));
    for (int c0 = 0; c0 < d; c0 += 4) {
        int dc = std::min(4, d - c0);
        for (int c = 0; c < nc; ++c) {
            int sb, se;
            own_range(h, c, sb, se);
            DISPATCH_D(dc, hipLaunchKernelGGL(gmgk::residual_norm_partials<D>, dim3(nblk), dim3(gmgk::kBlock), 0, h->stream, l.Aoff.slice_ptr, l.Aoff.col,
                                              l.Aoff.val, l.diag, l.b + (size_t)c0 * l.n_pad, l.x + (size_t)c0 * l.n_pad, w, l.n_pad, sb, se,
                                              h->d_partials + (size_t)c * nblk * 2 * dc));
        }
        hipLaunchKernelGGL(gmgk::reduce_partials, dim3(1), dim3(gmgk::kReduceBlock), 0, h->stream, h->d_partials, nblk * nc, 2 * dc, h->d_norm + 2 * c0,
                           (unsigned long long*)nullptr, 0ull, 0);
    }
    return GMG_OK;
}

// sums[2*c] / sums[2*c+1] = this rank's share of sum w r^2 / sum w b^2 for column c (host output; synchronises).
int gmg_dist_norm_partial(gmg_handle h, int type, double* sums) try {
    NEED_DEVICE();
    if (!sums) return fail(h, GMG_ERR_INVALID, "bad arguments");
    int rc = dist_norm_launch(h, type);
    if (rc) return rc;
    const int d = h->loaded_d;
    HIPCHK(hipMemcpyAsync(h->h_norm, h->d_norm, sizeof(double) * 2 * d, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::memcpy(sums, h->h_norm, sizeof(double) * 2 * d);
    return GMG_OK;
} GMG_CATCH_H

// The same steps over ALL rows of level 0: after the exchange that follows every colour sweep each rank holds the complete x, so residual,
// prolongation-add and the norm sums can be computed redundantly instead of being exchanged (16 collectives per V-cycle, one per colour
// sweep, instead of 24 + an all-reduce; the sums are then identical on all ranks).  on != 0: gmg_dist_residual_own / gmg_dist_prolong_own /
// gmg_dist_norm_partial cover every row until it is switched off again.
int gmg_dist_all_rows(gmg_handle h, int on) try {
    if (!h) return GMG_ERR_INVALID;
    if (on && h->partitioned) return fail(h, GMG_ERR_STATE, "this handle holds one rank's rows only (gmg_dist_partition)");
    h->dist_all_rows = on != 0;
    return GMG_OK;
} GMG_CATCH_H

int gmg_dist_gather(gmg_handle h, const double* src, const int64_t* idx, int64_t n, double* dst) try {
    NEED_DEVICE();
    if (n < 0 || (n > 0 && (!src || !idx || !dst))) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if (n) hipLaunchKernelGGL(gmgk::gather_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, idx, n, dst);
    return GMG_OK;
} GMG_CATCH_H
int gmg_dist_scatter(gmg_handle h, const double* src, const int64_t* pos, const int64_t* idx, int64_t n, double* dst) try {
    NEED_DEVICE();
    if (n < 0 || (n > 0 && (!src || !pos || !idx || !dst))) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if (n) hipLaunchKernelGGL(gmgk::scatter_entries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, src, pos, idx, n, dst);
    return GMG_OK;
} GMG_CATCH_H

// ---- measurement --------------------------------------------------------------------------------------------

int gmg_algorithmic_bytes(gmg_handle h, int kind, int k, int d, double* bytes_out) try {
    if (!h || !bytes_out) return GMG_ERR_INVALID;
    int rc = check_level(h, k, false);
    if (rc) return rc;
    const Level& l = h->lv[k];
    const double s = 8.0, n = l.n, z = (double)l.nnz, u = (double)h->U[k].nnz(), nc = h->lv[k + 1].n;
    // matrix stream (value + int32 index) + row pointer + the dense vectors, each touched once
    const double sweep = z * (s + 4) + 4 * (n + 1) + 3 * n * d * s;
    switch (kind) {
        case 0: case 1: *bytes_out = sweep; break;
        case 2: *bytes_out = u * (s + 4) + 4 * (nc + 1) + n * d * s + nc * d * s; break;
        case 3: *bytes_out = u * (s + 4) + 4 * (n + 1) + nc * d * s + 2 * n * d * s; break;
        case 4: *bytes_out = sweep - n * d * s + n * s; break;       // reads x, b, M; writes nothing
        default: return fail(h, GMG_ERR_INVALID, "unknown kernel kind");
    }
    return GMG_OK;
} GMG_CATCH_H

int gmg_bench_kernel(gmg_handle h, int kind, int k, int d, int reps, double* ms_avg, int* launches_out) try {
    NEED_DEVICE();
    int rc = check_level(h, k, false);
    if (rc) return rc;
    if ((rc = check_whole_system(h))) return rc;
    if (!ms_avg || reps <= 0 || d <= 0) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if ((rc = ensure_vectors(h, d))) return rc;
    Level& l = h->lv[k];
    int launches = 1;
    const bool il = k == 0 && d > 1 && d <= 4 && l.Aoff.lpr == 1;
    const bool il_p = il && h->L >= 2 && h->lv[1].ord.blocked && h->lv[1].use_ep && h->cfg.post_iters > 0 && h->cfg.smoother != GMG_SMOOTHER_JACOBI;
    auto body = [&]() {
        switch (kind) {
            case 0: launch_smooth<double>(h, l, d, 2); launches = (h->cfg.smoother == GMG_SMOOTHER_JACOBI || l.ord.blocked) ? 1 : l.ord.n_colors; break;
            // (level 0 with 2 .. 4 right-hand sides: the variants the cycle runs -- residual written / gathered as an interleaved multi-vector,
            // prolongation from the interleaved copy of level 1's x: engine_cycle.hip.hpp::enqueue_down / enqueue_up)
            case 1: launch_spmv<double>(h, l, d, 1, l.b, l.x, l.r, -1, il); break;
            case 2: launch_restrict<double>(h, l, h->lv[k + 1], d, l.r, h->lv[k + 1].b, il); break;
            case 3: launch_prolong_add<double>(h, l, h->lv[k + 1], d, il_p ? h->lv[k + 1].r : h->lv[k + 1].x, l.x, il_p); break;
            case 4: (void)launch_norm(h, d, 0); launches = 2; break;
            default: break;
        }
    };
    if (kind < 0 || kind > 4) return fail(h, GMG_ERR_INVALID, "unknown kernel kind");
    if (kind == 4 && k != 0) return fail(h, GMG_ERR_INVALID, "the norm kernel runs on level 0");
    for (int i = 0; i < 3; ++i) body();
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) body();
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_avg = (double)ms / reps / (kind == 0 ? 2 : 1);     // kind 0 enqueues two sweeps per repetition (ping-pong buffers)
    if (launches_out) *launches_out = launches;
    h->loaded_d = 0;
    return GMG_OK;
} GMG_CATCH_H

// ---- host-only: hierarchy -----------------------------------------------------------------------------------

int gmg_hierarchy_options_default(gmg_hierarchy_options* o) try {
    if (!o) return GMG_ERR_INVALID;
    o->ratio = 8.0; o->lower_bound = 1000; o->check_voronoi = 1; o->nested = 0; o->sampling = 0; o->weighting = 0; o->debug = 0; o->full_clustering = 0; o->use_device = 1;
    return GMG_OK;
} GMG_CATCH_0

// Device stage of the hierarchy builder (HierarchyOptions::device_select): uploads one level's selection inputs, runs
// gmgh::select_parents, downloads the per-point records.  A hierarchy is built before any handle exists, so the stage keeps its
// own stream and two pinned bounce buffers (process-wide, created on first use, one build at a time): the big arrays -- positions
// in, 38 bytes per point out -- cross PCIe in 16 MB pieces with worker threads copying the neighbouring piece between pageable
// memory and the bounce buffer (threaded first touch of the 110 MB result included).  Returns false (the host loop does the level)
// on any HIP failure.
namespace {
struct HierarchyXfer {
    std::mutex m;
    hipStream_t st = nullptr;
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ok = false, tried = false;
    int device = -1;                  // the stream's device: the calling thread's current device at first use
    // dev: the device the caller's allocations and launches go to; a process that later builds on another device keeps the host loop
    bool ready(int dev) {
        if (tried) return ok && dev == device;
        tried = true;
        device = dev;
        ok = hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i)
            ok = hipHostMalloc(&buf[i], kBounceBytes, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }
    bool up(void* dst, const void* src, size_t bytes, int threads) {
        int f = 0;
        for (size_t off = 0; off < bytes; off += kBounceBytes, f ^= 1) {
            const size_t len = std::min(kBounceBytes, bytes - off);
            if (hipEventSynchronize(ev[f]) != hipSuccess) return false;
            threaded_copy_bytes(buf[f], (const char*)src + off, len, threads);
            if (hipMemcpyAsync((char*)dst + off, buf[f], len, hipMemcpyHostToDevice, st) != hipSuccess || hipEventRecord(ev[f], st) != hipSuccess) return false;
        }
        return true;
    }
    bool down(void* dst, const void* src, size_t bytes, int threads) {
        const size_t nchunk = (bytes + kBounceBytes - 1) / kBounceBytes;
        auto issue = [&](size_t c) {
            const int f = (int)(c & 1);
            const size_t off = c * kBounceBytes, len = std::min(kBounceBytes, bytes - off);
            return hipMemcpyAsync(buf[f], (const char*)src + off, len, hipMemcpyDeviceToHost, st) == hipSuccess && hipEventRecord(ev[f], st) == hipSuccess;
        };
        if (hipEventSynchronize(ev[0]) != hipSuccess || hipEventSynchronize(ev[1]) != hipSuccess) return false;
        if (nchunk && !issue(0)) return false;
        for (size_t c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk && !issue(c + 1)) return false;
            const int f = (int)(c & 1);
            const size_t off = c * kBounceBytes, len = std::min(kBounceBytes, bytes - off);
            if (hipEventSynchronize(ev[f]) != hipSuccess) return false;
            threaded_copy_bytes((char*)dst + off, buf[f], len, threads);
        }
        return true;
    }
};
HierarchyXfer& hierarchy_xfer() { static HierarchyXfer* x = new HierarchyXfer(); return *x; }      // (leaked: no destruction order problems at exit)
}  // namespace

static bool hierarchy_select_on_device(const HierarchyOptions::SelectJob& j) {
    HierarchyXfer& X = hierarchy_xfer();
    std::lock_guard<std::mutex> lock(X.m);
    // (the builder runs this stage on a task thread: the device to use travels in the job)
    int dev = j.device;
    if ((dev < 0 && hipGetDevice(&dev) != hipSuccess) || hipSetDevice(dev) != hipSuccess || !X.ready(dev)) { (void)hipGetLastError(); return false; }
    const int threads = std::min(hw_threads(), 16);
    // one device allocation for the whole job, carved into 256-byte aligned pieces
    const size_t nf = (size_t)j.nf, nc = (size_t)j.nc;
    const size_t sizes[14] = {sizeof(double) * 3 * nf, sizeof(double) * 3 * nc, sizeof(int) * nf, j.nested ? sizeof(int) * nc : 0, sizeof(int) * (nc + 1),
                              sizeof(int) * (size_t)j.cadj_ptr[nc], sizeof(int) * 3 * (size_t)j.ntri, sizeof(int) * (nc + 1), sizeof(int) * (size_t)j.tof_ptr[nc],
                              sizeof(int) * nc * (size_t)j.Kc, nf, nf, sizeof(int) * 3 * nf, sizeof(double) * 3 * nf};
    size_t offs[15] = {0};
    for (int i = 0; i < 14; ++i) offs[i + 1] = offs[i] + (sizes[i] + 255) / 256 * 256;
    char* arena = nullptr;
    if (hipMalloc((void**)&arena, std::max<size_t>(offs[14], 256)) != hipSuccess) { (void)hipGetLastError(); return false; }
    const void* srcs[10] = {j.P, j.Pc, j.nearest, j.sample, j.cadj_ptr, j.cadj, j.tris, j.tof_ptr, j.tof, j.NBc};
    bool ok = true;
    for (int i = 0; i < 10 && ok; ++i) ok = sizes[i] == 0 || X.up(arena + offs[i], srcs[i], sizes[i], threads);
    if (ok) {
        hipLaunchKernelGGL(gmgh::select_parents, dim3((unsigned)((nf + 127) / 128)), dim3(128), 0, X.st, j.nf, j.Kc, j.weighting, j.nested,
                           (const double*)(arena + offs[0]), (const double*)(arena + offs[1]), (const int*)(arena + offs[2]), (const int*)(arena + offs[3]),
                           (const int*)(arena + offs[4]), (const int*)(arena + offs[5]), (const int*)(arena + offs[6]), (const int*)(arena + offs[7]),
                           (const int*)(arena + offs[8]), (const int*)(arena + offs[9]), (unsigned char*)(arena + offs[10]), (unsigned char*)(arena + offs[11]),
                           (int*)(arena + offs[12]), (double*)(arena + offs[13]));
        ok = hipGetLastError() == hipSuccess && X.down(j.cnt, arena + offs[10], nf, threads) && X.down(j.kind, arena + offs[11], nf, threads) &&
             X.down(j.col, arena + offs[12], sizeof(int) * 3 * nf, threads) && X.down(j.w, arena + offs[13], sizeof(double) * 3 * nf, threads);
    }
    (void)hipStreamSynchronize(X.st);          // nothing of this job may still be in flight when its buffers go
    (void)sync_hipFree(arena);
    if (!ok) (void)hipGetLastError();
    return ok;
}

int gmg_hierarchy_build(const double* pos, int n, const int* neigh, int K, const gmg_hierarchy_options* opt, gmg_hierarchy* out) try {
    if (!pos || !neigh || n <= 0 || K <= 0 || !out) return GMG_ERR_INVALID;
    gmg_hierarchy_options o;
    if (opt) o = *opt; else gmg_hierarchy_options_default(&o);
    if (o.sampling != 0) return GMG_ERR_UNSUPPORTED;      // only Sampling::FASTDISK (the default) is in scope
    if (o.weighting < 0 || o.weighting > 2 || !(o.ratio > 0)) return GMG_ERR_INVALID;
    for (size_t i = 0; i < (size_t)n * K; ++i) if (neigh[i] >= n) return GMG_ERR_INVALID;
    HierarchyOptions ho;
    ho.ratio = o.ratio; ho.lower_bound = o.lower_bound; ho.check_voronoi = o.check_voronoi != 0; ho.nested = o.nested != 0; ho.weighting = o.weighting; ho.keep_triangles = o.debug != 0; ho.full_clustering = o.full_clustering != 0;
    // the per-point selection stage runs on the GPU when there is one (same bits as the host loop; gmg_hierarchy_options::use_device = 0: host only)
    {
        int ndev = 0;
        if (o.use_device != 0 && n >= ho.device_select_min_points && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) ho.device_select = hierarchy_select_on_device;
        else (void)hipGetLastError();
    }
    // first use in a process: runtime start-up, code object load and the pinned buffers (~80 ms) happen beside the sequential
    // sampling / clustering sweeps of the first level instead of in front of the device stage
    std::future<void> device_warm;
    int caller_device = 0;
    if (ho.device_select && hipGetDevice(&caller_device) != hipSuccess) { (void)hipGetLastError(); ho.device_select = nullptr; }
    ho.device = caller_device;
    if (ho.device_select)
        device_warm = std::async(std::launch::async, [caller_device] {
            HierarchyXfer& X = hierarchy_xfer();
            std::lock_guard<std::mutex> lock(X.m);
            if (hipSetDevice(caller_device) != hipSuccess || !X.ready(caller_device)) { (void)hipGetLastError(); return; }
            hipLaunchKernelGGL(gmgh::select_parents, dim3(1), dim3(128), 0, X.st, 0, 0, 0, 0, (const double*)nullptr, (const double*)nullptr, (const int*)nullptr,
                               (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr,
                               (const int*)nullptr, (const int*)nullptr, (unsigned char*)nullptr, (unsigned char*)nullptr, (int*)nullptr, (double*)nullptr);
            (void)hipStreamSynchronize(X.st);
            (void)hipGetLastError();
        });
    gmg_hierarchy hh = new gmg_hierarchy_s();
    // A breadth-first order of the points over `neigh`, for inputs whose numbering has no locality (randomly ordered scans, point
    // clouds): one of the two base orders gmg_set_system may give the finest level (choose_base_order).  A sequential sweep of the
    // whole graph (~60 ms at 3 M points) on its own thread, beside the construction.
    std::future<std::vector<int>> fine_order;
    if (n > 65536 && mean_index_distance_table(neigh, n, K) > std::max(32768.0, n / 32.0))
        fine_order = std::async(std::launch::async, [neigh, n, K] { return bfs_point_order(neigh, n, K); });
    // ... and the point graph as a canonical sparsity pattern: what gmg_use_hierarchy gives the engine to prepare its structure for
    std::future<std::shared_ptr<const FineGraph>> graph = std::async(std::launch::async, [neigh, n, K] {
        auto g = std::make_shared<FineGraph>();
        g->n = n;
        neigh_pattern(neigh, n, K, g->ptr, g->idx);
        return std::shared_ptr<const FineGraph>(g);
    });
    struct JoinGraph { std::future<std::shared_ptr<const FineGraph>>& f; ~JoinGraph() { if (f.valid()) f.wait(); } } join_graph{graph};      // (reads the caller's table: never outlives the call)
    hh->res = HierarchyBuilder::build(pos, n, neigh, K, ho);
    if (fine_order.valid()) hh->fine_order = fine_order.get();
    hh->graph = graph.get();
    if (device_warm.valid()) device_warm.get();
    *out = hh;
    return GMG_OK;
} GMG_CATCH_0

void gmg_hierarchy_destroy(gmg_hierarchy hh) { delete hh; }

int gmg_hierarchy_num_levels(gmg_hierarchy hh) { return hh ? (int)hh->res.U.size() : GMG_ERR_INVALID; }

int gmg_hierarchy_level_shape(gmg_hierarchy hh, int k, int* n_fine, int* n_coarse, int* nnz) try {
    if (!hh || k < 0 || k >= (int)hh->res.U.size()) return GMG_ERR_INVALID;
    const Compressed& u = hh->res.U[k];
    if (n_fine) *n_fine = u.n_inner;
    if (n_coarse) *n_coarse = u.n_outer;
    if (nnz) *nnz = u.nnz();
    return GMG_OK;
} GMG_CATCH_0

int gmg_hierarchy_get_prolongation(gmg_hierarchy hh, int k, int* colptr, int* rowidx, double* val) try {
    if (!hh || k < 0 || k >= (int)hh->res.U.size()) return GMG_ERR_INVALID;
    const Compressed& u = hh->res.U[k];
    // (threaded: the destinations are usually fresh arrays -- 108 MB of first touches at 3 M vertices)
    if (colptr) threaded_copy_bytes(colptr, u.ptr.data(), sizeof(int) * (size_t)(u.n_outer + 1), hw_threads());
    if (rowidx) threaded_copy_bytes(rowidx, u.idx.data(), sizeof(int) * (size_t)u.nnz(), hw_threads());
    if (val) threaded_copy_bytes(val, u.val.data(), sizeof(double) * (size_t)u.nnz(), hw_threads());
    return GMG_OK;
} GMG_CATCH_0

int gmg_hierarchy_get_timing(gmg_hierarchy hh, const char* key, double* out) try {
    if (!hh || !key || !out) return GMG_ERR_INVALID;
    auto it = hh->res.timing.find(key);
    if (it == hh->res.timing.end()) return GMG_ERR_INVALID;
    *out = it->second;
    return GMG_OK;
} GMG_CATCH_0

int gmg_hierarchy_get_samples(gmg_hierarchy hh, int k, int* out) try {
    if (!hh || !out || k < 0 || k >= (int)hh->res.samples.size()) return GMG_ERR_INVALID;
    std::memcpy(out, hh->res.samples[k].data(), sizeof(int) * hh->res.samples[k].size());
    return GMG_OK;
} GMG_CATCH_0

int gmg_hierarchy_get_nearest(gmg_hierarchy hh, int k, int* out) try {
    if (!hh || !out || k < 0 || k >= (int)hh->res.nearest.size()) return GMG_ERR_INVALID;
    threaded_copy_bytes(out, hh->res.nearest[k].data(), sizeof(int) * hh->res.nearest[k].size(), hw_threads());
    return GMG_OK;
} GMG_CATCH_0

int gmg_hierarchy_get_points(gmg_hierarchy hh, int k, double* out_xyz) try {
    if (!hh || !out_xyz || k < 0 || k >= (int)hh->res.points.size()) return GMG_ERR_INVALID;
    std::memcpy(out_xyz, hh->res.points[k].data(), sizeof(double) * hh->res.points[k].size());
    return GMG_OK;
} GMG_CATCH_0

int gmg_hierarchy_get_triangles(gmg_hierarchy hh, int k, int* out, int* count) try {
    if (!hh || !count || k < 0 || k >= (int)hh->res.U.size()) return GMG_ERR_INVALID;
    if (k >= (int)hh->res.triangles.size()) { *count = 0; return GMG_OK; }
    const auto& t = hh->res.triangles[k];
    *count = (int)t.size();
    if (out && !t.empty()) std::memcpy(out, t.data(), sizeof(int) * 3 * t.size());
    return GMG_OK;
} GMG_CATCH_0

int gmg_hierarchy_get_fine_order(gmg_hierarchy hh, int* out, int* count) try {
    if (!hh || !count) return GMG_ERR_INVALID;
    *count = (int)hh->fine_order.size();
    if (out && !hh->fine_order.empty()) std::memcpy(out, hh->fine_order.data(), sizeof(int) * hh->fine_order.size());
    return GMG_OK;
} GMG_CATCH_0

int gmg_set_fine_order(gmg_handle h, int n, const int* order) try {
    if (!h || n < 0 || (n > 0 && !order)) return GMG_ERR_INVALID;
    if (n > 0) {
        if (h->L <= 0 || !h->U_set[0] || h->U[0].n_inner != n) return fail(h, GMG_ERR_STATE, "set the prolongations first: the order must have one entry per level-0 point");
        std::vector<unsigned char> seen((size_t)n, 0);
        for (int i = 0; i < n; ++i) {
            if (order[i] < 0 || order[i] >= n || seen[order[i]]) return fail(h, GMG_ERR_INVALID, "the fine order is not a permutation of the level-0 points");
            seen[order[i]] = 1;
        }
    }
    h->bfs_order.assign(order, order + n);
    if (h->d_bfs_order) { (void)dev_free(h->d_bfs_order); h->d_bfs_order = nullptr; }
    if (h->d_bfs_inv) { (void)dev_free(h->d_bfs_inv); h->d_bfs_inv = nullptr; }
    h->ord_cache_valid = false;        // cached orderings were built on another base order
    return GMG_OK;
} GMG_CATCH_H

int gmg_set_fine_graph(gmg_handle h, int n, int K, const int* neigh) try {
    if (!h || n < 0 || (n > 0 && (K <= 0 || !neigh))) return GMG_ERR_INVALID;
    if (n == 0) { h->fine_graph.reset(); return GMG_OK; }
    if (h->L <= 0 || !h->U_set[0] || h->U[0].n_inner != n) return fail(h, GMG_ERR_STATE, "set the prolongations first: the graph must have one row per level-0 point");
    {
        std::atomic<bool> bad{false};
        parallel_ranges(n, h->cfg.host_threads, [&](int lo, int hi, int) { for (size_t i = (size_t)lo * K; i < (size_t)hi * K; ++i) if (neigh[i] >= n) { bad = true; return; } }, 1 << 14);
        if (bad) return fail(h, GMG_ERR_INVALID, "neighbour index out of range");
    }
    auto g = std::make_shared<FineGraph>();
    g->n = n;
    neigh_pattern(neigh, n, K, g->ptr, g->idx);
    h->fine_graph = g;
    return GMG_OK;
} GMG_CATCH_H

int gmg_use_hierarchy(gmg_handle h, gmg_hierarchy hh) try {
    if (!h || !hh) return GMG_ERR_INVALID;
    int rc = gmg_set_num_levels(h, (int)hh->res.U.size());
    if (rc) return rc;
    for (int k = 0; k < (int)hh->res.U.size(); ++k) {
        const Compressed& u = hh->res.U[k];
        if ((rc = gmg_set_prolongation(h, k, u.n_inner, u.n_outer, u.ptr.data(), u.idx.data(), u.val.data()))) return rc;
    }
    if (!hh->res.U.empty() && (rc = gmg_set_fine_order(h, (int)hh->fine_order.size(), hh->fine_order.data()))) return rc;
    if (!hh->res.U.empty() && hh->graph && hh->graph->n == hh->res.U[0].n_inner) h->fine_graph = hh->graph;      // (shared, read-only: no copy)
    return gmg_finalize_hierarchy(h);
} GMG_CATCH_H

int gmg_finalize_hierarchy(gmg_handle h) try {
    if (!h) return GMG_ERR_INVALID;
    if (h->L <= 0) return fail(h, GMG_ERR_STATE, "no hierarchy set");
    for (int k = 0; k < h->L; ++k) if (!h->U_set[k]) return fail(h, GMG_ERR_STATE, "prolongation matrix missing for level " + std::to_string(k));
    if (!h->has_device) return GMG_OK;
    PoolScope pool_scope_(&h->pool);      // (everything below allocates and releases through the handle's pool: its byte counts are what gmg_p2p_stat reports)
    // data that belongs to the hierarchy, not to a system (gmg_set_system would make it on its first call otherwise):
    // the compact patches of the blocked levels, the device copies of U_k
    // (the patches are host work on helper threads, the transfers device work driven from this thread: side by side)
    std::future<void> patches;
    if (!h->patches_ready) patches = std::async(std::launch::async, [h] { build_patches(h); });
    int rc = GMG_OK;
    if (h->cfg.device_setup) {
        rc = hipSetDevice(h->cfg.device) == hipSuccess ? ensure_device_transfers(h) : fail(h, GMG_ERR_HIP, "hipSetDevice failed");
    }
    if (patches.valid()) patches.get();
    // with the point graph at hand: everything structural for the systems to come, on placeholder values (prepare_structure)
    if (rc == GMG_OK && h->cfg.prepare_structure && h->cfg.device_setup && h->cfg.device_rap && h->fine_graph && h->fine_graph->n == h->U[0].n_inner && !h->placeholder_ready &&
        !h->system_ready) {
        // (an optional preparation: should it fail -- device memory, a graph the device builders cannot take -- the handle is left as a handle
        // without a system and the first gmg_set_system pays for its structure; the reason stays readable through "structure_prepare_failed")
        const int prc = prepare_structure(h);
        h->timing["structure_prepare_failed"] = prc == GMG_OK ? 0.0 : 1.0;
        if (prc != GMG_OK) { drop_system(h); h->live_key_valid = false; h->system_ready = false; }
        h->fine_graph.reset();      // the digest of the prepared pattern is all that is needed from here on
    }
    return rc;
} GMG_CATCH_H

int gmg_host_galerkin(int n, const int* a_colptr, const int* a_rowidx, const double* a_val, int n_coarse, const int* u_colptr,
                      const int* u_rowidx, const double* u_val, int* c_colptr, int* c_rowidx, double* c_val) try {
    if (n <= 0 || n_coarse <= 0 || !a_colptr || !a_rowidx || !a_val || !u_colptr || !u_rowidx || !u_val || !c_colptr) return GMG_ERR_INVALID;
    Compressed A, U;
    A.assign(n, n, a_colptr, a_rowidx, a_val);
    U.assign(n_coarse, n, u_colptr, u_rowidx, u_val);
    Compressed C = galerkin_rap(A, U, hw_threads());
    std::memcpy(c_colptr, C.ptr.data(), sizeof(int) * (n_coarse + 1));
    if (c_rowidx) std::memcpy(c_rowidx, C.idx.data(), sizeof(int) * C.nnz());
    if (c_val) std::memcpy(c_val, C.val.data(), sizeof(double) * C.nnz());
    return GMG_OK;
} GMG_CATCH_0

int gmg_host_plan_level(int n, const int* colptr, const int* rowidx, const double* val, int mode, int block_rows, int sigma, int64_t* info,
                        int* new2old, int* color_begin, int* blk_begin, unsigned char* row_color) try {
    if (n <= 0 || !colptr || !rowidx || !val || mode < 0 || mode > 4) return GMG_ERR_INVALID;
    if (mode == 1 && (block_rows <= 0 || block_rows > gmgk::kBlockRows || block_rows % 64)) return GMG_ERR_INVALID;
    if (sigma < 0 || sigma % 64) return GMG_ERR_INVALID;
    if (mode == 4) {
        // the colouring that a cold gmg_set_system starts ahead of its inspection, on the caller's arrays as they are (greedy_coloring_ahead): -2 is reported
        // as GMG_ERR_INVALID (arrays that would take a reader out of bounds), a result goes through make_ordering like the one made after the inspection
        PreColoring pre;
        std::atomic<int> stop{0};
        pre.n_colors = greedy_coloring_ahead(n, colptr, rowidx, (int64_t)colptr[n], pre.c8, stop);
        if (pre.n_colors == -2) return GMG_ERR_INVALID;
        if (info) info[5] = pre.n_colors >= 0 ? 1 : 0;
        LevelOrdering o = make_ordering(PatternView{n, colptr, rowidx}, true, (block_rows > 0 && block_rows % 64 == 0) ? block_rows : 64, sigma, 0, nullptr, true, &pre);
        if (o.n_colors > 256) return GMG_ERR_UNSUPPORTED;
        if (info) { info[0] = o.n_pad; info[1] = o.n_colors; info[2] = 0; info[3] = 0; info[4] = 0; }
        if (new2old) std::memcpy(new2old, o.new2old.data(), sizeof(int) * o.n_pad);
        if (color_begin) std::memcpy(color_begin, o.color_begin.data(), sizeof(int) * (o.n_colors + 1));
        return GMG_OK;
    }
    Compressed A;
    A.assign(n, n, colptr, rowidx, val);
    // (mode 2: colour-major with the locality reordering forced -- the colouring then walks a visit ORDER; mode 3: colour-major, row indices declared ascending)
    LevelOrdering o = mode == 1 ? make_block_ordering(A, block_rows)
                                : make_ordering(A, true, (block_rows > 0 && block_rows % 64 == 0) ? block_rows : 64, sigma, mode == 2 ? 1 : 0, nullptr, mode == 3);
    if (o.n_colors > 256) return GMG_ERR_UNSUPPORTED;
    SellHost sa; std::vector<double> dg; std::string e;
    if (!build_operator_sell(A, o, 0, sa, dg, e)) return GMG_ERR_NUMERIC;
    if (info) { info[0] = o.n_pad; info[1] = o.n_colors; info[2] = o.n_blocks(); info[3] = sa.stored(); info[4] = sa.nnz_real; info[5] = 0; }
    if (new2old) std::memcpy(new2old, o.new2old.data(), sizeof(int) * o.n_pad);
    if (color_begin && !o.blocked) std::memcpy(color_begin, o.color_begin.data(), sizeof(int) * (o.n_colors + 1));
    if (blk_begin && o.blocked) std::memcpy(blk_begin, o.blk_begin.data(), sizeof(int) * o.blk_begin.size());
    if (row_color && o.blocked) std::memcpy(row_color, o.row_color.data(), o.row_color.size());
    return GMG_OK;
} GMG_CATCH_0

// set-up fault injection for the tests (gravomg_hip_internal.h)
int gmg_debug_set(gmg_handle h, const char* key, double value) try {
    if (!h || !key) return GMG_ERR_INVALID;
    if (std::string(key) == "col16_uncovered") { h->dbg_col16_uncovered = (int)value; return GMG_OK; }
    return fail(h, GMG_ERR_INVALID, std::string("unknown debug key: ") + key);
} GMG_CATCH_H

int gmg_host_fine_block_rule(int n, const int* colptr, const int* rowidx, const double* val, int* blocked, int* reason) try {
    if (n <= 0 || !colptr || !rowidx || !val || !blocked) return GMG_ERR_INVALID;
    int why = 0;
    if ((double)colptr[n] < kFineBlockMinRow * (double)n) why = 1;
    else if (!stieltjes_signs(n, colptr, rowidx, val, hw_threads())) why = 2;
    *blocked = why == 0 ? 1 : 0;
    if (reason) *reason = why;
    return GMG_OK;
} GMG_CATCH_0

int gmg_host_ldlt_solve(int n, const int* colptr, const int* rowidx, const double* val, const double* b, int d, double* x, int64_t* factor_nnz) try {
    if (n <= 0 || !colptr || !rowidx || !val || !b || !x || d <= 0) return GMG_ERR_INVALID;
    Compressed A;
    A.assign(n, n, colptr, rowidx, val);
    // the engine's coarsest-level solver (supernodal), cross-checked here against the simplicial implementation it replaced
    SupernodalLDLT f;
    if (!f.factor(A)) return GMG_ERR_NUMERIC;
    std::vector<double> w((size_t)n * d);
    f.solve_multi(b, (size_t)n, x, (size_t)n, d, w.data());
    if (factor_nnz) *factor_nnz = f.factor_nnz();
    return GMG_OK;
} GMG_CATCH_0

// Measurement / cross-check aid of the coarsest-level solver (gravomg_hip_internal.h; tests/test_host.py, scripts/ldlt_bench.py): factorises A,
// times the back-substitution on 1 .. 8 threads (`reps` solves per batch, best of 20 batches) and the numeric re-factorisation, compares the
// team solves with the one-thread solve bit for bit and the supernodal factor with the simplicial one, and writes the report (text lines) into
// `report` (cap bytes, NUL-terminated, truncated if need be).
int gmg_host_ldlt_probe(int n, const int* colptr, const int* rowidx, const double* val, const double* b, int reps, char* report, int cap) try {
    if (n <= 0 || !colptr || !rowidx || !val || !b || !report || cap <= 0) return GMG_ERR_INVALID;
    reps = std::max(1, reps);
    Compressed A;
    A.assign(n, n, colptr, rowidx, val);
    SupernodalLDLT f;
    if (!f.factor(A)) return GMG_ERR_NUMERIC;
    std::vector<double> w((size_t)n * 3), x((size_t)n);
    f.solve_multi(b, (size_t)n, x.data(), (size_t)n, 1, w.data());
    std::string text;
    auto say = [&](const char* fmt, auto... args) { char line[1024]; std::snprintf(line, sizeof(line), fmt, args...); text += line; };
    {
        std::vector<double> xb((size_t)n);
        double best = 1e30;
        for (int batch = 0; batch < 20; ++batch) {                     // minimum over batches: the host may be shared
            auto t0 = clk::now();
            for (int i = 0; i < reps; ++i) f.solve_multi(b, (size_t)n, xb.data(), (size_t)n, 1, w.data());
            best = std::min(best, 1e3 * ms_since(t0) / reps);
        }
        say("factorisation: ordering %.2f ms, symbolic %.2f ms, numeric %.2f ms\n", f.phase_ms[0], f.phase_ms[1], f.phase_ms[2]);
        {   // numeric re-factorisation (a system with the same sparsity pattern: the demos' new tau per frame), best and median of 15
            std::vector<double> ts;
            for (int i = 0; i < 15; ++i) { if (!f.factor(A, true)) return GMG_ERR_NUMERIC; ts.push_back(f.phase_ms[2]); }
            std::sort(ts.begin(), ts.end());
            say("numeric re-factorisation on %d thread(s): best %.2f ms, median %.2f ms\n", SupernodalLDLT::numeric_threads(), ts.front(), ts[ts.size() / 2]);
        }
        {   // the factor as the device reads it (export_device_factor) and the device kernel's schedule, on the host: some columns of the inverse by the
            // chunk algorithm against the back-substitution of the same unit vectors
            SupernodalLDLT::DeviceFactor E;
            f.export_device_factor(E);
            std::vector<double> col((size_t)n), e((size_t)n, 0.0), ref((size_t)n), wk((size_t)n);
            double worst = 0.0, scale = 0.0;
            const int picks = std::min(n, 24);
            for (int t = 0; t < picks; ++t) {
                const int c = (int)((long)t * (n - 1) / std::max(picks - 1, 1));          // factor numbering, spread from the first leaf to the root
                SupernodalLDLT::emulate_device_column(E, c, col.data());
                e[(size_t)f.perm[(size_t)c]] = 1.0;
                f.solve(e.data(), ref.data(), wk.data());
                e[(size_t)f.perm[(size_t)c]] = 0.0;
                for (int j = c; j < n; ++j) { worst = std::max(worst, std::fabs(col[(size_t)j] - ref[(size_t)f.perm[(size_t)j]])); scale = std::max(scale, std::fabs(ref[(size_t)f.perm[(size_t)j]])); }
            }
            say("device factor layout: %d chunks of <= %d columns in %d levels; %d columns of the inverse by the chunk algorithm vs back-substitution: max |difference| %.3e of max |entry| %.3e\n",
                E.nq, SupernodalLDLT::kChunk, E.nlev, picks, worst, scale);
        }
        long part[3];
        f.split_report(part);
        say("n=%d nnz(L)=%ld: %.2f us per single-column solve on one thread (best of 20 batches of %d); %d parts of the elimination "
                     "tree, panel entries in the lightest / heaviest part / above them: %ld / %ld / %ld\n", n, f.factor_nnz(), best, reps, f.parts(), part[0], part[1], part[2]);
        for (int threads = 2; threads <= std::min(8, f.parts()); threads += threads < 4 ? 1 : 2) {
            double best2 = 1e30, diff = 0.0;
            SpinTeam team(threads - 1);
            team.arm();
            std::vector<double> x2((size_t)n);
            for (int batch = 0; batch < 20; ++batch) {
                auto t0 = clk::now();
                for (int i = 0; i < reps; ++i) f.solve_multi(b, (size_t)n, x2.data(), (size_t)n, 1, w.data(), &team);
                best2 = std::min(best2, 1e3 * ms_since(t0) / reps);
            }
            team.disarm();
            for (int i = 0; i < n; ++i) diff = std::max(diff, std::fabs(x2[i] - xb[i]));
            double ph[6];
            team.arm();
            f.profile(b, w.data(), &team, reps, ph);
            team.disarm();
            say("%d threads: %.2f us per solve; max |difference| to the one-thread solve %.1e  (phases: parts down %.1f, top down %.1f, top up %.1f, "
                         "parts up %.1f us; %d top supernodes in %d chains)\n", threads, best2, diff, ph[0], ph[1], ph[2], ph[3], (int)ph[4], (int)ph[5]);
        }
        {   // three right-hand sides (the demos' n x 3 call) on the team of a V-cycle solve, against three single-column solves
            const int threads = std::min(8, std::max(2, std::min(f.parts() * 3, hw_threads() - 1)));
            SpinTeam team(threads - 1);
            team.arm();
            std::vector<double> b3((size_t)n * 3), x3((size_t)n * 3), w3((size_t)n * 3), x1((size_t)n);
            for (int c = 0; c < 3; ++c) for (int i = 0; i < n; ++i) b3[(size_t)c * n + i] = b[i] * (c + 1);
            double best3 = 1e30, diff3 = 0.0;
            for (int batch = 0; batch < 20; ++batch) {
                auto t0 = clk::now();
                for (int i = 0; i < reps; ++i) f.solve_multi(b3.data(), (size_t)n, x3.data(), (size_t)n, 3, w3.data(), &team);
                best3 = std::min(best3, 1e3 * ms_since(t0) / reps);
            }
            team.disarm();
            for (int c = 0; c < 3; ++c) {
                f.solve_multi(b3.data() + (size_t)c * n, (size_t)n, x1.data(), (size_t)n, 1, w3.data());
                for (int i = 0; i < n; ++i) diff3 = std::max(diff3, std::fabs(x1[i] - x3[(size_t)c * n + i]));
            }
            say("3 columns on %d threads: %.2f us per solve; max |difference| to single-column solves %.1e\n", threads, best3, diff3);
        }
        { double ph[6]; f.profile(b, w.data(), nullptr, reps, ph);
          say("1 thread phases: parts down %.1f, top down %.1f, top up %.1f, parts up %.1f us\n", ph[0], ph[1], ph[2], ph[3]); }
    }
    {   // the supernodal factorisation against the simplicial one it replaced
        SparseLDLT g;
        if (!g.factor(A)) return GMG_ERR_NUMERIC;
        std::vector<double> xr(n);
        for (int c = 0; c < 1; ++c) {
            g.solve(b + (size_t)c * n, xr.data(), w.data());
            double e2 = 0, n2 = 0;
            for (int i = 0; i < n; ++i) { const double dd = xr[i] - x[(size_t)c * n + i]; e2 += dd * dd; n2 += xr[i] * xr[i]; }
            say("column %d: supernodal vs simplicial relative difference %.3e (nnz(L) %ld vs %ld)\n", c, std::sqrt(e2 / std::max(n2, 1e-300)),
                         f.factor_nnz(), g.factor_nnz());
        }
    }
    std::snprintf(report, (size_t)cap, "%s", text.c_str());
    return GMG_OK;
} GMG_CATCH_0

}  // extern "C"

#include "engine_dist.hip.hpp"
