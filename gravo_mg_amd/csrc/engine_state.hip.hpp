// engine_state.hip.hpp -- part of libgravomg_hip.so's single translation unit (included by engine.hip, in this order:
// engine_state, engine_setup, engine_cycle).  Device memory pool, level / handle structures, error and upload helpers.
#pragma once

// ---- device-wide synchronisation against gated streams (process-wide) ------------------------------------------
// A V-cycle of a polled handle parks its stream behind a word only its own host thread writes (hipStreamWaitValue64, engine_cycle.hip.hpp::
// coarse_host_begin) and goes on enqueuing the way up before it serves that word.  hipFree / hipHostFree / hipStreamDestroy wait for the
// WHOLE device -- every stream, the parked one included -- while holding the runtime's lock: called from another thread (another handle
// being destroyed or re-set while this one solves) they wait for the gate, and the gate's thread waits for the runtime's lock in its next
// launch: a deadlock (found by scripts/soak_factor.py: two threads in gmg_destroy, one in gmg_solve, for ever).  Hence: a gate is held
// under a shared lock from before it is enqueued until its word is written, and the library's device-synchronising calls take the lock
// exclusively -- they wait until no gate of this process is pending.  (Device-wide synchronisations issued by OTHER code in the process
// while a solve runs in another thread cannot be fenced this way: GMG_NO_STREAM_GATE=1 gives the gate up.)
inline std::shared_mutex& gate_mutex() { static std::shared_mutex* m = new std::shared_mutex(); return *m; }      // (leaked: usable during exit)
static thread_local int tl_gates_held = 0;          // gates pending on this thread (it must not take the exclusive lock itself)
struct DeviceSyncGuard {
    bool locked = false;
    DeviceSyncGuard() { if (tl_gates_held == 0) { gate_mutex().lock(); locked = true; } }
    ~DeviceSyncGuard() { if (locked) gate_mutex().unlock(); }
    DeviceSyncGuard(const DeviceSyncGuard&) = delete;
    DeviceSyncGuard& operator=(const DeviceSyncGuard&) = delete;
};
static inline hipError_t sync_hipFree(void* p) { DeviceSyncGuard g; return hipFree(p); }
static inline hipError_t sync_hipHostFree(void* p) { DeviceSyncGuard g; return hipHostFree(p); }
static inline hipError_t sync_hipStreamDestroy(hipStream_t s) { DeviceSyncGuard g; return hipStreamDestroy(s); }

// ---- device memory pool (per handle) ------------------------------------------------------------------------
// hipFree costs ~0.2 ms and synchronises the device; a setup allocates and releases ~100 arrays.  Blocks released
// by a handle are parked in its pool and handed out again (same stream => stream order makes the reuse safe); the
// pool is emptied when the handle is destroyed or when the parked bytes exceed what is in use.
struct DevPool {
    std::multimap<size_t, void*> parked;
    std::map<void*, size_t> size_of;      // every block this pool handed out (live or parked)
    size_t parked_bytes = 0, live_bytes = 0;
    size_t peak_live_bytes = 0;           // high-water mark of live_bytes since reset_peak() (a partitioned set-up holds whole operators for a while)
    void reset_peak() { peak_live_bytes = live_bytes; }
    static size_t round_up(size_t b) { return (std::max<size_t>(b, 1) + 511) & ~(size_t)511; }
    hipError_t alloc(void** p, size_t bytes) {
        const size_t need = round_up(bytes);
        auto it = parked.lower_bound(need);
        if (it != parked.end() && it->first <= need + need / 8 + 65536) {
            *p = it->second; parked_bytes -= it->first; live_bytes += it->first; parked.erase(it);
            peak_live_bytes = std::max(peak_live_bytes, live_bytes);
            return hipSuccess;
        }
        hipError_t e = hipMalloc(p, need);
        if (e != hipSuccess) { trim(); (void)hipGetLastError(); e = hipMalloc(p, need); }
        if (e == hipSuccess) { size_of[*p] = need; live_bytes += need; peak_live_bytes = std::max(peak_live_bytes, live_bytes); }
        return e;
    }
    void release(void* p) {
        auto it = size_of.find(p);
        if (it == size_of.end()) { (void)sync_hipFree(p); return; }          // not ours (allocated outside a pool scope)
        parked.emplace(it->second, p); parked_bytes += it->second; live_bytes -= std::min(live_bytes, it->second);
        if (parked_bytes > std::max<size_t>(live_bytes, (size_t)2 << 30)) trim();
    }
    // the parked blocks of at least min_bytes back to the device (the few big temporaries of a partitioned set-up; the many small ones stay
    // parked: a hipFree costs ~0.2 ms and synchronises the device)
    void trim_large(size_t min_bytes) {
        DeviceSyncGuard g;
        for (auto it = parked.lower_bound(min_bytes); it != parked.end();) { (void)hipFree(it->second); size_of.erase(it->second); parked_bytes -= it->first; it = parked.erase(it); }
    }
    void trim() {
        { DeviceSyncGuard g; for (auto& kv : parked) { (void)hipFree(kv.second); size_of.erase(kv.second); } }
        parked.clear(); parked_bytes = 0;
    }
};
static thread_local DevPool* tl_pool = nullptr;     // set for the duration of a C-ABI call on a handle (PoolScope)
struct PoolScope {
    DevPool* prev;
    explicit PoolScope(DevPool* p) : prev(tl_pool) { tl_pool = p; }
    ~PoolScope() { tl_pool = prev; }
};
static inline hipError_t dev_malloc(void** p, size_t bytes) { return tl_pool ? tl_pool->alloc(p, bytes) : hipMalloc(p, bytes); }
static inline hipError_t dev_free(void* p) { if (!p) return hipSuccess; if (tl_pool) { tl_pool->release(p); return hipSuccess; } return sync_hipFree(p); }

namespace {

struct DevSell {
    int n_slices = 0;
    int lpr = 1;                  // lanes per row of the SELL layout (1 or 4)
    int64_t stored = 0, nnz_real = 0;
    int64_t* slice_ptr = nullptr;
    int* col = nullptr;
    double* val = nullptr;
    float* val32 = nullptr;       // fp32 copy of val (mixed-precision inner cycle); shares slice_ptr / col / row_of
    int* row_of = nullptr;
    unsigned* col16 = nullptr;         // level 0: 16-bit column codes, two to a word, indexed like col (the first half of every slice's region is used)
    int* win_base = nullptr;           //          + the window bases of every slice, first one -1 = slice on 32-bit indices (gmgs::compress_cols); null = the kernels read col
    int c16_dbits = 13;                //          offset bits of a code: 13 = 8 windows of 8 192 columns per slice, 11 = 32 windows of 2 048
    int c16_mode = 0;                  //          0 none, 1 codes from slice c16_from on (uncovered slices are a short prefix), 2 uncovered slices flagged one by one
    int c16_from = 0;
    bool resident = false;             //          the layout is small enough to stay in the memory-side cache between the launches that read it: ordinary loads (c16_sel)
    int uniform_w = 0;                 //          > 0: every slice is this many entries wide, slice s starts at 64 uniform_w s (the kernels then do not read slice_ptr: row_dot)
    int c16_arg() const { return (c16_mode == 1 ? c16_from : 0) | uniform_w << 24 | (c16_dbits == 11 ? 1 << 30 : 0); }      // the kernels' c16_arg (kernels.hip.hpp::row_dot_sel)
    int c16_sel() const { return c16_mode ? c16_mode + (resident ? 2 : 0) : 0; }      // the kernels' C16 template argument: 3 / 4 = modes 1 / 2 read with ordinary loads
};

// natural-numbering compressed matrix on the device (A_k, U_k by coarse column)
struct DevCsr {
    int n_outer = 0;
    int *ptr = nullptr, *idx = nullptr;
    double* val = nullptr;
};

// U_k regrouped by fine row (<= 3 entries per row, sorted by coarse column) for the prolongation layout and the RAP.
struct DevEll3 {
    int n = 0;
    int *cnt = nullptr, *col = nullptr;
    double* val = nullptr;
};

struct Level {
    int n = 0, n_pad = 0;
    int64_t nnz = 0;              // entries of A_k
    LevelOrdering ord;
    Compressed A;                 // natural numbering, host copy (Abar[k]); filled on demand (ensure_host_A) except on level L
    bool hostA_pattern = false, hostA_values = false;
    DevCsr dA;                    // natural numbering, device copy: RAP input, layout source, source of the lazy host copy
    DevSell Aoff;                 // off-diagonal part, device numbering
    double* diag = nullptr;       // n_pad
    DevSell P, R;                 // U_k (rows: this level) and U_k^T (rows: next level); unused on level L
    // blocked levels (block-hybrid Gauss-Seidel): in-block part (16-bit local columns) + off-block part
    DevSell Ain, Aout;
    unsigned short* ain_col16 = nullptr;
    // big blocked levels: block-CSR storage instead of the two padded SELL operators (kernels.hip.hpp::gs_blockcsr)
    bool use_bcsr = false;
    int *bc_ptr = nullptr, *bc_mid = nullptr, *bc_col = nullptr;
    double* bc_val = nullptr;
    float* bc_val32 = nullptr;
    int bc_cap = 0;               // entries of the largest block, rounded up to 64 (LDS capacity of the sweep)
    int64_t bc_nnz = 0;
    // big blocked levels, default: the unpadded block sweep (kernels.hip.hpp::gs_block_ep) on two block-ordered CSRs --
    // ee_*: "explicit" part (off-block entries + in-block entries with a later column; device columns),
    // ep_*: "lower" part (in-block entries with an earlier column; 16-bit local columns).  No SELL split, no bc_* then.
    bool use_ep = false;
    int *ee_ptr = nullptr, *ee_col = nullptr, *ep_ptr = nullptr;
    unsigned short* ep_col = nullptr;
    double *ee_val = nullptr, *ep_val = nullptr;
    float *ee_val32 = nullptr, *ep_val32 = nullptr;
    int64_t ee_nnz = 0, ep_nnz = 0;
    int ep_cap_e = 0, ep_cap_l = 0;      // most explicit / lower entries of one block (explicit: rounded up to 64): LDS capacity of the sweep
    int *d_blk_begin = nullptr, *d_blk_ncolors = nullptr;
    unsigned char* d_row_color = nullptr;
    // where every stored slot of the operator layouts came from in dA.val (-1: padding), made with the layouts: a system with the live
    // sparsity pattern refreshes the values by plain gathers (engine_setup.hip.hpp::device_refill_level); not made for a partitioned set-up
    int *src_A = nullptr, *src_diag = nullptr, *src_ee = nullptr, *src_ep = nullptr;
    int* d_new2old = nullptr;
    int* d_old2new = nullptr;         // kept after the layout (with d_blk_of_row) so that a system with the same pattern can
    int* d_blk_of_row = nullptr;      // refill the value arrays in place (refresh_system_values)
    double *x = nullptr, *b = nullptr, *r = nullptr, *tmp = nullptr;   // n_pad * dcap
    float *diag32 = nullptr, *x32 = nullptr, *b32 = nullptr, *r32 = nullptr, *tmp32 = nullptr;   // mixed precision
};

}  // namespace

// The level-0 point graph a hierarchy was built from (`neigh` + the diagonal) as a canonical sparsity pattern (host_plan.hpp::neigh_pattern):
// the pattern of every system the hierarchy is built for (tau M + S, M + tau S of that mesh / point cloud).  Shared, read-only.
struct FineGraph {
    int n = 0;
    RawVec<int> ptr, idx;
};

struct gmg_hierarchy_s {
    HierarchyResult res;
    std::shared_ptr<const FineGraph> graph;      // made beside the construction; gmg_use_hierarchy hands it to the engine (gmg_set_fine_graph does the same from a table)
    std::vector<int> fine_order;         // breadth-first order of the level-0 points over `neigh` (new -> old); empty when the input order is local already
};

struct DistP2P;

// Who owns and who reads what in a row-partitioned job: computed identically on every rank from the orderings of levels 0 / 1,
// the level-0 / level-1 patterns and U_0 (engine_part.hip.hpp::build_dist_plan).  Level 0: every colour class (padded to 64 * world rows) is cut
// into `world` equal contiguous pieces, rank p owns piece p of every colour.  Level 1 (shard1): by 64-row blocks, a block to the rank that owns
// most of the fine rows its points prolong into.  Lists are in device numbering, ascending.
struct DistPlan {
    int rank = 0, world = 1, n_colors = 0;
    bool shard1 = false;
    uint64_t key[2] = {0, 0};                             // pattern_key of the system the plan was made for (orderings and plan go together)
    std::vector<int> blk_owner;                           // [block of level 1] -> rank
    std::vector<std::vector<int>> own_blocks;             // [rank]: its blocks, ascending (block b = device rows 64 b .. 64 b + 63)
    std::vector<std::vector<int>> halo;                   // [(s * world + t) * (C + 1) + k]: level-0 rows rank s publishes to rank t for colour k (k = C: all colours)
    std::vector<std::vector<int>> halo1, halo0r;          // [s * world + t]: x1 entries / r0 entries rank s publishes to rank t
};

struct gmg_solver_s {
    DevPool pool;
    DistP2P* p2p = nullptr;              // engine-driven multi-GPU cycle (engine_dist.hip.hpp)
    int dbg_col16_uncovered = 0;         // gmg_debug_set("col16_uncovered"): set-up fault injection of the tests (gravomg_hip_internal.h)
    gmg_config cfg;
    std::string err;
    bool has_device = false;
    int n_cus = 256;                      // compute units of the device (launch geometry of the persistent kernels)
    hipStream_t stream = nullptr;
    int L = -1;
    std::vector<Compressed> U;
    std::vector<char> U_set;
    std::vector<DevCsr> dU;               // device copies of U_k (kept while the hierarchy is unchanged)
    std::vector<DevEll3> dE3;             // and their by-row regrouping
    bool dU_ready = false;
    // patches of the blocked levels k >= 1, grown over the coarse point graph of U_{k-1} (hierarchy data, host only)
    std::vector<PatchSet> patches;
    std::vector<int> cluster_order;       // locality-preserving order of the level-0 points derived from U (new -> old)
    int *d_cluster_order = nullptr, *d_cluster_inv = nullptr;      // device copies (order, and old -> new position)
    std::vector<int> bfs_order;           // ... and the breadth-first order over the point graph, when the caller / the hierarchy object supplied one (gmg_set_fine_order)
    // The pattern the next systems are expected to have (gmg_set_fine_graph / gmg_use_hierarchy).  gmg_finalize_hierarchy then builds everything
    // STRUCTURAL for it -- orderings, colourings, layouts, symbolic Galerkin products, symbolic LDL^T: a whole set-up on placeholder values --
    // and leaves the handle in the "placeholder" state: no system to solve with (system_ready stays false), but the first gmg_set_system whose
    // pattern digest equals the prepared one only moves values (refresh_system_values), like any later system with the live pattern.
    std::shared_ptr<const FineGraph> fine_graph;
    bool placeholder_ready = false;
    bool mass_dirty = false;              // h->mass changed while no ordering existed to permute it with: uploaded by the next set-up / refresh
    int *d_bfs_order = nullptr, *d_bfs_inv = nullptr;
    int base_order_choice = 0;            // what the last reordered set-up used: 0 cluster order, 1 breadth-first order
    RawVec<int> reo_ptr, reo_idx;         // LHS pattern permuted into cluster order (staging for the level-0 colouring)
    bool patches_ready = false;
    bool dU_flagged = false;              // ell3_from_csc found a U row with more than 3 entries (host paths only)
    std::vector<double> mass;
    std::vector<Level> lv;
    SupernodalLDLT coarse;
    bool system_ready = false;
    int dcap = 0;
    double *d_mass = nullptr, *d_minv = nullptr;
    double* d_stage = nullptr; size_t stage_cap = 0;
    double* h_stage[2] = {nullptr, nullptr}; size_t h_stage_cap = 0;      // pinned host staging (double-buffered) for b / x
    void* bounce[2] = {nullptr, nullptr};                                  // pinned bounce buffers for the set-up's pageable copies
    hipEvent_t bounce_ev[2] = {nullptr, nullptr}; int bounce_flip = 0;
    hipEvent_t h_stage_ev[2] = {nullptr, nullptr}; int h_stage_flip = 0;
    hipEvent_t h_chunk_ev[16] = {};      // per-chunk arrival of a download (to_host): 8 per staging buffer
    double* d_partials = nullptr; int partial_blocks = 0;
    double* d_norm = nullptr;
    // Head of the next cycle (gmg_config::speculate_head; engine.hip::solve_common): the solve loop's decision is taken by the check's
    // reduction on the device (d_watch: {double least; int go}), the first colour launch of the next cycle is enqueued behind it before the
    // host has seen the norm, and returns at once when the iteration stopped.  watch_*: what the reduction needs to decide; head_enqueued:
    // the next level-0 pre-smoothing starts with its second launch.
    double* d_watch = nullptr;
    bool watch_active = false; int watch_mode = 0, watch_type = 0, watch_cycles_done = 0; double watch_tol = 0.0;
    bool head_enqueued = false;
    double* h_pinned = nullptr; size_t pinned_cap = 0;     // coarse rhs / solution staging
    double* h_norm = nullptr;
    // polled completion (stream launches only): a kernel writes its small result into pinned memory and then a sequence
    // number into h_flag[slot]; the host spins on that word (wait_flag) instead of a copy + hipStreamSynchronize
    // residual check folded into the last colour launch of the post-smoothing (engine_cycle.hip.hpp): type of the check the cycle
    // being enqueued ends with (-1: none / not foldable), and the number of partial-sum blocks that launch produced (0: it did not)
    int fuse_norm_type = -1, fuse_norm_blocks = 0;
    // the same for the residual after the pre-smoothing: where the last colour launch shall write its rows' residual (null: nowhere),
    // and the first slice it covered (0: it did not)
    double* fuse_res_out = nullptr; int fuse_res_from = 0;
    unsigned long long* h_flag = nullptr; unsigned long long flag_seq[3] = {0, 0, 0};
    // h_flag[16]: written by the HOST -- the stream waits on it (hipStreamWaitValue64) in front of the work that needs the host's
    // coarsest solution, so that work is enqueued before the host solves (engine_cycle.hip.hpp::coarse_host_begin / _serve)
    const void* sweep_prev = nullptr;    // the iterate the last block sweep of launch_block_sweeps started from (nullptr: zero) ...
    bool sweep_prev_valid = false;       // ... valid until the next launch touches the level (enqueue_down consumes it: residual_delta_ep)
    bool gate_ok = true;                 // false once hipStreamWaitValue64 was refused: launch after the solve instead
    bool gate_proven = false;            // true once a published right-hand side was seen by the polling host while the stream was busy (ungated)
    bool coarse_pending = false;         // a gate is enqueued and the host has not answered it yet
    bool gate_shared = false;            // ... and this thread holds gate_mutex() shared for it (released by coarse_host_serve)
    bool coarse_warm = false;            // the solving thread has read the current factor once (SupernodalLdlt::warm)
    double coarse_warm_sink = 0.0;
    int coarse_pending_d = 0;
    bool poll = true;             // GMG_POLL=0: copy + hipStreamSynchronize instead (the waiting thread then sleeps instead of spinning)
    double* d_ainv = nullptr;                              // coarse_device: dense A_L^-1, level numbering (engine.hip::build_coarse_inverse_device)
    int ainv_n = 0, ainv_ld = 0;                           // ... n x n with rows ainv_ld doubles apart (n rounded up to 8: rows at 64-byte boundaries)
    bool preparing_structure = false;                      // inside prepare_structure (placeholder values)
    bool first_sweep_fused = false;                        // enqueue_down: the restriction into the next level ran that level's first pre-sweep (launch_restrict_sweep0)
    bool coarse_device = false;                            // the coarsest solve of the live system runs on the device (gmg_config::coarse_mode, decided per system)
    std::vector<double> coarse_work;
    std::unique_ptr<SpinTeam> coarse_helper;        // the other threads of the coarsest back-substitution, armed for the duration of a solve (HelperScope)
    std::map<std::string, double> timing;
    std::map<int, hipGraphExec_t> graphs;
    int loaded_d = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second stream + event: the numeric Galerkin pass of level 1 runs there, range by range, behind the chunks of a values upload (set_system_impl)
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_ev = nullptr;
    int* d_aux_err = nullptr;             // error flag of the kernels queued on aux_stream
    // the coarse rows of level 1 in the order in which the rows of A_0 they need arrive: d_rap_order[i] = coarse row, rap_need[i] = an upper bound of the
    // largest fine row the rows d_rap_order[0 .. i] prolong from (ascending): those are computable once that row has arrived (ensure_rap_order)
    std::vector<int> rap_need;
    int* d_rap_order = nullptr;
    // ... and the same for level 2: its row q is computable once the level-1 rows it prolongs from (U_1's column q) are, i.e. once the fine row
    // rap_need2[i] = max over those of their rap_need has arrived
    std::vector<int> rap_need2;
    int* d_rap_order2 = nullptr;
    std::vector<hipEvent_t> prof_ev;     // gmg_profile_cycle: events at the boundaries of a cycle's legs (prof_on: record them)
    bool prof_on = false; int prof_n = 0;
    bool il_r0 = false;                   // enqueue_down, level 0, d > 1: the residual is being written as an interleaved multi-vector
    void* il_sweep_out = nullptr;         // enqueue_up: where level 1's last post-sweep writes the interleaved copy of its x ...
    bool il_sweep_done = false;           // ... and whether it did (the level-0 prolongation then gathers from it)
    // multi-GPU (one process per GPU): this rank's share of level 0, externally owned level-0 vectors
    hipStream_t own_stream = nullptr;
    int rank = 0, world = 1;
    bool dist_ready = false;
    // gmg_dist_partition: the next set-ups lay out and keep only rank part_rank's rows of levels 0-1 (of part_world ranks; 1 = whole systems).
    // `partitioned`: the live system is such a share -- its level-0 / level-1 operators hold this rank's rows only, the natural-numbering
    // copies (A_0, A_1, U_0) were released after the set-up, and only the gmg_p2p_* / gmg_dist_* entry points may run on it.
    int part_rank = 0, part_world = 1;
    bool partitioned = false;
    std::shared_ptr<DistPlan> plan;       // of the live (or last) partitioned system; reused while the pattern digest and the partition stand
    bool refill_ready = false;        // the live layout was built by the device builders from device-resident A_k: a system with
                                      // the same sparsity pattern only needs its values refreshed
    bool dist_all_rows = false;
    double *own_x0 = nullptr, *own_b0 = nullptr, *own_r0 = nullptr;   // engine-owned buffers parked while external ones are bound
    bool bound = false;
    // orderings of the last system, reusable while the sparsity pattern of the LHS and the hierarchy are unchanged
    bool ord_cache_valid = false;
    uint64_t ord_cache_key[2] = {0, 0};
    std::vector<LevelOrdering> ord_cache;
    // the orderings live in the levels while a system is set; they move into ord_cache when the next one arrives
    uint64_t live_key[2] = {0, 0};
    bool live_key_valid = false;
};

namespace {

// Classes of rows a partitioned level 0 is cut by (engine_dist.hip.hpp): its colour classes, or -- blocked (gmg_config::block_fine) -- the one range
// of all rows (LevelOrdering::color_begin = {0, n_pad}; n_colors then counts the colours INSIDE a block).
inline int dist_classes(const LevelOrdering& o) { return o.blocked ? 1 : o.n_colors; }

int fail(gmg_handle h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define HIPCHK(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) return fail(h, GMG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define NEED_DEVICE()                                                                     \
    if (!h) return GMG_ERR_INVALID;                                                       \
    PoolScope pool_scope_(&h->pool);                                                      \
    do {                                                                                  \
        if (!h->has_device) return fail(h, GMG_ERR_NO_DEVICE, "no usable HIP device (libgravomg_hip has no CPU fallback)"); \
    } while (0)

// Host <-> device copies of pageable memory (the caller's arrays, std::vectors) go through two pinned bounce buffers:
// worker threads copy a chunk in while the previous one is on the wire.  Handing pageable memory to hipMemcpyAsync
// makes the runtime pin and unpin it (252 MB for the LHS at 3 M vertices); on hosts where that is expensive (IOMMU
// translation on) the deferred unpinning stalled the NEXT DMA by 10-35 ms -- seen as a late start of the first
// right-hand-side upload after a set-up -- and the set-up itself was slower.  The bounce costs a threaded memcpy and
// behaves the same everywhere.  A side effect: the source may be freed as soon as h2d returns.
constexpr size_t kBounceBytes = (size_t)16 << 20;
constexpr size_t kBounceMin = (size_t)256 << 10;       // smaller copies use the runtime's own staging

static int ensure_bounce(gmg_handle h) {
    for (int i = 0; i < 2; ++i) {
        if (!h->bounce[i]) HIPCHK(hipHostMalloc(&h->bounce[i], kBounceBytes, hipHostMallocDefault));
        if (!h->bounce_ev[i]) HIPCHK(hipEventCreateWithFlags(&h->bounce_ev[i], hipEventDisableTiming));
    }
    return GMG_OK;
}

static void threaded_copy_bytes(void* dst, const void* src, size_t bytes, int threads) {
    const int T = (int)std::min<size_t>(std::max(1, std::min(threads, 16)), bytes / ((size_t)1 << 20) + 1);
    // small copies stay on the calling thread: a range handed to the worker pool queues behind whatever the set-up's
    // ordering tasks have submitted (1 MB took 2 ms that way, 0.1 ms inline)
    if (T <= 1 || bytes < ((size_t)8 << 20)) { std::memcpy(dst, src, bytes); return; }
    parallel_ranges(T, T, [&](int t0, int t1, int) {
        for (int t = t0; t < t1; ++t) {
            const size_t lo = bytes * t / T / 64 * 64, hi = t + 1 == T ? bytes : bytes * (t + 1) / T / 64 * 64;
            std::memcpy((char*)dst + lo, (const char*)src + lo, hi - lo);
        }
    }, 1);
}

// after_chunk (optional): called after every chunk has been put on the wire, with the bytes issued so far and the event that completes with that chunk
static int h2d(gmg_handle h, void* dst, const void* src, size_t bytes, const std::function<void(size_t, hipEvent_t)>* after_chunk = nullptr) {
    if (bytes < kBounceMin) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream)); return GMG_OK; }
    int rc = ensure_bounce(h);
    if (rc) return rc;
    const bool trace = EnvSwitches::get().trace_setup;
    double t_wait = 0, t_copy = 0, t_issue = 0;
    auto t_all = clk::now();
    for (size_t off = 0; off < bytes; off += kBounceBytes) {
        const size_t len = std::min(kBounceBytes, bytes - off);
        const int f = h->bounce_flip;
        h->bounce_flip ^= 1;
        auto t0 = clk::now();
        HIPCHK(hipEventSynchronize(h->bounce_ev[f]));                  // the previous DMA out of this buffer is done
        auto t1 = clk::now();
        threaded_copy_bytes(h->bounce[f], (const char*)src + off, len, h->cfg.host_threads);
        auto t2 = clk::now();
        HIPCHK(hipMemcpyAsync((char*)dst + off, h->bounce[f], len, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipEventRecord(h->bounce_ev[f], h->stream));
        if (after_chunk) (*after_chunk)(off + len, h->bounce_ev[f]);
        if (trace) { t_wait += std::chrono::duration<double, std::milli>(t1 - t0).count(); t_copy += std::chrono::duration<double, std::milli>(t2 - t1).count(); t_issue += ms_since(t2); }
    }
    if (trace && bytes >= ((size_t)32 << 20))
        std::fprintf(stderr, "[gmg setup] h2d %.0f MB in %.2f ms: waiting for a free bounce buffer %.2f, host copy %.2f, issuing %.2f ms\n", bytes / 1048576.0, ms_since(t_all), t_wait, t_copy, t_issue);
    return GMG_OK;
}

// (synchronous: the data is in dst when this returns)
static int d2h(gmg_handle h, void* dst, const void* src, size_t bytes) {
    if (bytes < kBounceMin) {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return GMG_OK;
    }
    int rc = ensure_bounce(h);
    if (rc) return rc;
    const size_t nchunk = (bytes + kBounceBytes - 1) / kBounceBytes;
    auto issue = [&](size_t c) -> int {
        const int f = (int)(c & 1);
        const size_t off = c * kBounceBytes, len = std::min(kBounceBytes, bytes - off);
        HIPCHK(hipMemcpyAsync(h->bounce[f], (const char*)src + off, len, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipEventRecord(h->bounce_ev[f], h->stream));
        return GMG_OK;
    };
    HIPCHK(hipEventSynchronize(h->bounce_ev[0]));
    HIPCHK(hipEventSynchronize(h->bounce_ev[1]));
    if ((rc = issue(0))) return rc;
    for (size_t c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk && (rc = issue(c + 1))) return rc;        // next chunk on the wire while this one is copied out
        const int f = (int)(c & 1);
        const size_t off = c * kBounceBytes, len = std::min(kBounceBytes, bytes - off);
        HIPCHK(hipEventSynchronize(h->bounce_ev[f]));
        threaded_copy_bytes((char*)dst + off, h->bounce[f], len, h->cfg.host_threads);
    }
    h->bounce_flip = 0;
    return GMG_OK;
}

template <class T, class A>
int upload(gmg_handle h, T** dst, const std::vector<T, A>& src) {
    if (*dst) { (void)dev_free(*dst); *dst = nullptr; }
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(dev_malloc((void**)dst, bytes));
    if (!src.empty()) return h2d(h, *dst, src.data(), src.size() * sizeof(T));
    return GMG_OK;
}

template <class T>
int upload(gmg_handle h, T** dst, const RawVec<T>& src) {
    if (*dst) { (void)dev_free(*dst); *dst = nullptr; }
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(dev_malloc((void**)dst, bytes));
    if (!src.empty()) return h2d(h, *dst, src.data(), src.size() * sizeof(T));
    return GMG_OK;
}

void free_sell(DevSell& s) {
    if (s.slice_ptr) (void)dev_free(s.slice_ptr);
    if (s.col) (void)dev_free(s.col);
    if (s.val) (void)dev_free(s.val);
    if (s.val32) (void)dev_free(s.val32);
    if (s.row_of) (void)dev_free(s.row_of);
    if (s.col16) (void)dev_free(s.col16);
    if (s.win_base) (void)dev_free(s.win_base);
    s = DevSell();
}

int upload_sell(gmg_handle h, DevSell& d, const SellHost& s) {
    free_sell(d);
    d.n_slices = s.n_slices; d.stored = s.stored(); d.nnz_real = s.nnz_real; d.lpr = s.lpr;
    int rc;
    if ((rc = upload(h, &d.slice_ptr, s.slice_ptr))) return rc;
    if ((rc = upload(h, &d.col, s.col))) return rc;
    if ((rc = upload(h, &d.val, s.val))) return rc;
    if (!s.row_of.empty() && (rc = upload(h, &d.row_of, s.row_of))) return rc;
    return GMG_OK;
}

void free_csr(DevCsr& m) {
    if (m.ptr) (void)dev_free(m.ptr);
    if (m.idx) (void)dev_free(m.idx);
    if (m.val) (void)dev_free(m.val);
    m = DevCsr();
}

void free_ell3(DevEll3& e) {
    if (e.cnt) (void)dev_free(e.cnt);
    if (e.col) (void)dev_free(e.col);
    if (e.val) (void)dev_free(e.val);
    e = DevEll3();
}

void drop_device_transfers(gmg_handle h) {
    if (h->d_cluster_order) { (void)dev_free(h->d_cluster_order); h->d_cluster_order = nullptr; }
    if (h->d_cluster_inv) { (void)dev_free(h->d_cluster_inv); h->d_cluster_inv = nullptr; }
    if (h->d_bfs_order) { (void)dev_free(h->d_bfs_order); h->d_bfs_order = nullptr; }
    if (h->d_bfs_inv) { (void)dev_free(h->d_bfs_inv); h->d_bfs_inv = nullptr; }
    for (auto& m : h->dU) free_csr(m);
    for (auto& e : h->dE3) free_ell3(e);
    h->dU.clear(); h->dE3.clear();
    h->dU_ready = false;
    h->dU_flagged = false;
}

// Patches of every blocked level k >= 1 (see coarse_point_graph): depend on the hierarchy and on block_rows only.
void build_patches(gmg_handle h) {
    const int L = h->L;
    const bool trace = EnvSwitches::get().trace_setup;
    auto t_bp = clk::now();
    auto tr = [&](const char* what, clk::time_point t0) { if (trace) std::fprintf(stderr, "[gmg setup] build_patches %-18s %.2f ms (at %.2f)\n", what, ms_since(t0), ms_since(t_bp)); };
    h->patches.assign(L + 1, PatchSet());
    h->cluster_order.clear();
    const bool mc = h->cfg.smoother == GMG_SMOOTHER_MULTICOLOR_GS;
    // U_k by fine row: shared by the coarse point graphs (patches of level k + 1) and the cluster order of level 0
    std::vector<Compressed> Urows(L);
    {
        std::vector<std::future<void>> jobs;
        for (int k = 0; k < L; ++k) jobs.push_back(std::async(std::launch::async, [h, k, &Urows] { Urows[k] = transpose_parallel(h->U[k]); }));
        for (auto& j : jobs) j.get();
    }
    tr("transposes", t_bp);
    std::vector<std::future<void>> jobs;
    if (mc && h->cfg.block_rows > 0)
        for (int k = std::max(1, h->cfg.block_from_level); k < L; ++k)
            jobs.push_back(std::async(std::launch::async, [h, k, &Urows, &tr] {
                auto t0 = clk::now();
                Compressed G = coarse_point_graph(h->U[k - 1], Urows[k - 1]);
                tr(("point graph l" + std::to_string(k)).c_str(), t0); t0 = clk::now();
                h->patches[k] = grow_patch_set(G, h->cfg.block_rows);
                // A prolongation that is not the hierarchy's (piecewise-constant aggregation: U^T U is diagonal, the point graph has no edges) leaves
                // patches of one or two points -- a 64-row block per point.  Below a quarter of the block size on average the patches are
                // dropped and the level's blocks are grown over its own operator at set-up time (make_block_ordering's fallback)
                const long n_patches = (long)h->patches[k].mem_begin.size() - 1;
                if (n_patches > 0 && (long)G.n_outer * 4 < n_patches * (long)h->cfg.block_rows) h->patches[k] = PatchSet();
                tr(("grow patches l" + std::to_string(k)).c_str(), t0);
            }));
    if (mc && h->cfg.reorder_fine != 0 && L > 0)
        jobs.push_back(std::async(std::launch::async, [h, L, &Urows, &tr] {
            auto t0 = clk::now();
            Compressed GL = coarse_point_graph(h->U[L - 1], Urows[L - 1]);
            h->cluster_order = cluster_order(Urows, GL);
            tr("cluster order", t0);
        }));
    for (auto& j : jobs) j.get();
    h->patches_ready = true;
}

// Patches of a blocked level 0: runs of block_rows consecutive points of the hierarchy's cluster order (points grouped by parent,
// parents by grandparent, ...: a run of 64 is about one cell of level 2 -- a compact 2-D patch whatever the input numbering is,
// every block full).  Depends on the hierarchy only: made on first use (after build_patches), kept while the hierarchy stands.  Null
// without a cluster order (reorder_fine = 0): make_block_ordering then grows the blocks breadth-first over the operator's graph.
const PatchSet* level0_patches(gmg_handle h, int n) {
    if (h->patches.empty() || (int)h->cluster_order.size() != n || h->cfg.block_rows <= 0) return nullptr;
    PatchSet& p = h->patches[0];
    if (p.valid() && p.n == n) return &p;
    const int br = h->cfg.block_rows, nb = (n + br - 1) / br;
    p.n = n;
    p.members = h->cluster_order;
    p.block_of.assign((size_t)n, 0);
    p.mem_begin.resize((size_t)nb + 1);
    for (int b = 0; b <= nb; ++b) p.mem_begin[b] = std::min(n, b * br);
    parallel_ranges(n, std::min(hw_threads(), 32), [&](int lo, int hi, int) { for (int r = lo; r < hi; ++r) p.block_of[p.members[r]] = r / br; }, 1 << 16);
    return &p;
}

void free_level(Level& l) {
    free_csr(l.dA);
    free_sell(l.Aoff); free_sell(l.P); free_sell(l.R); free_sell(l.Ain); free_sell(l.Aout);
    if (l.ain_col16) { (void)dev_free(l.ain_col16); l.ain_col16 = nullptr; }
    for (int** p : {&l.bc_ptr, &l.bc_mid, &l.bc_col}) { if (*p) (void)dev_free(*p); *p = nullptr; }
    if (l.bc_val) { (void)dev_free(l.bc_val); l.bc_val = nullptr; }
    if (l.bc_val32) { (void)dev_free(l.bc_val32); l.bc_val32 = nullptr; }
    l.use_bcsr = false; l.bc_cap = 0; l.bc_nnz = 0;
    for (int** q : {&l.ee_ptr, &l.ee_col, &l.ep_ptr}) { if (*q) (void)dev_free(*q); *q = nullptr; }
    if (l.ep_col) { (void)dev_free(l.ep_col); l.ep_col = nullptr; }
    for (double** q : {&l.ee_val, &l.ep_val}) { if (*q) (void)dev_free(*q); *q = nullptr; }
    for (float** q : {&l.ee_val32, &l.ep_val32}) { if (*q) (void)dev_free(*q); *q = nullptr; }
    l.use_ep = false; l.ee_nnz = l.ep_nnz = 0; l.ep_cap_e = l.ep_cap_l = 0;
    if (l.d_blk_begin) { (void)dev_free(l.d_blk_begin); l.d_blk_begin = nullptr; }
    if (l.d_blk_ncolors) { (void)dev_free(l.d_blk_ncolors); l.d_blk_ncolors = nullptr; }
    if (l.d_row_color) { (void)dev_free(l.d_row_color); l.d_row_color = nullptr; }
    for (double** p : {&l.diag, &l.x, &l.b, &l.r, &l.tmp}) { if (*p) (void)dev_free(*p); *p = nullptr; }
    for (float** p : {&l.diag32, &l.x32, &l.b32, &l.r32, &l.tmp32}) { if (*p) (void)dev_free(*p); *p = nullptr; }
    for (int** q : {&l.src_A, &l.src_diag, &l.src_ee, &l.src_ep}) { if (*q) (void)dev_free(*q); *q = nullptr; }
    if (l.d_new2old) { (void)dev_free(l.d_new2old); l.d_new2old = nullptr; }
    if (l.d_old2new) { (void)dev_free(l.d_old2new); l.d_old2new = nullptr; }
    if (l.d_blk_of_row) { (void)dev_free(l.d_blk_of_row); l.d_blk_of_row = nullptr; }
}

void drop_graphs(gmg_handle h) {
    for (auto& kv : h->graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    h->graphs.clear();
}

void unbind_level0(gmg_handle h) {
    if (h->bound && !h->lv.empty()) { h->lv[0].x = h->own_x0; h->lv[0].b = h->own_b0; h->lv[0].r = h->own_r0; }
    h->bound = false; h->own_x0 = h->own_b0 = h->own_r0 = nullptr;
}

void drop_system(gmg_handle h) {
    h->refill_ready = false;
    h->placeholder_ready = false;
    drop_graphs(h);
    unbind_level0(h);
    h->dist_ready = false;
    for (auto& l : h->lv) free_level(l);
    h->lv.clear();
    h->system_ready = false;
    h->dcap = 0;
    h->loaded_d = 0;
    if (h->d_mass) { (void)dev_free(h->d_mass); h->d_mass = nullptr; }
    if (h->d_minv) { (void)dev_free(h->d_minv); h->d_minv = nullptr; }
    if (h->d_ainv) { (void)dev_free(h->d_ainv); h->d_ainv = nullptr; }
    h->ainv_n = 0;
}

// blocked levels smaller than this use 4 lanes per row (measured in round 4, profiles/r04/d_quad_threshold_ab.txt)
constexpr int kQuadLevelRows = 65536;
constexpr int kEpMaxBlockEntries = 6144;        // largest explicit chunk of a block the unpadded sweep keeps in LDS (48 KB of fp64 products)
constexpr int kEpMaxBlockLower = 3584;          // ... and largest lower chunk (staged as 16-byte records: 56 KB; the launch stays below the 64 KB of dynamic LDS)
constexpr int kBcsrMaxBlockEntries = 4096;      // largest block the block-CSR sweep stages in LDS (48 KB of fp64 entries)
inline bool wants_block_csr(gmg_handle h, int lpr) { return h->cfg.block_csr != 0 && lpr == 1 && h->cfg.block_rows == 64; }
inline bool wants_block_ep(gmg_handle h, int lpr) { return h->cfg.block_ep != 0 && lpr == 1 && h->cfg.block_rows == 64; }

}  // namespace
