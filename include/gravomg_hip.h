/*
 * gravomg_hip.h -- C-ABI of libgravomg_hip.so: the MI355X (gfx950) engine for the Gravo MG
 * multigrid V-cycle hot path.
 *
 * The reference (rubenwiersma/gravo_mg) has no FFI seam for this path: MGBS::MultigridSolver::solve
 * calls Eigen expression templates inline.  This header IS the seam a
 * maintainer would put directly under `multiGridVCycleGS` / `solve`; each entry point names the
 * reference code it replaces (paths relative to the reference repository root).  INTEGRATION.md shows
 * the binding a reference maintainer would add (C++ member functions of MGBS::MultigridSolver and the
 * pybind11 shim).
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns a gmg_status (0 = ok, < 0 = error);
 *     gmg_last_error(h) returns a human-readable message for the last failing call on that handle.
 *   - sparse matrices are passed exactly as Eigen::SparseMatrix<double> stores them: CSC, int32
 *     indices, sorted inner indices (`colptr[ncols+1]`, `rowidx[nnz]`, `val[nnz]`).  System matrices
 *     must be symmetric (the reference's Gauss-Seidel depends on it, multigrid_solver.cpp:1200-1208).
 *   - dense multi-vectors are column-major n x d (Eigen::MatrixXd): column c starts at ptr + c*n.
 *   - host pointers are borrowed for the duration of the call only; device memory, the HIP stream and
 *     the (optional) captured hipGraphs are owned by the handle.  A handle is not thread-safe.
 *   - all device entry points fail with GMG_ERR_NO_DEVICE when no HIP device is usable; there is no
 *     CPU fallback behind this interface.
 */
#ifndef GRAVOMG_HIP_H
#define GRAVOMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gmg_solver_s* gmg_handle;
typedef struct gmg_hierarchy_s* gmg_hierarchy;

typedef enum {
    GMG_OK = 0,
    GMG_ERR_INVALID = -1,      /* bad argument / shape */
    GMG_ERR_NO_DEVICE = -2,    /* no usable HIP device */
    GMG_ERR_HIP = -3,          /* a HIP runtime call failed */
    GMG_ERR_STATE = -4,        /* call order: hierarchy / system not set */
    GMG_ERR_NUMERIC = -5,      /* zero/missing diagonal, singular coarsest operator */
    GMG_ERR_UNSUPPORTED = -6,  /* option outside the hot-path scope (F/W-cycle, SIG06, ...) */
    GMG_DIVERGED = 1           /* gmg_solve only, NOT an error: the iteration did not contract (see gmg_solve); x holds the last iterate */
} gmg_status;

enum { GMG_SMOOTHER_MULTICOLOR_GS = 0, GMG_SMOOTHER_JACOBI = 1 };
/* where the coarsest direct solve (multigrid_solver.cpp:1075) is applied.  HOST_LDLT: back-substitution with the host's supernodal factor, one
 * device -> host -> device round trip per cycle.  DEVICE_INVERSE: one dense symmetric matrix-vector product with A_L^-1 on the device; the inverse
 * is built on the device from the host's factor at gmg_set_system (timing keys "coarse_inverse_ms").  AUTO (default): DEVICE_INVERSE while the
 * coarsest level has at most 8 192 unknowns (the reference's lower_bound = 1000, ratio = 8 keep it below 8 000), else HOST_LDLT. */
enum { GMG_COARSE_HOST_LDLT = 0, GMG_COARSE_DEVICE_INVERSE = 1, GMG_COARSE_AUTO = 2 };

typedef struct {
    int device;            /* HIP device ordinal */
    int smoother;          /* GMG_SMOOTHER_*: multicolour Gauss-Seidel (default) or weighted Jacobi */
    double jacobi_omega;   /* damping for GMG_SMOOTHER_JACOBI (default 0.67) */
    int pre_iters;         /* MultigridSolver::preIters  (gravomg_bindings/src/cpp/core.cpp:55) */
    int post_iters;        /* MultigridSolver::postIters (core.cpp:56) */
    int coarse_mode;       /* GMG_COARSE_*: where the coarsest direct solve is applied (default GMG_COARSE_AUTO) */
    int use_graph;         /* 1: replay the V-cycle legs from captured hipGraphs; 0 (default): plain stream launches */
    int sigma;             /* length-sorting window (rows) inside the colour classes of a colour-major level (multiple of 64; 0 = no sorting,
                              the default: sorting rows by length trades SELL padding for locality of the gathers, and on the irregular
                              workloads measured the locality is worth more -- 2 M point cloud: fine sweep 131 -> 108 us, residual 109 -> 82 us) */
    int row_align;         /* colour classes padded to this many rows (multiple of 64; default 64) */
    int block_rows;        /* block-hybrid Gauss-Seidel: rows per block (multiple of 64, <= 1024, <= 256 unless
                              block_lanes = 1; 0 = off; default 64 = one wavefront per block) */
    int block_lanes;       /* SELL lanes per row on the blocked levels and in the restriction: 1, 4 ("quad" layout: 16 rows
                              per wavefront) or 0 = automatic (default): 4 on levels with < 262144 rows (latency-bound),
                              1 on larger ones (throughput-bound) */
    int block_from_level;  /* levels >= this use the block-hybrid sweep (one launch per sweep); default 1:
                              level 0 keeps the exact multicolour sweep, the launch-bound coarse levels are blocked */
    int device_setup;      /* 1 (default): build the SELL layouts on the GPU from the uploaded matrices + orderings;
                              0: build them with the host planner (the specification the device builder is tested against) */
    int device_rap;        /* 1 (default): Galerkin products U^T A U on the GPU (bitwise equal to the host implementation, which
                              remains the fallback and the specification); needs device_setup = 1 */
    int reorder_fine;      /* locality reordering of the finest level before colouring: 0 never, 1 always, 2 (default)
                              automatic = when the input vertex order has a large bandwidth (random-order scans, point clouds) */
    int inner_precision;   /* 0 (default): everything in fp64.  1: mixed precision -- the V-cycle runs in fp32 (fp32 copies of all
                              operators) as the correction operator of an fp64 defect-correction loop: r = b - A x in fp64,
                              e = Vcycle32(r) from a zero guess, x += e.  Same iteration as the fp64 V-cycle with initial guess
                              (every stage is affine); the residual check uses the fp64 residual that feeds the next cycle */
    int block_csr;         /* 1 (default): blocked levels with one lane per row and 64-row blocks (the big ones) keep their
                              off-block operator also as a block-ordered CSR, staged through LDS by the sweep used for more
                              than one right-hand side (the padded SELL form stores 3x its real entries: a block's slice is as
                              wide as its longest row; 104 -> 60 us per sweep at d = 3 on a 506 k-row level); 0: SELL only */
    int host_threads;      /* threads for host-side setup (RAP, layout); 0 = all cores */
    int verbose;
    double gs_omega;       /* relaxation factor of the level-0 multicolour sweep: x_i <- x_i + omega (x_i^GS - x_i) (SOR).  1.0 is the
                              reference's Gauss-Seidel update (multigrid_solver.cpp:1200-1208) in colour order; the default over-relaxes
                              (see gmg_config_default) because the colour ordering smooths a little less per sweep than the
                              reference's lexicographic one (V-cycles to 1e-4 on the 3 M Poisson problem).
                              Must lie in (0, 2): SOR converges for every symmetric positive definite system in that range. */
    int restrict_sigma;    /* length-sorting window (rows) of the restriction operators U_k^T (multiple of 64; 0 = no sorting; default 64).
                              Separate from `sigma`: sorting coarse rows by length across wide windows scatters neighbouring rows
                              (their children are neighbours in the fine vector) and adds an output-row indirection -- 39.5 us with
                              1024-row windows, 31.1 us with 64-row windows on the 3 M-vertex level (profiles/README.md) */
    int block_ep;          /* 1 (default): big blocked levels (one lane per row, 64-row blocks, block_csr = 1) run the entry-parallel
                              block sweep: in-block and off-block operators as unpadded block-ordered CSR, one 64-lane gather per
                              64 ENTRIES instead of one per padded column of the block's longest row; 0: the SELL / block-CSR sweeps */
    int dist_shard_levels; /* multi-GPU (gmg_p2p_*): how many levels are partitioned over the ranks.  2 (default): level 0 by rows per
                              colour AND level 1 by blocks; 1: level 0 only (levels >= 1 replicated on every rank) */
    int block_fine;        /* 1 (default): level 0 runs the block-hybrid sweep as well (as if block_from_level were 0) when its multicolour
                              sweep would fall apart into a dozen small launches AND the block sweep is known to converge: >= 9 stored
                              entries per row on average, positive diagonal, no positive off-diagonal entry (a Stieltjes matrix: kNN
                              graph Laplacians of point clouds; 2 M points: 11 colour launches per sweep -> 1).  Triangle-mesh
                              operators (7 entries per row, 4-7 colours) keep the over-relaxed multicolour sweep, Bilaplacians fail the
                              sign test.  The blocks are runs of 64 points of the hierarchy's cluster order.  0: level 0 is blocked only
                              by block_from_level = 0.  The multi-GPU path (gmg_p2p_*) cuts such a level 0 into runs of whole blocks, one halo
                              exchange per sweep (a PARTITIONED set-up, gmg_dist_partition, keeps level 0 colour-major) */
    int fine_col16;        /* 1 (default): level 0 keeps, beside the int32 column indices, 16-bit column codes (window of the slice + offset, two to
                              a word) that its kernels read instead: 2 of an entry's 12 bytes less per launch, the same columns in the same order
                              (DESIGN.md section 3); 0: 32-bit indices only */
    int stream_gate;       /* 1 (default): the way up of a V-cycle is enqueued BEFORE the host solves the coarsest system, parked behind a stream
                              wait on a word the host writes after its back-substitution.  0: enqueue it afterwards -- for applications that issue
                              device-wide synchronisations (hipDeviceSynchronize, hipFree ...) from OTHER threads while a solve is in flight */
    int dist_exchange;     /* multi-GPU (gmg_p2p_*): how the ranks exchange.  0 (default): device-initiated stores into the peers' mailboxes (hipIpc mappings,
                              one launch per exchange).  1: the north star's collective -- pack -> ncclAllGather of the packed halo over RCCL -> unpack, three
                              stream-ordered launches per exchange enqueued by the engine, no hipIpc (gmg_p2p_connect_rccl instead of gmg_p2p_export /
                              gmg_p2p_connect).  2: the same sequence with the all-gather emulated by stores through hipIpc mappings: for ranks that
                              share one device (RCCL refuses that), i.e. for tests of the collective path on a 1-GPU box.  Same iterates in every mode */
    int prepare_structure; /* 1 (default): when the handle knows the level-0 point graph (gmg_set_fine_graph, gmg_use_hierarchy), gmg_finalize_hierarchy
                              builds everything structural for a system with that sparsity pattern -- orderings, colourings, layouts, symbolic Galerkin
                              products, symbolic LDL^T -- so that the first gmg_set_system with it only moves values.  0: the first system pays for its
                              structure like any system with an unannounced pattern */
    int fuse_restrict_sweep; /* 1 (default): the restriction into a level that runs the entry-parallel block sweep also runs that level's first pre-sweep
                              (the coarse correction starts from zero, multigrid_solver.cpp:1072-1073: its first sweep needs only the block's own right-hand
                              sides) -- one launch less per such level and cycle, the same bits; 0: two launches */
    int speculate_head;    /* 1 (default): the solve loop (multigrid_solver.cpp:1408-1419: cycle, residual check, decision) enqueues the FIRST colour launch
                              of the next cycle behind the check before the host has seen the norm; the check's reduction takes the loop's decision on the
                              device (same sums, same arithmetic) and that launch returns at once when the iteration has stopped -- the ~6 us of host latency
                              between two cycles disappear, iterates and iteration counts are unchanged.  Stream launches, fp64, colour-major level 0, d <= 4,
                              one device; otherwise, and with 0, the host decides before anything of the next cycle is enqueued */
    int uniform_slices;    /* 1 (default): a level-0 operator all of whose 64-row slices are equally wide (a regular mesh: every vertex has six neighbours) is
                              read without its slice pointers -- a slice's place follows from its number, so a wave's first loads are the entries and not two
                              pointers they would wait for.  Same entries in the same order.  0: always through the pointers */
    int color_ahead;       /* 1 (default): a gmg_set_system that cannot be a values-only refresh starts the greedy colouring of level 0 (one core, 10-13 ms at
                              3 M vertices: the longest task of a cold set-up) at entry, beside the inspection of the caller's arrays, with every index checked;
                              the result is used when the inspection and the layout decisions allow it -- the same colours, ~6 ms earlier.  0: after them */
} gmg_config;

/* ---- lifetime ------------------------------------------------------------------------------- */
int gmg_config_default(gmg_config* cfg);
/* sizeof(gmg_config) as the LIBRARY was built: a binding compiled against another edition of this header (the struct grows at its end) compares it
 * with its own before it hands a gmg_config over -- the pybind module and the ctypes mirror refuse to run on a mismatch instead of passing a
 * misread configuration. */
int gmg_config_size(void);
/* Replaces the construction of the solver state the pybind shim owns
 * (gravomg_bindings/src/cpp/core.cpp:27,138). */
int gmg_create(const gmg_config* cfg, gmg_handle* out);
void gmg_destroy(gmg_handle h);
const char* gmg_last_error(gmg_handle h);
/* Number of usable HIP devices (0 on a CPU-only box); never fails. */
/* Host threads a handle uses unless gmg_config::host_threads says otherwise: the CPUs this PROCESS may use (hardware threads,
 * affinity mask, cgroup v1/v2 quota; GMG_HOST_THREADS overrides) divided by LOCAL_WORLD_SIZE when one process per GPU shares a
 * node -- worker pool, staging copies and the coarsest back-substitution's team are sized by it. */
int gmg_host_threads(void);
int gmg_device_count(void);

/* ---- hierarchy input ------------------------------------------------------------------------ */
/* Declare how many transfer levels follow (L = U.size()).  Drops any previous hierarchy/system. */
int gmg_set_num_levels(gmg_handle h, int L);
/* U[k], n_k x n_{k+1}, CSC.  Replaces the `U` member / set_prolongation_matrices
 * (gravomg/include/gravomg/multigrid_solver.h:106, gravomg_bindings/src/cpp/core.cpp:86-88). */
int gmg_set_prolongation(gmg_handle h, int k, int n_fine, int n_coarse, const int* colptr, const int* rowidx, const double* val);
/* Lumped mass diagonal M (n_0 entries) used by the residual norms, multigrid_solver.h:96-97. */
int gmg_set_mass(gmg_handle h, int n, const double* mass_diag);
/* LHS for the next solves.  Performs what solve() does before its loop (multigrid_solver.cpp:1387-1401):
 * Galerkin products Abar[k] = U[k-1]^T Abar[k-1] U[k-1], coarsest LDL^T factorisation, plus the device
 * layout (colouring, SELL) and the upload.  Timings land in "reduction", "coarsest_solve", "upload".
 * The reference recomputes all of this on every solve(); here a matrix with the sparsity pattern of the live system
 * (recognised by a 128-bit digest of colptr/rowidx) only refreshes values in place -- numeric Galerkin passes, layout
 * refill, numeric LDL^T; timing key "setup_values_only" = 1 -- and a matrix with a pattern seen before on this handle
 * reuses the orderings ("setup_ordering_cached" = 1).  Results are those of a fresh handle in every case.
 * A call that FAILS (GMG_ERR_INVALID for an index out of range, GMG_ERR_NUMERIC for a zero diagonal / pivot, ...) leaves the handle WITHOUT a
 * system: a matrix of the live system's size has its values uploaded over the resident A_0 -- and the refresh started -- while the pattern is
 * still being inspected, so the previous system does not survive a rejected call; the next solve returns GMG_ERR_STATE until a gmg_set_system
 * succeeds (the reference has no persistent system either: solve() receives the LHS every time, multigrid_solver.cpp:1367). */
int gmg_set_system(gmg_handle h, int n, const int* colptr, const int* rowidx, const double* val);

/* ---- introspection -------------------------------------------------------------------------- */
int gmg_num_levels(gmg_handle h);                       /* L, or < 0 */
/* n_k, nnz(A_k), number of colours, padded device length of level k (0 <= k <= L). */
int gmg_level_info(gmg_handle h, int k, int* n, int64_t* nnz, int* n_colors, int* n_pad);
/* Copy out A_k (CSC).  k = 0 is the LHS, k >= 1 the Galerkin operators (the reference's Abar[k]). */
int gmg_get_level_operator(gmg_handle h, int k, int* colptr, int* rowidx, double* val);
/* Device numbering of level k: new2old[n_pad] (-1 = padding row), color_begin[n_colors+1]. */
int gmg_get_level_ordering(gmg_handle h, int k, int* new2old, int* color_begin);
/* Blocked levels (block-hybrid Gauss-Seidel): *n_blocks (0 if the level is colour-major), blk_begin[n_blocks+1]
 * (device rows) and row_color[n_pad] (colour of a row inside its block).  Output pointers may be NULL. */
int gmg_get_level_blocks(gmg_handle h, int k, int* n_blocks, int* blk_begin, unsigned char* row_color);
/* Named timers in ms, same keys as the reference's solverTiming (multigrid_solver.cpp:1394,1403,1445-1448):
 * "reduction", "coarsest_solve", "cycles", "solver_total", "iterations", "residue"; plus "upload",
 * "coarse_host_ms" (host back-substitutions inside the cycles). */
int gmg_get_timing(gmg_handle h, const char* key, double* out);

/* ---- operators (host in / host out; natural numbering).  Used by the parity tests -------------- */
/* `iters` smoothing sweeps on level k: replaces GaussSeidelSmoother, multigrid_solver.cpp:1194-1226. */
int gmg_smooth(gmg_handle h, int k, const double* b, double* x, int d, int iters);
/* The first half of the way down on level k as the cycle runs it: `iters` smoothing sweeps (multigrid_solver.cpp:1063; from_zero
 * != 0: from the zero vector, :1072-1073, x is output only) followed by res = b - A x (:1066).  On a level with the unpadded block
 * storage the residual comes out of the last sweep's explicit part (r_i = sum_E a_ij (x_old_j - x_new_j), csrc/kernels.hip.hpp::
 * residual_delta_ep; timing key "residual_from_sweep" = 1), otherwise from the residual SpMV: the entry point of the parity test
 * of that shortcut against b - A x. */
int gmg_smooth_residual(gmg_handle h, int k, const double* b, double* x, int d, int iters, int from_zero, double* r);
/* r = b - A_k x : multigrid_solver.cpp:1066. */
int gmg_residual(gmg_handle h, int k, const double* b, const double* x, int d, double* r);
/* y = A_k x. */
int gmg_spmv(gmg_handle h, int k, const double* x, int d, double* y);
/* rc = U_k^T r : multigrid_solver.cpp:1069. */
int gmg_restrict(gmg_handle h, int k, const double* r, int d, double* rc);
/* x += U_k e : multigrid_solver.cpp:1082. */
int gmg_prolong_add(gmg_handle h, int k, const double* e, int d, double* x);
/* e = A_L^{-1} rc with the configured coarse solver: multigrid_solver.cpp:1075. */
int gmg_coarse_solve(gmg_handle h, const double* rc, int d, double* e);
/* residualCheck(A_0, b, x, type), multigrid_solver.cpp:1228-1277 (types 0..3). */
int gmg_residual_norm(gmg_handle h, const double* b, const double* x, int d, int type, double* out);

/* ---- the hot path --------------------------------------------------------------------------- */
/* One V-cycle on level 0, x updated in place: multiGridVCycleGS, multigrid_solver.cpp:1059-1088. */
int gmg_vcycle(gmg_handle h, const double* b, double* x, int d);
/* The MG branch of solve(), multigrid_solver.cpp:1408-1419: do { V-cycle; residualCheck } while
 * (residue > tol && it < max_iter).  x arrives holding the initial guess (the binding passes x0 = rhs,
 * gravomg_bindings/src/cpp/core.cpp:69).  conv (optional) receives (elapsed_ms, residue) pairs and must
 * hold 2*max_iter doubles.  Like the reference, x ALWAYS receives the last iterate.  The return value tells a caller what the
 * reference leaves it to find out: GMG_OK when the loop ended the way the reference's does (residue <= tol, or max_iter reached while
 * the residue was at or below the one after the first cycle), GMG_DIVERGED (= 1, not an error; gmg_get_timing(h, "diverged") is 1 too)
 * when it ended above tol with a residue that is not finite or larger than after the first cycle.  The loop also stops early -- it
 * cannot recover -- at the first non-finite residue and once the residue exceeds 1e4 x the smallest one seen (iters_out then is
 * < max_iter).  The default smoothers are parallel orderings / block variants of the reference's Gauss-Seidel without its
 * convergence guarantee; a handle created with block_rows = 0 and gs_omega = 1 runs Gauss-Seidel in colour order on every level
 * (convergent for every SPD matrix) -- the retry the C++ mirror (MGBS::MultigridSolver::solve) performs by itself from the same
 * initial guess. */
int gmg_solve(gmg_handle h, const double* rhs, double* x, int d, double tol, int stop_type, int max_iter,
              int* iters_out, double* residue_out, double* conv);
/* The same with the initial guess x0 = rhs, the only one the reference's Python binding ever passes (core.cpp:69): x is OUTPUT only --
 * the caller does not fill it with a copy of rhs, and the engine copies rhs to x on the device instead of comparing and uploading. */
int gmg_solve_x0_rhs(gmg_handle h, const double* rhs, double* x, int d, double tol, int stop_type, int max_iter,
                     int* iters_out, double* residue_out, double* conv);

/* ---- resident problem (device-resident b / x; what gmg_solve and bench.py are built from) ---------- */
/* Upload rhs and the initial guess (natural numbering, n_0 x d) and keep them resident. */
int gmg_load_problem(gmg_handle h, const double* b, const double* x0, int d);
/* Run n_cycles V-cycles on the resident problem.  stop_type >= 0: each cycle is followed by the residual
 * check of that type exactly as the solve loop does (residues[i] receives it; residues may be NULL);
 * stop_type < 0: cycles only.  Returns after the stream has drained. */
int gmg_run_cycles(gmg_handle h, int n_cycles, int stop_type, double* residues);
/* Download the resident iterate (natural numbering). */
int gmg_fetch_solution(gmg_handle h, double* x);

/* ---- multi-GPU: one process per GPU ---------------------------------------------------------- */
/* The finest level is split by rows: every colour class (padded to 64*world rows: create the handle with
 * row_align = 64*world) is cut into `world` equal contiguous pieces and rank p owns piece p of each colour.
 * Levels >= 1 are replicated on every rank.  The caller owns the level-0 vectors (device memory, device
 * numbering, n_pad_0 x d column-major) and exchanges them between the steps below (gravo_mg_amd/dist.py: RCCL
 * all-gather of a colour's segment of x after each colour, all-reduce of the norm sums); because the colours are
 * global the result does not depend on the number of ranks. */
/* Run all launches on this HIP stream (e.g. torch's current stream) instead of the handle's own; NULL restores it. */
int gmg_set_stream(gmg_handle h, void* hip_stream);
int gmg_dist_setup(gmg_handle h, int rank, int world);
/* Partitioned SET-UP (the work split of multigrid_solver.cpp:1059-1088 applied to its preamble :1387-1403).  Call before
 * gmg_set_system (and before gmg_finalize_hierarchy / gmg_use_hierarchy) on a handle created with row_align = 64 * world: this handle is rank
 * `rank` of `world` ranks of a row-partitioned job, and gmg_set_system then lays out, and keeps, only this rank's rows of level 0 (its piece of
 * every colour class) and of level 1 (its blocks, gmg_config::dist_shard_levels = 2) -- operator, block-CSR parts, prolongation, restriction,
 * 16-bit column codes -- in the GLOBAL device numbering, so that the cycle's kernels run unchanged on this rank's slices and blocks; levels
 * >= 2 and all vectors stay whole.  The Galerkin chain still sees the whole fine operator (the replicated coarse levels need every row of A_1
 * and there is no exchange during the set-up): A_0, A_1 and U_0 are uploaded, used and RELEASED, so the memory a rank holds afterwards is its
 * share plus the replicated small levels (gmg_p2p_stat "device_bytes"; "device_bytes_peak" is the high-water mark of the set-up).  The partition
 * plan (who publishes what to whom) is made during the set-up, while the patterns are at hand, and kept with the orderings under the pattern
 * digest.  Only the gmg_p2p_* / gmg_dist_* entry points run on such a handle; a system with the live pattern is set up again from scratch
 * (no values-only refresh: there is no whole operator to refresh).  world = 1 returns to whole systems. */
int gmg_dist_partition(gmg_handle h, int rank, int world);
/* Use caller-owned device buffers as level-0 x, b, r (until the next gmg_set_system / a larger d). */
int gmg_dist_bind(gmg_handle h, double* x0, double* b0, double* r0, int d);
/* Colour c of one Gauss-Seidel sweep on this rank's rows (multigrid_solver.cpp:1194-1226, row-partitioned).  GMG_ERR_STATE on a level 0 that runs the
 * block sweep (gmg_config::block_fine): such a level is ONE class of rows cut into `world` runs of whole 64-row blocks, swept block-wise with one halo
 * exchange per sweep by the engine-driven cycle (gmg_p2p_cycles); the step-by-step entry points below serve colour-major levels. */
int gmg_dist_smooth_color(gmg_handle h, int c);
/* r0[own rows] = b0 - A x0 (:1066). */
int gmg_dist_residual_own(gmg_handle h);
/* Replicated coarse part of the V-cycle: U0^T r0 (complete r0 required), levels 1..L-1, coarsest solve (:1069-1079). */
int gmg_dist_coarse_cycle(gmg_handle h);
/* x0[own rows] += U0 x1 (:1082). */
int gmg_dist_prolong_own(gmg_handle h);
/* This rank's share of the residual-norm sums: sums[2c] = sum w r^2, sums[2c+1] = sum w b^2 for rhs column c. */
int gmg_dist_norm_partial(gmg_handle h, int type, double* sums);
/* on != 0: the three steps above cover ALL rows of level 0 until switched off again.  After the exchange that follows every colour sweep each rank
 * holds the complete x0, so residual, prolongation and norm sums can be computed redundantly instead of being exchanged (no collective for r0,
 * for the prolongated x0 or for the norm sums, which then are identical on every rank).  Not on a partitioned handle. */
int gmg_dist_all_rows(gmg_handle h, int on);
/* Halo exchange helpers (gravo_mg_amd/dist.py, `halo` mode: after a colour sweep a rank publishes only the entries of
 * x0 that rows of other ranks read).  dst[i] = src[idx[i]] and dst[idx[i]] = src[pos[i]] for i < n, on the handle's
 * stream; every pointer is a device pointer. */
int gmg_dist_gather(gmg_handle h, const double* src, const int64_t* idx, int64_t n, double* dst);
int gmg_dist_scatter(gmg_handle h, const double* src, const int64_t* pos, const int64_t* idx, int64_t n, double* dst);

/* ---- multi-GPU, engine-driven: one process per GPU, device-initiated peer-to-peer exchanges ----------------------------
 * The same row-partitioned V-cycle (multigrid_solver.cpp:1059-1088 + the residual check :1228-1277), but orchestrated inside
 * the library: every exchange is one small kernel that stores this rank's halo values straight into the peers' mailboxes
 * (fine-grained device memory mapped through hipIpc handles, xGMI stores) and waits for theirs -- no collective-library call
 * and no host code between the launches of a cycle.  Call order, on every rank: gmg_set_system on a handle created with
 * row_align = 64 * world -> gmg_p2p_prepare -> gmg_p2p_export -> (the caller gathers the `world` blobs in rank order by any
 * means: torch.distributed.all_gather_object, MPI, a file) -> gmg_p2p_connect -> gmg_p2p_load -> gmg_p2p_cycles ... ->
 * gmg_p2p_fetch.  All ranks must run the same sequence of gmg_p2p_cycles / gmg_p2p_fetch calls; a missing peer shows up as
 * GMG_ERR_STATE after a ~4 s device-side time-out, never as a hang.  Iterates are bitwise those of one GPU. */
int gmg_p2p_blob_bytes(void);
int gmg_p2p_prepare(gmg_handle h, int rank, int world, int d);
int gmg_p2p_export(gmg_handle h, void* blob_out);
/* blobs: world x gmg_p2p_blob_bytes() bytes in rank order.  GMG_ERR_STATE when two ranks of the job sit on the same device (the blobs carry
 * the device's UUID): their exchange kernels would wait for each other on one GPU -- a mis-mapped HIP_VISIBLE_DEVICES then is an error at
 * connect time instead of seconds per exchange.  GMG_P2P_SHARED_DEVICE=1 in the environment allows it (functional tests on a one-GPU box). */
int gmg_p2p_connect(gmg_handle h, const void* blobs);
/* gmg_config::dist_exchange = 1: instead of gmg_p2p_export / gmg_p2p_connect.  Rank 0 makes the 128-byte RCCL id (ncclGetUniqueId), the caller gives
 * it to every rank by any means, every rank calls gmg_p2p_connect_rccl (ncclCommInitRank: collective).  GMG_ERR_UNSUPPORTED without librccl. */
int gmg_p2p_rccl_unique_id(void* id_out_128_bytes);
int gmg_p2p_connect_rccl(gmg_handle h, const void* id_128_bytes);
int gmg_p2p_load(gmg_handle h, const double* b, const double* x0);
int gmg_p2p_cycles(gmg_handle h, int n_cycles, int stop_type, double* residues);
int gmg_p2p_fetch(gmg_handle h, double* x);
/* gmg_solve as a collective: load (x = initial guess, n x d column-major like b), the reference's do-while loop
 * (multigrid_solver.cpp:1408-1419) of gmg_p2p_cycles(1) until residue <= tol or max_iter cycles, fetch into x.  The residues are
 * identical on all ranks, so all ranks stop in the same iteration.  This is what gravomg.MultigridSolver.solve() runs after
 * enable_distributed() (gravo_mg_amd/dropin/gravomg/core.py). */
int gmg_p2p_solve(gmg_handle h, const double* b, double* x, double tol, int stop_type, int max_iter, int* iters_out, double* residue_out);
int gmg_p2p_stat(gmg_handle h, const char* key, double* out);
/* Level-0 smoother of the partitioned cycle (multigrid_solver.cpp:1194-1226 split over ranks).  0 (default): multicolour
 * Gauss-Seidel with an exchange after every colour -- the single-GPU iterates bit for bit, (pre + post) x colours exchanges per cycle;
 * 1: hybrid -- Gauss-Seidel inside a rank, Jacobi across ranks, ONE exchange per sweep (pre + post per cycle); the iterates then depend
 * on the number of ranks;  2: as 0 -- the same iterates -- with the exchange of a colour folded into that colour's sweep launch (mailbox
 * backend: publishing waves store their rows into the peers' mailboxes, the last one pulls; no exchange launch for the colour halos).
 * Every rank must make the same choice. */
int gmg_p2p_set_smoother(gmg_handle h, int mode);
/* Memory ordering of the mailbox exchanges (replaces nothing in the reference: the cost of splitting multigrid_solver.cpp:1194-1226 over devices).
 * 1 (default): the HIP memory model's publication idiom -- a system-scope release fence between the halo stores and the store of the sequence
 * number, an acquire fence after the poll that saw it; portable, and ~3.5 us per exchange launch for writing the L2 back.  0: the gfx942 /
 * gfx950 form without the cache write-back (write-through system-scope stores, drained with s_waitcnt vmcnt(0) ahead of the number, system-scope
 * loads on the other side; csrc/kernels.hip.hpp::publish_order) -- opt-in: a caller that takes it should check the first cycles against a run it
 * trusts (bench.py --gpus N does: the single-GPU residues, and falls back to the fenced form).  Default of new plans: GMG_P2P_FENCE_FREE=1 makes
 * it 0.  Every rank must make the same choice (a mixed job is still ordered correctly -- each side fences or not for itself). */
int gmg_p2p_set_fences(gmg_handle h, int fenced);

/* ---- host-only: hierarchy construction (no device needed) ----------------------------------- */
typedef struct {
    double ratio;          /* core.py:10 ratio=8.0 */
    int lower_bound;       /* lower_bound=1000 */
    int check_voronoi;     /* 1 */
    int nested;            /* 0 */
    int sampling;          /* Sampling enum (multigrid_solver.h:40-46); only 0 = FASTDISK is supported */
    int weighting;         /* Weighting enum (multigrid_solver.h:48-52): 0 BARYCENTRIC, 1 UNIFORM, 2 INVDIST */
    int debug;             /* the reference's `debug` member: keep every level's candidate triangles (allTriangles, multigrid_solver.cpp:281) */
    int full_clustering;   /* 0 (default): the Dijkstra clustering sweep (multigrid_solver.cpp:1015-1056) is replaced by what it provably does after the
                              FASTDISK sampler -- resetting the samples (csrc/host_hierarchy.hpp::voronoi_dijkstra); 1: run the sweep as written
                              (same hierarchy bit for bit; the cross-check of tests/test_hierarchy_restatement.py) */
    int use_device;        /* 1 (default): with a HIP device the per-point parent selection (multigrid_solver.cpp:291-452) of levels with >= 200 000 points runs
                              on it -- same prolongations, bit for bit; 0: host only */
} gmg_hierarchy_options;

int gmg_hierarchy_options_default(gmg_hierarchy_options* o);
/* Replaces MGBS::MultigridSolver::buildHierarchy / constructProlongation
 * (gravomg/src/multigrid_solver.cpp:43-60, 62-469).  pos: n x 3 row-major; neigh: n x K row-major,
 * padded with -1 (gravomg_bindings/src/cpp/core.cpp:15-18).  Works without a GPU; with one, the per-point parent selection
 * (:291-452) of levels with >= 200 000 points runs on it -- same prolongations, bit for bit (gmg_hierarchy_options::use_device = 0: host only). */
int gmg_hierarchy_build(const double* pos, int n, const int* neigh, int K, const gmg_hierarchy_options* opt,
                        gmg_hierarchy* out);
void gmg_hierarchy_destroy(gmg_hierarchy hh);
int gmg_hierarchy_num_levels(gmg_hierarchy hh);                            /* U.size() */
int gmg_hierarchy_level_shape(gmg_hierarchy hh, int k, int* n_fine, int* n_coarse, int* nnz);
int gmg_hierarchy_get_prolongation(gmg_hierarchy hh, int k, int* colptr, int* rowidx, double* val);
/* hierarchyTiming keys of the reference (multigrid_solver.cpp:21,57,90-97). */
int gmg_hierarchy_get_timing(gmg_hierarchy hh, const char* key, double* out);
/* What the reference keeps beside U after buildHierarchy (multigrid_solver.h:99-104; getters of the pybind class,
 * gravomg_bindings/src/cpp/core.cpp:90-116): samples[k] = fine index of every coarse point of level k+1 (n_{k+1} ints),
 * nearest[k] = the cluster (coarse point) of every point of level k (n_k ints; `nearestSource`), points[k] = positions of the
 * coarse points of level k+1 (n_{k+1} x 3, row-major; `levelV`, which the reference only fills with debug = true). */
int gmg_hierarchy_get_samples(gmg_hierarchy hh, int k, int* out);
int gmg_hierarchy_get_nearest(gmg_hierarchy hh, int k, int* out);
int gmg_hierarchy_get_points(gmg_hierarchy hh, int k, double* out_xyz);
/* allTriangles[k] (multigrid_solver.h:101; only kept when the hierarchy was built with debug != 0): the candidate triangles of the
 * coarse points of level k+1, *count triples of coarse indices, in construction order.  out may be NULL (size query). */
int gmg_hierarchy_get_triangles(gmg_hierarchy hh, int k, int* out, int* count);
/* A breadth-first order of the level-0 points over `neigh` (new -> old), made beside the construction when the input numbering
 * has no locality (randomly ordered scans, point clouds); *count = 0 otherwise.  out may be NULL (size query).  Not in the
 * reference: a by-product the MI355X engine uses to number the finest level (gmg_set_fine_order). */
int gmg_hierarchy_get_fine_order(gmg_hierarchy hh, int* out, int* count);
/* Optional, after the prolongations and before gmg_finalize_hierarchy / gmg_set_system: a locality-preserving order of the
 * level-0 points (a permutation, new -> old; n = 0 clears it).  When the system's numbering has no locality the engine
 * renumbers its finest level; with this order at hand it scores it against the order it derives from the hierarchy on the
 * actual matrix and uses the better one (results do not depend on the numbering beyond rounding).  gmg_use_hierarchy passes
 * the hierarchy object's order by itself. */
int gmg_set_fine_order(gmg_handle h, int n, const int* order);
/* Optional, after the prolongations and before gmg_finalize_hierarchy: the level-0 point graph -- the `neigh` table the hierarchy was built from
 * (n x K row-major, padded with -1; gravomg_bindings/src/cpp/core.cpp:15-18).  The systems a hierarchy is built for (tau M + S, M + tau S of that
 * mesh or point cloud: experiments/python/comparisons.py:75-78, demos/smoothing.py:43-47) have exactly this graph plus the diagonal as their
 * sparsity pattern, so the engine can do at hierarchy time what the reference redoes in every solve() preamble although it depends on the pattern
 * only (gmg_config::prepare_structure).  A system with another pattern (a Bilaplacian's two-ring) simply takes the cold path.  n = 0 clears it. */
int gmg_set_fine_graph(gmg_handle h, int n, int K, const int* neigh);
/* Convenience: feed every U_k of a built hierarchy into a solver handle, its fine order and point graph with them (and finalize it, see below). */
int gmg_use_hierarchy(gmg_handle h, gmg_hierarchy hh);
/* Optional, after the last gmg_set_prolongation: build what depends on the hierarchy only (the reference's
 * buildHierarchy phase, multigrid_solver.cpp:15-60) -- the device copies of U_k, the compact row patches of the
 * block-hybrid smoother's levels and, with a fine graph, the structure of the systems to come (gmg_config::prepare_structure;
 * timing key "structure_prepare_ms") -- now instead of inside the first gmg_set_system.  Idempotent. */
int gmg_finalize_hierarchy(gmg_handle h);


/* Host-only: x = A^{-1} b with the coarsest-level solver (minimum-degree + sparse LDL^T, host_ldlt.hpp),
 * b/x column-major n x d.  Returns GMG_ERR_NUMERIC on a zero pivot.  factor_nnz (optional) = nnz(L). */
int gmg_host_ldlt_solve(int n, const int* colptr, const int* rowidx, const double* val, const double* b, int d, double* x,
                        int64_t* factor_nnz);
#ifdef __cplusplus
}
#endif
#endif /* GRAVOMG_HIP_H */
