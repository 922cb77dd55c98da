/*
 * gravomg_hip_internal.h -- test and measurement hooks of libgravomg_hip.so.  NOT part of the drop-in boundary (include/gravomg_hip.h):
 * nothing a reference maintainer would bind.  tests/ and scripts/ reach the host-side building blocks (Galerkin product, layout planner,
 * LDL^T probe), the device-resident layouts and two fault-injection knobs of the set-up through these symbols.
 */
#ifndef GRAVOMG_HIP_INTERNAL_H
#define GRAVOMG_HIP_INTERNAL_H

#include "gravomg_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test access to the device-resident SELL layouts of level k: which = 0 A (off-diagonal part), 1 A_in, 2 A_out,
 * 3 P (U_k), 4 R (U_k^T).  info[0..3] = n_slices, lanes per row, stored entries, has row_of.  Copy-out pointers may be
 * NULL; col receives 32-bit columns also for the 16-bit A_in; diag (which = 0 only) the level's diagonal. */
int gmg_debug_sell_info(gmg_handle h, int k, int which, int64_t* info);
int gmg_debug_sell_copy(gmg_handle h, int k, int which, int64_t* slice_ptr, int* col, double* val, int* row_of, double* diag);

/* Set-up fault injection for the tests, per handle, effective from the next gmg_set_system.  Keys:
 *   "col16_uncovered" = N > 0: every N-th level-0 slice is treated as not covered by its column windows (flagged one by one, c16 mode 2);
 *                       N < 0: the first -N slices (a prefix, c16 mode 1); 0: off.  Results must not change (tests/test_gpu_setup.py). */
int gmg_debug_set(gmg_handle h, const char* key, double value);

/* Host-only Galerkin product Ac = U^T A U (CSC in / CSC out, caller sizes the output with the first call:
 * pass colptr_out only to get nnz in colptr_out[n_coarse]).  Exposed for the RAP parity tests. */
int gmg_host_galerkin(int n, const int* a_colptr, const int* a_rowidx, const double* a_val,
                      int n_coarse, const int* u_colptr, const int* u_rowidx, const double* u_val,
                      int* c_colptr, int* c_rowidx, double* c_val);

/* Host-only view of the device layout planner (colouring / block growing / SELL-64), for CPU tests of the
 * host logic.  mode 0: colour-major ordering (exact multicolour Gauss-Seidel), mode 1: block ordering
 * (block-hybrid Gauss-Seidel, `block_rows` rows per block), mode 2: mode 0 with the locality reordering forced (rows of a colour in breadth-first
 * patch order), mode 3: mode 0 for a matrix whose row indices are ascending (the short-cut of the colouring loop), mode 4: mode 3 with the colouring that a cold
 * gmg_set_system starts ahead of its inspection -- on the arrays as given, every index checked (GMG_ERR_INVALID for arrays that would take a reader out of bounds;
 * info[5] = 1 when that colouring was used, 0 when it gave up at 64 colours; no SELL statistics).  In modes 0 / 2 / 3 / 4 `block_rows` is the colour-class
 * alignment (the config's row_align; 0 = 64).  info[0..5] = n_pad, n_colors, n_blocks,
 * stored SELL entries (off-diagonal), real off-diagonal entries, 0.  Output pointers may be NULL: call once
 * with NULL outputs for the sizes, then with new2old / row_color of n_pad entries, color_begin of
 * n_colors + 1 and blk_begin of n_blocks + 1 entries. */
int gmg_host_plan_level(int n, const int* colptr, const int* rowidx, const double* val, int mode, int block_rows, int sigma,
                        int64_t* info, int* new2old, int* color_begin, int* blk_begin, unsigned char* row_color);

/* Host-only view of the rule behind gmg_config::block_fine (engine_setup.hip.hpp::fine_level_blocked), for CPU tests: would level 0 of this
 * system run the block-hybrid sweep under the DEFAULT configuration, given a hierarchy?  *blocked = 1: at least 9 stored entries per row on
 * average, every diagonal entry positive, no positive off-diagonal entry (a Stieltjes matrix, for the symmetric positive definite systems
 * the solver takes: the block sweep is a regular splitting).  reason (optional): 0 chosen, 1 rows too short, 2 signs. */
int gmg_host_fine_block_rule(int n, const int* colptr, const int* rowidx, const double* val, int* blocked, int* reason);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Average duration (ms) of one unit of level-k work, measured with HIP events on the engine stream:
 * kind 0 = full smoothing sweep (all colours), 1 = residual r=b-Ax, 2 = restrict, 3 = prolong_add,
 * 4 = residual-norm kernels.  The repetitions are enqueued back to back between two events (the way the
 * V-cycle issues them); launches_out = kernel launches per repetition (colours for the sweep). */
int gmg_bench_kernel(gmg_handle h, int kind, int k, int d, int reps, double* ms_avg, int* launches_out);
/* Leg-by-leg time of a V-cycle + residual check on the resident problem (HIP events at the leg boundaries, average over `reps` cycles):
 * ms_out[k], k < levels: level k's share (multigrid_solver.cpp:1063-1069 on the way down, :1082-1085 on the way up); ms_out[levels]: the
 * coarsest solve (:1075) with its host round trip; ms_out[levels + 1]: the residual check (:1228-1277).  n_out >= levels + 2. */
int gmg_profile_cycle(gmg_handle h, int stop_type, int reps, double* ms_out, int n_out);
/* Algorithmic (compulsory) bytes of the same unit of work, SURVEY.md 8(d). */
int gmg_algorithmic_bytes(gmg_handle h, int kind, int k, int d, double* bytes_out);

/* average duration (ms) of one exchange of the cycle, `reps` back to back (collective; measurement): "color<k>", "halo_all", "rows0" (every rank's level-0 rows: what a cycle with level 1
 * replicated moves once), and with level 1 partitioned "x1_halo" (after every level-1 sweep), "rows1" (r1 to all, once per
 * cycle), "r0_halo" (before the restriction, once per cycle).  Overwrites halo entries: gmg_p2p_load afterwards. */
int gmg_p2p_bench_kind(gmg_handle h, const char* kind, int reps, double* ms_avg);

/* Collective backend (gmg_config::dist_exchange != 0), after gmg_p2p_load: one exchange of every rank's level-0 rows of x (pack -> all-gather ->
 * unpack), then this rank's slot of the gathered buffer against what it packed.  max_abs_diff must be 0; doubles = values compared.  With
 * dist_exchange = 1 this is ncclAllGather's delivery seen from the host -- also on ONE rank (RCCL takes a one-rank communicator). */
int gmg_p2p_debug_collective_roundtrip(gmg_handle h, double* max_abs_diff, long long* doubles);

/* Host-only probe of the coarsest-level solver (csrc/host_ldlt.hpp): factorises A, times the back-substitution on 1 .. 8 threads (`reps` solves
 * per batch, best of 20) and the numeric re-factorisation, compares the team solves with the one-thread solve bit for bit and the supernodal
 * factor with the simplicial one.  The report (text lines) goes to `report` (cap bytes, NUL-terminated). */
int gmg_host_ldlt_probe(int n, const int* colptr, const int* rowidx, const double* val, const double* b, int reps, char* report, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GRAVOMG_HIP_INTERNAL_H */
