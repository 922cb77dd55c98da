// kernels.hip.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the V-cycle hot path.
//
// All sparse operators are SELL-64 (host_plan.hpp): one slice = 64 rows = one wavefront, one row per
// lane, entries j-major inside the slice.  A wave's j-th load of col[] is 256 contiguous bytes and of
// val[] 512 contiguous bytes (perfectly coalesced streams); the only irregular access is the gather of
// x[col].  Each lane accumulates its row sequentially in stored (ascending device column) order, so the
// result is deterministic and independent of the launch geometry.  These are HBM-bound irregular sparse
// contractions (0.17-0.25 flop/byte): no MFMA -- the slice pointer is wave-uniform (scalar loads), x lives in
// L2 / Infinity Cache (24 MB at 3 M unknowns), matrices stream once per launch.  Staging x tiles of the fine level in LDS
// (the north star's suggestion) was measured not to pay: PMC shows HBM traffic within 5-16 % of the compulsory bytes in
// every vertex ordering -- the gathers are served by L2, what separates orderings is cache lines per gather instruction,
// which a staging pass would touch just the same (profiles/README.md, round 3).  What did pay on the fine level is FEWER
// bytes: 16-bit column codes, below.  LDS is used where rows of a block depend on each other (block sweeps, levels >= 1).
//
// Dense multi-vectors: column-major, leading dimension ld (= padded level size), D columns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmgk {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kWavesPerBlock = kBlock / 64;

// Slice handled by this wave.  With xcd_swizzle the grid is re-mapped so that each XCD (block b runs on
// XCD b % 8, MI355X_MICROARCH.md "Workgroup dispatch") walks one contiguous eighth of the slice range:
// neighbouring slices gather neighbouring x entries, which then hit the same per-XCD L2.
__device__ __forceinline__ int wave_slice(int n_slices_total, int xcd_swizzle) {
    int nblk = gridDim.x;
    int b = blockIdx.x;
    if (xcd_swizzle && (nblk & 7) == 0) b = (b & 7) * (nblk >> 3) + (b >> 3);
    int w = b * kWavesPerBlock + (threadIdx.x >> 6);
    (void)n_slices_total;
    return __builtin_amdgcn_readfirstlane(w);
}

// W entries of this lane's row, all loads issued before the first use: W column loads + W value loads in flight,
// then W*D gathers in flight, then the FMAs in stored order (the accumulation order is that of a plain loop).
// XI = 1: x is an INTERLEAVED multi-vector (row-major n x D: the D values of a row are 8 D contiguous bytes -- one cache line per gathered
// row instead of D; round 4, level-0 restriction / prolongation at d > 1), XI = 0: column-major with leading dimension ld.
template <int D, int XI> __device__ __forceinline__ int64_t x_at(int c, int d, int ld) { return XI ? (int64_t)c * D + d : c + (int64_t)d * ld; }

template <class T, int D, int W, int XI = 0>
__device__ __forceinline__ void row_dot_group(const int* __restrict__ cp, const T* __restrict__ vp, const T* x, int ld,
                                              T (&acc)[D]) {
    int c[W];
    T v[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { c[j] = __builtin_nontemporal_load(cp + j * 64); v[j] = __builtin_nontemporal_load(vp + j * 64); }
    T xv[W][D];
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int d = 0; d < D; ++d) xv[j][d] = x[x_at<D, XI>(c[j], d, ld)];
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] += v[j] * xv[j][d];
}

// acc[c] = sum_j val[j] * x[col[j] + c*ld] over this lane's row of slice s.  The slice width is wave-uniform, so
// the dispatch on it is a scalar branch: full groups of 8, then one width-specialised tail (no serialized
// remainder loop -- with 6-7 entries per mesh row and 3 per prolongation row the tail IS the row).
// (G entries per group: 8 with one or two right-hand sides; 6 with three -- a mesh row's six entries are one group, 90 VGPRs, 5 wavefronts
// per SIMD: 22.9 us per colour launch against 23.5 with groups of 4 and 8 wavefronts --; 4 with four)
template <int D> struct DotGroup { static constexpr int value = D >= 4 ? 4 : (D == 3 ? 6 : 8); };
template <class T, int D, int G = DotGroup<D>::value, int XI = 0>
__device__ __forceinline__ void row_dot(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                        const T* __restrict__ val, const T* x, int ld, int s, int lane,
                                        T (&acc)[D], int uw = 0) {
    // uw > 0: every slice of this operator is uw entries wide (DevSell::uniform_w) -- the slice's place follows from its number and the wave's
    // first loads are the entries themselves, not the two pointers they would otherwise wait for (one dependent memory round trip less per wave)
    const int64_t p0 = uw ? (int64_t)s * (uw << 6) : slice_ptr[s];
    int w = uw ? uw : (int)((slice_ptr[s + 1] - p0) >> 6);
    const int* cp = col + p0 + lane;
    const T* vp = val + p0 + lane;
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.0;
    for (; w >= G; w -= G, cp += G * 64, vp += G * 64) row_dot_group<T, D, G, XI>(cp, vp, x, ld, acc);
    switch (w) {                                                   // w < G here
        case 1: row_dot_group<T, D, 1, XI>(cp, vp, x, ld, acc); break;
        case 2: row_dot_group<T, D, 2, XI>(cp, vp, x, ld, acc); break;
        case 3: row_dot_group<T, D, 3, XI>(cp, vp, x, ld, acc); break;
        case 4: if (G > 4) row_dot_group<T, D, 4, XI>(cp, vp, x, ld, acc); break;
        case 5: if (G > 4) row_dot_group<T, D, 5, XI>(cp, vp, x, ld, acc); break;
        case 6: if (G > 4) row_dot_group<T, D, 6, XI>(cp, vp, x, ld, acc); break;
        case 7: if (G > 4) row_dot_group<T, D, 7, XI>(cp, vp, x, ld, acc); break;
        default: break;
    }
}

// ---- 16-bit column codes (level 0; setup_kernels.hip.hpp::compress_cols) ------------------------------------------------------------
// The fine-level kernels run at the HBM limit for the bytes they move, so the way to a faster launch is fewer bytes: a column index is
// 4 of the 12 bytes of an entry.  The columns of the 64 rows of a slice are the neighbours of 64 consecutive rows: a few short ranges
// (one or two per colour class).  Every slice gets up to NW "windows" [base_k, base_k + 2^dbits) that cover its columns -- 8 windows of
// 8 192 columns (dbits = 13) for meshes, 32 windows of 2 048 (dbits = 11) for kNN graphs, whose 64 rows reach into every colour class --;
// code = k << dbits | (column - base_k), two codes to a 32-bit word (entries 2 q and 2 q + 1 of a row share word q of the slice's
// region), and the bases of the slice sit in lanes 0 .. NW-1 of one register, picked per entry by ds_bpermute.  Same columns in the same
// order: the results are bit-identical to the 32-bit path.  A slice that NW windows cannot cover (rows of tiny colour classes,
// scattered over the mesh) carries -1 as its first base and is read through the 32-bit indices: the wave learns that from the base
// register AFTER it has issued its first group of loads, so the other slices never wait for the answer.
template <class T, int D, int W, bool FLAGS, int XI = 0, bool KEEP = false>
__device__ __forceinline__ void row_dot_group16(const unsigned* __restrict__ cp, const int* __restrict__ cp32, const T* __restrict__ vp, const T* x, int ld,
                                                int basev, int dbits, int& fallback, T (&acc)[D]) {
    unsigned pk[(W + 1) / 2];
    T v[W];
    // KEEP: the level's operators are small enough to stay on the chip between the launches that read them (DevSell::resident) -- ordinary
    // loads; otherwise the matrix is streamed past the caches, which then belong to the vectors
#pragma unroll
    for (int q = 0; q < (W + 1) / 2; ++q) pk[q] = KEEP ? cp[q * 64] : __builtin_nontemporal_load(cp + q * 64);
#pragma unroll
    for (int j = 0; j < W; ++j) v[j] = KEEP ? vp[j * 64] : __builtin_nontemporal_load(vp + j * 64);
    if constexpr (FLAGS) { if (fallback < 0) fallback = __builtin_amdgcn_readfirstlane(basev) < 0 ? 1 : 0; }
    int c[W];
    if (FLAGS && fallback) {
#pragma unroll
        for (int j = 0; j < W; ++j) c[j] = __builtin_nontemporal_load(cp32 + j * 64);
    } else {
        const unsigned mask = (1u << dbits) - 1u;
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const unsigned u = (j & 1) ? pk[j >> 1] >> 16 : pk[j >> 1] & 0xffffu;
            c[j] = __builtin_amdgcn_ds_bpermute((int)((u >> dbits) << 2), basev) + (int)(u & mask);
        }
    }
    T xv[W][D];
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int d = 0; d < D; ++d) xv[j][d] = x[x_at<D, XI>(c[j], d, ld)];
#pragma unroll
    for (int j = 0; j < W; ++j)
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] += v[j] * xv[j][d];
}

// row_dot with the columns read from the codes (G is even: a group's codes are whole words)
template <class T, int D, bool FLAGS, int G = DotGroup<D>::value, int XI = 0, bool KEEP = false>
__device__ __forceinline__ void row_dot16(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col, const unsigned* __restrict__ col16,
                                          const int* __restrict__ win_base, int dbits, const T* __restrict__ val, const T* x, int ld, int s, int lane, T (&acc)[D], int uw = 0) {
    const int wshift = 16 - dbits;                                        // windows per slice = 1 << wshift
    const int basev = win_base[((int64_t)s << wshift) + (lane & ((1 << wshift) - 1))];
    const int64_t p0 = uw ? (int64_t)s * (uw << 6) : slice_ptr[s];        // (uw: see row_dot)
    int w = uw ? uw : (int)((slice_ptr[s + 1] - p0) >> 6);
    const unsigned* cp = col16 + p0 + lane;
    const int* cp32 = col + p0 + lane;
    const T* vp = val + p0 + lane;
    int fallback = -1;                                                    // not known yet
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.0;
    for (; w >= G; w -= G, cp += (G / 2) * 64, cp32 += G * 64, vp += G * 64) row_dot_group16<T, D, G, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc);
    switch (w) {
        case 1: row_dot_group16<T, D, 1, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc); break;
        case 2: row_dot_group16<T, D, 2, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc); break;
        case 3: row_dot_group16<T, D, 3, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc); break;
        case 4: if (G > 4) row_dot_group16<T, D, 4, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc); break;
        case 5: if (G > 4) row_dot_group16<T, D, 5, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc); break;
        case 6: if (G > 4) row_dot_group16<T, D, 6, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc); break;
        case 7: if (G > 4) row_dot_group16<T, D, 7, FLAGS, XI, KEEP>(cp, cp32, vp, x, ld, basev, dbits, fallback, acc); break;
        default: break;
    }
}

// C16 = 3 / 4: as 1 / 2 with ordinary instead of non-temporal loads of the operator (a fine level that fits the memory-side cache).
// C16 = 0: 32-bit indices.  C16 = 1: codes from slice `from` on (the uncovered slices are a short prefix of the numbering -- tiny colour
// classes come first -- and the branch on the wave-uniform slice number has nothing to wait for).  C16 = 2: uncovered slices anywhere,
// found through their flag.  c16_arg = from (24 bits) | uniform slice width << 24 (6 bits, 0: widths differ -- read the slice pointers) | format << 30
// (format 0: 13 offset bits / 8 windows, 1: 11 offset bits / 32 windows).
template <class T, int D, int C16, int G = DotGroup<D>::value, int XI = 0>
__device__ __forceinline__ void row_dot_sel(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col, const unsigned* __restrict__ col16,
                                            const int* __restrict__ win_base, int c16_arg, const T* __restrict__ val, const T* x, int ld, int s, int lane,
                                            T (&acc)[D]) {
    if constexpr (C16 == 1 || C16 == 3) {
        const int dbits = (c16_arg >> 30) & 1 ? 11 : 13;
        const int uw = (c16_arg >> 24) & 63;
        if (s >= (c16_arg & 0xffffff)) row_dot16<T, D, false, G, XI, C16 == 3>(slice_ptr, col, col16, win_base, dbits, val, x, ld, s, lane, acc, uw);
        else row_dot<T, D, G, XI>(slice_ptr, col, val, x, ld, s, lane, acc, uw);
    } else if constexpr (C16 == 2 || C16 == 4) {
        row_dot16<T, D, true, G, XI, C16 == 4>(slice_ptr, col, col16, win_base, (c16_arg >> 30) & 1 ? 11 : 13, val, x, ld, s, lane, acc, (c16_arg >> 24) & 63);
    } else row_dot<T, D, G, XI>(slice_ptr, col, val, x, ld, s, lane, acc);
}

// Quad layout (LPR = 4 lanes per row): add the four sub-lane partial sums; every lane of the quad gets the total.
template <class T, int D>
__device__ __forceinline__ void quad_reduce(T (&acc)[D]) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
        acc[c] += __shfl_xor(acc[c], 1, 64);
        acc[c] += __shfl_xor(acc[c], 2, 64);
    }
}

// One colour of a multicolour Gauss-Seidel sweep: rows [slice_begin*64, slice_end*64).
//   x_i <- (b_i - sum_{j != i} a_ij x_j) / a_ii          (gravomg/src/multigrid_solver.cpp:1200-1208)
// Rows of one colour do not couple, so the parallel update equals the reference's sequential sweep in
// the colour-permuted ordering.  omega != 1 relaxes the update (SOR, gmg_config::gs_omega).  FINE tags the level-0 instantiation so that profilers report the dominant
// (fine-level) launches under their own kernel name.
template <class T, int D, int FINE, int OM = 0>
__global__ __launch_bounds__(kBlock) void gs_color(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                   const T* __restrict__ val, const T* __restrict__ diag,
                                                   const T* __restrict__ b, T* x, int ld, int slice_begin,
                                                   int slice_end, int xcd_swizzle, T omega, const unsigned* __restrict__ col16 = nullptr,
                                                   const int* __restrict__ win_base = nullptr, int c16_arg = 0,
                                                   const unsigned long long* __restrict__ plain_rows = nullptr, const int* __restrict__ go = nullptr) {
    // `go`: this launch was enqueued AHEAD of the solve loop's decision (the first colour launch of the next cycle, behind the residual check of
    // this one: engine.hip, "head of the next cycle"); the check's reduction wrote the decision -- a stopped iteration's x stays as it is
    if (go != nullptr && *go == 0) return;
    const int s = slice_begin + wave_slice(slice_end - slice_begin, xcd_swizzle);
    if (s >= slice_end) return;
    const int lane = threadIdx.x & 63;
    const int row = s * 64 + lane;
    T acc[D];
    row_dot_sel<T, D, (FINE >= 2 ? FINE - 1 : 0)>(slice_ptr, col, col16, win_base, c16_arg, val, x, ld, s, lane, acc);      // FINE 2 / 3: level 0 with 16-bit column codes (row_dot_sel mode 1 / 2)
    const T dg = diag[row];
    if constexpr (OM != 0) {
        // hybrid Gauss-Seidel of a partitioned level 0 (engine_dist.hip.hpp::p2p_smooth): rows whose bit is set in plain_rows[slice] couple to rows of
        // another rank, whose values are a sweep old -- a Jacobi coupling, which over-relaxation amplifies: those rows take the plain update
        const T om = (plain_rows[s] >> lane) & 1ull ? (T)1.0 : omega;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const T xi = x[row + (int64_t)c * ld];
            x[row + (int64_t)c * ld] = xi + om * ((b[row + (int64_t)c * ld] - acc[c]) / dg - xi);
        }
        return;
    }
    if (omega == (T)1.0) {                 // kernel argument: a scalar branch.  The reference's update, no read of x_i
#pragma unroll
        for (int c = 0; c < D; ++c) x[row + (int64_t)c * ld] = (b[row + (int64_t)c * ld] - acc[c]) / dg;
    } else {                               // successive over-relaxation: x_i <- x_i + omega (x_i^GS - x_i)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const T xi = x[row + (int64_t)c * ld];
            x[row + (int64_t)c * ld] = xi + omega * ((b[row + (int64_t)c * ld] - acc[c]) / dg - xi);
        }
    }
}

// The LAST colour launch of a V-cycle's post-smoothing, when a residual check follows: the same update, plus this colour's share of
// the check's sums.  Right after its update a row's residual needs no second pass over the matrix: the off-diagonal sum is in
// registers and no later launch changes x before the check, so r_i = sum_{j != i} a_ij x_j + a_ii x_i^new - b_i is exactly the value
// (same expression, same operands) the norm kernel would compute -- which then only visits the rows of the other colours (a quarter
// less of its 336 MB at four colours).  (The algebraic shortcut r_i = a_ii (1 - omega)(x_i^GS - x_i^old) is NOT used: near the
// attainable accuracy it under-reports the residue, 5.9e-8 for a true 6.2e-8 on a 7 680-vertex Poisson problem.)  partials[blockIdx][2 D] = sum w r^2 / sum w b^2 over the block's rows (zeros for blocks
// beyond the range), added by reduce_partials after the norm kernel's own.
template <int D, int C16 = 0>
__global__ __launch_bounds__(kBlock) void gs_color_norm(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                        const double* __restrict__ val, const double* __restrict__ diag,
                                                        const double* __restrict__ b, double* x, int ld, int slice_begin, int slice_end,
                                                        double omega, const double* __restrict__ weight, double* __restrict__ partials,
                                                        const unsigned* __restrict__ col16 = nullptr, const int* __restrict__ win_base = nullptr, int c16_arg = 0) {
    __shared__ double red[kWavesPerBlock][2 * D];
    const int s = slice_begin + wave_slice(slice_end - slice_begin, 1);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double sums[2 * D];
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) sums[c] = 0.0;
    if (s < slice_end) {
        const int row = s * 64 + lane;
        double acc[D];
        row_dot_sel<double, D, C16>(slice_ptr, col, col16, win_base, c16_arg, val, x, ld, s, lane, acc);
        const double dg = diag[row];
        const double w = weight ? weight[row] : 1.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double bi = b[row + (int64_t)c * ld];
            const double xi = x[row + (int64_t)c * ld];
            const double xn = omega == 1.0 ? (bi - acc[c]) / dg : xi + omega * ((bi - acc[c]) / dg - xi);
            x[row + (int64_t)c * ld] = xn;
            const double r = acc[c] + dg * xn - bi;                  // the expression of residual_norm_slices, on the same operands
            sums[2 * c] = (r * w) * r;
            sums[2 * c + 1] = (bi * w) * bi;
        }
    }
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) {
        double v = sums[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        double v = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kWavesPerBlock; ++w2) v += red[w2][threadIdx.x];
        partials[(int64_t)blockIdx.x * (2 * D) + threadIdx.x] = v;
    }
}

// The LAST colour launch of the pre-smoothing: the same update, plus the residual r_i = b_i - (sum_{j != i} a_ij x_j + a_ii x_i^new) of
// its own rows -- the expression of spmv_full<MODE 1> on the same operands (no later launch changes x before the residual), so the
// residual kernel only visits the rows of the other colours.
template <int D, int C16 = 0, int YI = 0>
__global__ __launch_bounds__(kBlock) void gs_color_residual(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                            const double* __restrict__ val, const double* __restrict__ diag,
                                                            const double* __restrict__ b, double* x, double* __restrict__ r, int ld,
                                                            int slice_begin, int slice_end, double omega, const unsigned* __restrict__ col16 = nullptr,
                                                            const int* __restrict__ win_base = nullptr, int c16_arg = 0) {
    const int s = slice_begin + wave_slice(slice_end - slice_begin, 1);
    if (s >= slice_end) return;
    const int lane = threadIdx.x & 63;
    const int row = s * 64 + lane;
    double acc[D];
    row_dot_sel<double, D, C16>(slice_ptr, col, col16, win_base, c16_arg, val, x, ld, s, lane, acc);
    const double dg = diag[row];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const double bi = b[row + (int64_t)c * ld];
        double xn;
        if (omega == 1.0) xn = (bi - acc[c]) / dg;
        else { const double xi = x[row + (int64_t)c * ld]; xn = xi + omega * ((bi - acc[c]) / dg - xi); }
        x[row + (int64_t)c * ld] = xn;
        const double ax = acc[c] + dg * xn;
        r[YI ? (int64_t)row * D + c : row + (int64_t)c * ld] = bi - ax;           // (YI: the residual as an interleaved multi-vector, what the restriction gathers from)
    }
}

// Block-hybrid Gauss-Seidel sweep, ONE launch per sweep (coarse levels, where a launch per colour is
// latency-bound: 13-18 colours on the Galerkin operators).
//   * one workgroup (up to 1024 rows = 16 wavefronts) per compact block of rows (host_plan.hpp);
//   * couplings that leave the block use the PREVIOUS iterate x_in (Jacobi between blocks) and are summed
//     up front by all lanes at once -- their gathers are independent of the in-block ordering;
//   * the block's own x lives in LDS; in-block couplings are applied colour by colour (exact multicolour
//     Gauss-Seidel inside the block) with __syncthreads() between colours.  Each lane keeps its row's
//     in-block entries (16-bit local column + value) in registers, loaded before the colour loop, so a
//     colour step is LDS gathers + FMAs only;
//   * x_out != x_in (T buffer): other blocks read x_in while this one writes, so the result is
//     deterministic.
// In matrix form one sweep is x_out = x_in + T^{-1} (b - A x_in), T = D + strict-lower(A restricted to the
// block diagonal, device order); tests/test_gpu_parity.py checks exactly that.
constexpr int kBlockRows = 1024;
template <class T, int D, int WIN>
__global__ __launch_bounds__(kBlockRows) void gs_block(const int* __restrict__ blk_begin, const int* __restrict__ blk_ncolors,
                                                       const unsigned char* __restrict__ row_color,
                                                       const int64_t* __restrict__ in_ptr, const unsigned short* __restrict__ in_col,
                                                       const T* __restrict__ in_val, const int64_t* __restrict__ out_ptr,
                                                       const int* __restrict__ out_col, const T* __restrict__ out_val,
                                                       const T* __restrict__ diag, const T* __restrict__ b,
                                                       const T* __restrict__ x_in, T* __restrict__ x_out, int ld) {
    __shared__ T xs[D][kBlockRows];
    const int blk = blockIdx.x;
    const int r0 = blk_begin[blk];
    const int nrows = blk_begin[blk + 1] - r0;            // multiple of 64
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool active = wave * 64 < nrows;                 // wave-uniform
    const int row = r0 + t;
    T rhs[D], dg = 1.0;
    int mycolor = -1, w = 0;
    int64_t p0 = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) rhs[c] = 0.0;
    // In-block entries of this lane's row, zero-padded to WIN so the colour loop can run unconditional
    // chunks of CH entries (CH independent LDS gathers in flight instead of one wait per entry).  The 16-bit
    // local columns are packed two per register.
    constexpr int CH = D == 1 ? 8 : 4;
    T v[WIN];
    unsigned cpk[WIN / 2];
#pragma unroll
    for (int j = 0; j < WIN; ++j) v[j] = 0.0;
#pragma unroll
    for (int j = 0; j < WIN / 2; ++j) cpk[j] = 0u;
    if (active) {
        const int s = (r0 >> 6) + wave;
#pragma unroll
        for (int c = 0; c < D; ++c) xs[c][t] = x_in ? x_in[row + (int64_t)c * ld] : (T)0.0;      // x_in == nullptr: sweep from a zero iterate
        p0 = in_ptr[s];
        w = (int)((in_ptr[s + 1] - p0) >> 6);
#pragma unroll
        for (int j0 = 0; j0 < WIN; j0 += 8)
            if (j0 < w) {
#pragma unroll
                for (int j = j0; j < j0 + 8; ++j)
                    if (j < w) {
                        v[j] = in_val[p0 + (int64_t)j * 64 + lane];
                        cpk[j >> 1] |= (unsigned)in_col[p0 + (int64_t)j * 64 + lane] << ((j & 1) * 16);
                    }
            }
        T acc[D];
        if (x_in) row_dot<T, D, (D == 1 ? 8 : 4)>(out_ptr, out_col, out_val, x_in, ld, s, lane, acc);
        else {                     // zero iterate: nothing couples in from outside the block
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < D; ++c) rhs[c] = b[row + (int64_t)c * ld] - acc[c];
        dg = 1.0 / diag[row];      // reciprocal once, outside the sequential colour loop
        mycolor = row_color[row];
    }
    __syncthreads();
    const int nc = blk_ncolors[blk];
    for (int col = 0; col < nc; ++col) {
        if (mycolor == col) {
            T s_[D];
#pragma unroll
            for (int c = 0; c < D; ++c) s_[c] = 0.0;
#pragma unroll
            for (int j0 = 0; j0 < WIN; j0 += CH)
                if (j0 < w) {
                    T xv[CH][D];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int cj = (cpk[(j0 + j) >> 1] >> (((j0 + j) & 1) * 16)) & 0xffff;
#pragma unroll
                        for (int c = 0; c < D; ++c) xv[j][c] = xs[c][cj];
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int c = 0; c < D; ++c) s_[c] += v[j0 + j] * xv[j][c];
                }
            for (int j = WIN; j < w; ++j) {                 // rows longer than the register window (rare)
                const T vj = in_val[p0 + (int64_t)j * 64 + lane];
                const int cj = in_col[p0 + (int64_t)j * 64 + lane];
#pragma unroll
                for (int c = 0; c < D; ++c) s_[c] += vj * xs[c][cj];
            }
#pragma unroll
            for (int c = 0; c < D; ++c) xs[c][t] = (rhs[c] - s_[c]) * dg;
        }
        __syncthreads();
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < D; ++c) x_out[row + (int64_t)c * ld] = xs[c][t];
    }
}

// The same sweep with the OFF-BLOCK operator in block-CSR storage (big blocked levels, 64-row blocks, one wavefront per
// block).  A block's SELL slice is as wide as its longest row, and off-block entries are concentrated on the block's
// border rows (interior rows have none): the padded off-block SELL operator stored 2.99x its real entries on the
// 506 k-row level (PMC: the sweep moved 218 MB for 103 MB of entries and was bound by it).  Here the off-block
// entries of the level are a plain CSR in device numbering; rows of a block are contiguous, so a block's entries are
// ONE contiguous chunk, which the wave copies into LDS with perfectly coalesced loads; every lane then sums its row's
// couplings from there (x gathered from the previous iterate).  The in-block operator stays in SELL (its entries go
// to registers for the colour loop, exactly as in gs_block).  Same arithmetic, same order: results are bitwise those
// of gs_block.
template <class T, int D, int WIN>
__global__ __launch_bounds__(64) void gs_block_csrout(const int* __restrict__ blk_begin, const int* __restrict__ blk_ncolors,
                                                      const unsigned char* __restrict__ row_color,
                                                      const int64_t* __restrict__ in_ptr, const unsigned short* __restrict__ in_col,
                                                      const T* __restrict__ in_val, const int* __restrict__ out_ptr,
                                                      const int* __restrict__ out_col, const T* __restrict__ out_val,
                                                      const T* __restrict__ diag, const T* __restrict__ b, const T* __restrict__ x_in,
                                                      T* __restrict__ x_out, int ld, int cap) {
    extern __shared__ unsigned char smem_raw[];
    T* sval = reinterpret_cast<T*>(smem_raw);                          // cap values
    int* scol = reinterpret_cast<int*>(sval + cap);                    // cap columns
    T* xs = reinterpret_cast<T*>(scol + cap);                          // D x 64: the block's x
    const int blk = blockIdx.x;
    const int lane = threadIdx.x;
    const int r0 = blk_begin[blk];                                     // 64 rows (padding rows: no entries, diag 1, b 0)
    const int row = r0 + lane;
    const int s = r0 >> 6;
    const int e0 = out_ptr[r0], e1 = out_ptr[r0 + 64];
    const int cnt = e1 - e0;
    if (x_in) for (int e = lane; e < cnt; e += 64) { sval[e] = out_val[e0 + e]; scol[e] = out_col[e0 + e]; }
#pragma unroll
    for (int c = 0; c < D; ++c) xs[c * 64 + lane] = x_in ? x_in[row + (int64_t)c * ld] : (T)0.0;     // nullptr: zero iterate
    // in-block entries of this lane's row: SELL -> registers (zero-padded window, 16-bit local columns packed in pairs)
    constexpr int CH = D == 1 ? 8 : 4;
    T v[WIN];
    unsigned cpk[WIN / 2];
#pragma unroll
    for (int j = 0; j < WIN; ++j) v[j] = 0.0;
#pragma unroll
    for (int j = 0; j < WIN / 2; ++j) cpk[j] = 0u;
    const int64_t p0 = in_ptr[s];
    const int w = (int)((in_ptr[s + 1] - p0) >> 6);
#pragma unroll
    for (int j0 = 0; j0 < WIN; j0 += 8)
        if (j0 < w) {
#pragma unroll
            for (int j = j0; j < j0 + 8; ++j)
                if (j < w) {
                    v[j] = in_val[p0 + (int64_t)j * 64 + lane];
                    cpk[j >> 1] |= (unsigned)in_col[p0 + (int64_t)j * 64 + lane] << ((j & 1) * 16);
                }
        }
    const int mb = out_ptr[row] - e0, me = out_ptr[row + 1] - e0;
    T rhs[D];
    const T dg = (T)1.0 / diag[row];
    const int mycolor = row_color[row];
    __syncthreads();
    // ---- off-block couplings (previous iterate), four gathers in flight; summed in stored order like gs_block
    {
        T acc[D];
#pragma unroll
        for (int c = 0; c < D; ++c) acc[c] = 0.0;
        int e = x_in ? mb : me;            // zero iterate: no off-block couplings
        for (; e + 4 <= me; e += 4) {
            int cc[4]; T vv[4]; T xv[4][D];
#pragma unroll
            for (int j = 0; j < 4; ++j) { cc[j] = scol[e + j]; vv[j] = sval[e + j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < D; ++c) xv[j][c] = x_in[cc[j] + (int64_t)c * ld];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < D; ++c) acc[c] += vv[j] * xv[j][c];
        }
        for (; e < me; ++e) {
            const int cj = scol[e]; const T vj = sval[e];
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] += vj * x_in[cj + (int64_t)c * ld];
        }
#pragma unroll
        for (int c = 0; c < D; ++c) rhs[c] = b[row + (int64_t)c * ld] - acc[c];
    }
    const int nc = blk_ncolors[blk];
    for (int col = 0; col < nc; ++col) {
        if (mycolor == col) {
            T s_[D];
#pragma unroll
            for (int c = 0; c < D; ++c) s_[c] = 0.0;
#pragma unroll
            for (int j0 = 0; j0 < WIN; j0 += CH)
                if (j0 < w) {
                    T xv[CH][D];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int cj = (cpk[(j0 + j) >> 1] >> (((j0 + j) & 1) * 16)) & 0xffff;
#pragma unroll
                        for (int c = 0; c < D; ++c) xv[j][c] = xs[c * 64 + cj];
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int c = 0; c < D; ++c) s_[c] += v[j0 + j] * xv[j][c];
                }
            for (int j = WIN; j < w; ++j) {                            // rows longer than the register window (rare)
                const T vj = in_val[p0 + (int64_t)j * 64 + lane];
                const int cj = in_col[p0 + (int64_t)j * 64 + lane];
#pragma unroll
                for (int c = 0; c < D; ++c) s_[c] += vj * xs[c * 64 + cj];
            }
#pragma unroll
            for (int c = 0; c < D; ++c) xs[c * 64 + lane] = (rhs[c] - s_[c]) * dg;
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < D; ++c) x_out[row + (int64_t)c * ld] = xs[c * 64 + lane];
}

// The same sweep for the big blocked levels (64-row blocks, one wavefront per block) on UNPADDED storage.  A block's
// SELL slice is as wide as its longest row (2.96x the real off-block entries and 1.61x the in-block ones on the 506 k-row
// level: the SELL sweep moved 200 MB for 103 MB of entries and paid one load -> gather round trip per group of 8 padded
// columns).  Here the operator is two block-ordered CSRs in device numbering (rows of a block are contiguous, so a block's
// entries are ONE chunk; a wave's loads are contiguous runs of entries whatever the row lengths are):
//   E  the entries that multiply the PREVIOUS iterate: everything that leaves the block, and the in-block entries whose
//      column comes later in the block than the row ("upper").  Handled ENTRY-PARALLEL: lane l gathers x_in for ITS
//      ENTRIES (ceil(entries / 64) gathers per block instead of one per padded column), writes the products to LDS in
//      entry order, and every row then sums its run of products (ascending column order).
//   L  the in-block entries whose column comes earlier than the row ("lower", 16-bit local column): the sequential part.
//      The slots are transposed through LDS into the row's lane (registers); the colour loop is then gathers from the
//      block's x in LDS + FMAs only: a row reads updated values exclusively (rows are colour-sorted inside a block, so
//      "lower" == "of an earlier colour").
// Every load a block needs is issued before the first use.  Blocks / rows with more entries than the register windows
// (kEpE * 64 explicit entries, kEpL * 64 lower entries per block, kEpW lower entries per row) read the excess from global
// memory / LDS in the same order.  Same matrix form as the sweeps above, x_out = x_in + T^-1 (b - A x_in) with T = D +
// strict lower triangle of the block diagonal in device order; only the summation grouping differs (tested to 1e-12).
// What bounds it: rounds 2 and 3 read it as bound by the dependent LDS round trips of one block (a first entry-parallel
// version, which also reduced the in-block products through LDS colour by colour, had ~85 of them instead of ~25 and was slower
// than the SELL sweep); round 4 found the instruction count to be the bound -- see the kernel's comment below.
constexpr int kEpE = 12, kEpL = 8, kEpW = 16;
constexpr int kEpZeroBytes = 256;                                      // zero region in LDS: 16 staged records / 32 doubles (see ep_lds_bytes)
// a staged lower entry: value + byte offset of its column inside the block's x (one address select serves both reads)
template <class T> struct EpRec;
template <> struct EpRec<double> { typedef int type __attribute__((ext_vector_type(4))); };
template <> struct EpRec<float> { typedef int type __attribute__((ext_vector_type(2))); };

// Raw buffer access to a wave-uniform chunk (base pointer and byte count in SGPRs): ONE instruction per load, no per-lane bounds
// test and no 64-bit address arithmetic -- reads past `bytes` return zero without touching memory.  (gfx950 buffer resource, word 3 =
// 0x00020000: raw 32-bit data format, bounds checking on.)  `NT`: streamed once (matrix arrays).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ep_chunk(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
template <class T, bool NT> __device__ __forceinline__ T ep_load(__amdgpu_buffer_rsrc_t r, int byte_off) {
    static_assert(sizeof(T) == 8 || sizeof(T) == 4 || sizeof(T) == 2, "");
    constexpr int aux = NT ? 2 : 0;
    T out;
    if constexpr (sizeof(T) == 8) { auto w = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, aux); __builtin_memcpy(&out, &w, 8); }
    else if constexpr (sizeof(T) == 4) { auto w = __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, aux); __builtin_memcpy(&out, &w, 4); }
    else { auto w = __builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, aux); __builtin_memcpy(&out, &w, 2); }
    return out;
}

// dynamic LDS of gs_block_ep: the block's x (D x 64), the zero region, and the staging region -- first the products of the explicit
// entries (at least the register window's 64 kEpE, so that no slot needs a guard), then the lower entries as records
template <class T> inline size_t ep_lds_bytes(int D, int cap_e, int cap_l) {
    const size_t pe = (size_t)(cap_e > 64 * kEpE ? cap_e : 64 * kEpE), sl = (size_t)(cap_l > 64 * kEpL ? cap_l : 64 * kEpL);
    const size_t stage = pe * sizeof(T) > sl * sizeof(typename EpRec<T>::type) ? pe * sizeof(T) : sl * sizeof(typename EpRec<T>::type);
    return (size_t)D * 64 * sizeof(T) + kEpZeroBytes + stage;
}

// The sequential half of the entry-parallel block sweep (gs_block_ep below; also the tail of gmgk::restrict_sweep0): the lower in-block entries go
// through LDS into the row's lane, then the colours of the block one after the other on the block's x in LDS, then the store.  `rhs` = b minus
// the explicit part.  One wave = one block; every __syncthreads() here is reached by exactly the waves of the workgroup that are still alive
// (restrict_sweep0 retires three of its four waves before it calls this: S_BARRIER "waits on only the surviving waves" of a workgroup whose
// other waves have terminated -- CDNA ISA guide, S_BARRIER).
template <class T, int D>
__device__ __forceinline__ void ep_block_lower(T* xs, T* zeroT, typename EpRec<T>::type* srec, int lane, int row, int blk, int q0, int nL, int lb, int nlow, int mycolor, T dg,
                                               T (&rhs)[D], T (&lv)[kEpL], unsigned short (&lc)[kEpL], const unsigned short* __restrict__ l_col, const T* __restrict__ l_val,
                                               const int* __restrict__ blk_ncolors, T* __restrict__ x_out, int ld, T* __restrict__ x_out_i) {
    typedef typename EpRec<T>::type Rec;
    const Rec* zeroR = reinterpret_cast<const Rec*>(zeroT);
    // ---- L: slots -> LDS (records: value, byte offset of the column in xs) -> the row's lane
    auto make_rec = [](T val, unsigned short col) {
        Rec rr;
        if constexpr (sizeof(T) == 8) { long long bits; __builtin_memcpy(&bits, &val, 8); rr.x = (int)bits; rr.y = (int)(bits >> 32); rr.z = (int)col * 8; rr.w = 0; }
        else { int bits; __builtin_memcpy(&bits, &val, 4); rr.x = bits; rr.y = (int)col * 4; }
        return rr;
    };
#pragma unroll
    for (int m = 0; m < kEpL; ++m) srec[64 * m + lane] = make_rec(lv[m], lc[m]);
    if (nL > 64 * kEpL)
        for (int e = 64 * kEpL + lane; e < nL; e += 64) srec[e] = make_rec(l_val[q0 + e], l_col[q0 + e]);
    __syncthreads();
    T v[kEpW];
    int xa[kEpW];                                                      // byte offsets of the columns inside xs (column 0 of a multi-vector)
    {
        const Rec* ra = srec + lb;
#pragma unroll
        for (int j = 0; j < kEpW; ++j) {                               // (one select of the ADDRESS: slots beyond the row's run read a zero record)
            const Rec rr = (j < nlow ? ra : zeroR)[j];
            if constexpr (sizeof(T) == 8) { const long long bits = ((long long)(unsigned)rr.x) | ((long long)rr.y << 32); __builtin_memcpy(&v[j], &bits, 8); xa[j] = rr.z; }
            else { const int bits = rr.x; __builtin_memcpy(&v[j], &bits, 4); xa[j] = rr.y; }
        }
    }
    const int nc = blk_ncolors[blk];
    // four lower entries of this lane's row: all gathers in flight, then the FMAs in stored order
    auto chunk4 = [&](int j0, T (&s_)[D]) {
        T xv[4][D];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < D; ++c) xv[j][c] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(xs + c * 64) + xa[j0 + j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < D; ++c) s_[c] += v[j0 + j] * xv[j][c];
    };
    int col0 = 0;
    // rows of the first colour have no lower entries (rows of one colour do not couple): their update is rhs / diag, written by every lane
    // at once -- the other lanes' values are overwritten when their colour comes (nothing reads them before: a row's lower entries
    // point at earlier colours only)
    if (!__builtin_amdgcn_ballot_w64(mycolor == 0 && nlow > 0)) {
#pragma unroll
        for (int c = 0; c < D; ++c) xs[c * 64 + lane] = (rhs[c] - (T)0.0) * dg;
        col0 = 1;
        __syncthreads();
    }
    for (int col = col0; col < nc; ++col) {
        if (mycolor == col) {
            T s_[D];
#pragma unroll
            for (int c = 0; c < D; ++c) s_[c] = (T)0.0;
            // (wave-uniform tests: most rows of the early colours have few lower entries)
            if (__builtin_amdgcn_ballot_w64(nlow > 0)) {
                chunk4(0, s_);
                if (__builtin_amdgcn_ballot_w64(nlow > 4)) {
                    chunk4(4, s_);
                    if (__builtin_amdgcn_ballot_w64(nlow > 8)) {
                        chunk4(8, s_);
                        if (__builtin_amdgcn_ballot_w64(nlow > 12)) {
                            chunk4(12, s_);
                            for (int j = kEpW; j < nlow; ++j) {       // rows with more lower entries than the register window (rare)
                                const Rec rr = srec[lb + j];
                                T vj; int cb;
                                if constexpr (sizeof(T) == 8) { const long long bits = ((long long)(unsigned)rr.x) | ((long long)rr.y << 32); __builtin_memcpy(&vj, &bits, 8); cb = rr.z; }
                                else { const int bits = rr.x; __builtin_memcpy(&vj, &bits, 4); cb = rr.y; }
#pragma unroll
                                for (int c = 0; c < D; ++c) s_[c] += vj * *reinterpret_cast<const T*>(reinterpret_cast<const char*>(xs + c * 64) + cb);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < D; ++c) xs[c * 64 + lane] = (rhs[c] - s_[c]) * dg;
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < D; ++c) x_out[row + (int64_t)c * ld] = xs[c * 64 + lane];
    if (x_out_i) {                                                     // a second copy as an interleaved multi-vector (what the prolongation into the finer level gathers from)
#pragma unroll
        for (int c = 0; c < D; ++c) x_out_i[(int64_t)row * D + c] = xs[c * 64 + lane];
    }
}

// Round 4: the sweep is bound by the INSTRUCTIONS a wave issues -- a SIMD retires one block every ~3.2 us whether four or five
// waves share it (persistent launches with 8 ... 20 workgroups per compute unit: 12 us per block and wave up to 4 waves per SIMD,
// 16.5 us with 5; profiles/README.md round 4), 1 375 instructions per block at ~5.5 cycles each, of which the guarded loads (a
// scalar branch, a lane mask and 64-bit address arithmetic per slot), the selects of the row sums and of the transposition were
// more than half.  This version issues about half as many: chunk loads through buffer resources (above), no guards at all on
// the register window (LDS is sized for the whole window; unused slots read zeros and multiply x[0] by zero), out-of-run reads
// redirected to a zero region of LDS by ONE address select instead of value selects.  Same operations in the same order: the
// results are bitwise those of the round-2 kernel.
// `vgrid` > gridDim.x: persistent workgroups -- workgroup w sweeps the virtual blocks w, w + gridDim.x, ... (the XCD-aware map is
// applied to the virtual index, so a workgroup's blocks stay on its XCD's part of the level).
// STREAM: the operator's chunks are read with non-temporal loads (a level whose operator is too big to stay in the 256 MB memory-side cache
// between the launches that read it -- level 0 of a point cloud); otherwise with ordinary loads: the ~76 MB of the 506 k-row level of the
// 3 M mesh are read by five consecutive launches per cycle, and the second to fifth then find them on the chip (level 1: 140 -> 125 us per
// cycle, profiles/r05/t_*).
template <class T, int D, bool STREAM>
__global__ __launch_bounds__(64) void gs_block_ep(const int* __restrict__ blk_begin, const int* __restrict__ blk_ncolors,
                                                  const unsigned char* __restrict__ row_color, const int* __restrict__ l_ptr,
                                                  const unsigned short* __restrict__ l_col, const T* __restrict__ l_val,
                                                  const int* __restrict__ e_ptr, const int* __restrict__ e_col,
                                                  const T* __restrict__ e_val, const T* __restrict__ diag, const T* __restrict__ b,
                                                  const T* __restrict__ x_in, T* __restrict__ x_out, int ld, int cap_e, int cap_l, int n_blocks, int blk0,
                                                  int vgrid, T* __restrict__ x_out_i = nullptr) {
    extern __shared__ unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);                            // D x 64: the block's new x
    typedef typename EpRec<T>::type Rec;
    T* zeroT = xs + D * 64;                                            // the zero region: where reads beyond a row's run go
    const Rec* zeroR = reinterpret_cast<const Rec*>(zeroT);
    T* pbuf = zeroT + kEpZeroBytes / sizeof(T);                        // products of one column's explicit entries ...
    Rec* srec = reinterpret_cast<Rec*>(pbuf);                          // ... later the staged lower entries
    const int chunk = vgrid >> 3;
    const int lane = threadIdx.x;
    reinterpret_cast<int*>(zeroT)[lane] = 0;                           // kEpZeroBytes = 64 x 4
  for (int vb = (int)blockIdx.x; vb < vgrid; vb += (int)gridDim.x) {
    // XCD-aware block map: workgroup w runs on XCD w % 8; every XCD gets a contiguous run of (spatially neighbouring) blocks
    const int blk = __builtin_amdgcn_readfirstlane((vb & 7) * chunk + (vb >> 3));
    if (blk >= n_blocks) continue;
    // 64 rows (padding rows: no entries, diag 1, b 0, colour 0).  Without a block table the blocks are the level's own, in order:
    // block b starts at row 64 (blk0 + b) -- one dependent load less in front of everything else the wave does
    const int r0 = __builtin_amdgcn_readfirstlane(blk_begin ? blk_begin[blk] : (blk0 + blk) << 6);
    const int row = r0 + lane;
    const int e0 = e_ptr[r0], e1 = e_ptr[r0 + 64];
    const int q0 = l_ptr[r0], q1 = l_ptr[r0 + 64];
    const int nE = x_in ? e1 - e0 : 0;                                 // zero iterate: the explicit part vanishes
    const int nL = q1 - q0;
    int lane_b = lane;                                                 // (opaque per block: keeps the slot offsets in the loads' immediate fields instead of in hoisted registers)
    asm volatile("" : "+v"(lane_b));
    // ---- all loads in flight (register windows: 64 kEpE explicit, 64 kEpL lower entries; what a block has beyond them is read later)
    const __amdgpu_buffer_rsrc_t rc = ep_chunk(e_col + e0, nE * 4), rv = ep_chunk(e_val + e0, nE * (int)sizeof(T));
    const __amdgpu_buffer_rsrc_t rlv = ep_chunk(l_val + q0, nL * (int)sizeof(T)), rlc = ep_chunk(l_col + q0, nL * 2);
    int ec[kEpE];
    T ev[kEpE];
    if (nE > 0) {
#pragma unroll
        for (int k = 0; k < kEpE; ++k) { ec[k] = ep_load<int, STREAM>(rc, lane_b * 4 + 256 * k); ev[k] = ep_load<T, STREAM>(rv, lane_b * (int)sizeof(T) + 64 * k * (int)sizeof(T)); }
    } else {
#pragma unroll
        for (int k = 0; k < kEpE; ++k) { ec[k] = 0; ev[k] = (T)0.0; }
    }
    T lv[kEpL];
    unsigned short lc[kEpL];
#pragma unroll
    for (int m = 0; m < kEpL; ++m) { lv[m] = ep_load<T, STREAM>(rlv, lane_b * (int)sizeof(T) + 64 * m * (int)sizeof(T)); lc[m] = ep_load<unsigned short, STREAM>(rlc, lane_b * 2 + 128 * m); }
    const int eb = e_ptr[row] - e0, ee = x_in ? e_ptr[row + 1] - e0 : eb;
    const int lb = l_ptr[row] - q0, nlow = l_ptr[row + 1] - q0 - lb;
    const int mycolor = row_color[row];
    const T dg = (T)1.0 / diag[row];
    T rhs[D];
#pragma unroll
    for (int c = 0; c < D; ++c) { rhs[c] = b[row + (int64_t)c * ld]; xs[c * 64 + lane] = (T)0.0; }      // (zero: padded register slots multiply x[0] by 0)
    // ---- E: explicit part, one right-hand side at a time through the product buffer
    if (nE > 0) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const __amdgpu_buffer_rsrc_t rx = ep_chunk(x_in + (int64_t)c * ld, ld * (int)sizeof(T));
            T xg[kEpE];
#pragma unroll
            for (int k = 0; k < kEpE; ++k) xg[k] = ep_load<T, false>(rx, ec[k] * (int)sizeof(T));
#pragma unroll
            for (int k = 0; k < kEpE; ++k) pbuf[64 * k + lane] = ev[k] * xg[k];
            if (nE > 64 * kEpE) {                                      // beyond the register window (rare)
                const T* xc = x_in + (int64_t)c * ld;
                for (int e = 64 * kEpE + lane; e < nE; e += 64) pbuf[e] = e_val[e0 + e] * xc[e_col[e0 + e]];
            }
            __syncthreads();
            // every row sums its run in stored order, eight reads in flight; reads beyond the run hit the zero slots
            T acc = (T)0.0;
            const T* pa = pbuf + eb;
            int rem = ee - eb;
            while (__builtin_amdgcn_ballot_w64(rem > 0)) {
                T p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = (rem > j ? pa : zeroT)[j];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += p[j];
                pa += 8; rem -= 8;
            }
            rhs[c] -= acc;
            __syncthreads();                                           // the buffer is free again
        }
    }
    ep_block_lower<T, D>(xs, zeroT, srec, lane, row, blk, q0, nL, lb, nlow, mycolor, dg, rhs, lv, lc, l_col, l_val, blk_ncolors, x_out, ld, x_out_i);
    __syncthreads();                                                   // xs and the staging area are free for the next block
  }
}

// Restriction INTO a blocked level fused with that level's first pre-sweep (multigrid_solver.cpp:1069 + :1072-1073 + the first trip of :1063).
// The coarse correction starts from the zero iterate, so its first block sweep needs nothing but the right-hand side of its own block: no
// explicit part, no halo.  The restriction (quad layout: 16 coarse rows per slice, rows length-sorted inside 64-row windows) produces exactly
// the 64 right-hand sides of block b in the four slices 4 b .. 4 b + 3 -- one workgroup of four waves -- so the same workgroup goes on: the
// values are written to b (the later sweeps and the residual read them) AND kept in LDS, wave 0 runs the sequential half of the block sweep
// on them (ep_block_lower) and writes the level's first iterate.  One launch less per level and cycle, and b is not read back; the arithmetic
// of both halves is that of transfer<.., 0, 4, ..> and gs_block_ep<..>(x_in = nullptr): the same bits.
// Grid: vgrid = n_blocks rounded up to 8 workgroups of 256 threads (the XCD-aware block map of gs_block_ep).  Dynamic LDS: ep_lds_bytes(D, 0, cap_l)
// + D * 64 values (bs_off: where those begin).
template <class T, int D, bool STREAM, int C16, int XI>
__global__ __launch_bounds__(256) void restrict_sweep0(const int64_t* __restrict__ r_slice_ptr, const int* __restrict__ r_col, const T* __restrict__ r_val,
                                                       const int* __restrict__ r_row_of, const T* __restrict__ r_fine, int ldx,
                                                       const unsigned* __restrict__ r_col16, const int* __restrict__ r_win_base, int r_c16_arg,
                                                       T* __restrict__ b_out, const int* __restrict__ blk_ncolors, const unsigned char* __restrict__ row_color,
                                                       const int* __restrict__ l_ptr, const unsigned short* __restrict__ l_col, const T* __restrict__ l_val,
                                                       const T* __restrict__ diag, T* __restrict__ x_out, int ld, int n_blocks, int vgrid, int bs_off,
                                                       const int* __restrict__ blk_list = nullptr) {
    extern __shared__ unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);
    typedef typename EpRec<T>::type Rec;
    T* zeroT = xs + D * 64;
    Rec* srec = reinterpret_cast<Rec*>(zeroT + kEpZeroBytes / sizeof(T));
    T* bs = reinterpret_cast<T*>(smem_raw + bs_off);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int chunk = vgrid >> 3;
    const int vb = __builtin_amdgcn_readfirstlane(((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3));
    if (vb >= n_blocks) return;
    // blk_list: the blocks of ONE rank of a level partitioned by blocks (engine_dist.hip.hpp), n_blocks of them; else the level's blocks in order
    const int blk = blk_list ? __builtin_amdgcn_readfirstlane(blk_list[vb]) : vb;
    const int r0 = blk << 6;
    // wave 0: everything the sweep will need is requested before the restriction's own loads
    int q0 = 0, nL = 0, lb = 0, nlow = 0, mycolor = 0;
    T dg = (T)1.0;
    T lv[kEpL];
    unsigned short lc[kEpL];
    if (wave == 0) {
        reinterpret_cast<int*>(zeroT)[lane] = 0;
        const int row = r0 + lane;
        q0 = l_ptr[r0];
        nL = l_ptr[r0 + 64] - q0;
        const __amdgpu_buffer_rsrc_t rlv = ep_chunk(l_val + q0, nL * (int)sizeof(T)), rlc = ep_chunk(l_col + q0, nL * 2);
#pragma unroll
        for (int m = 0; m < kEpL; ++m) { lv[m] = ep_load<T, STREAM>(rlv, lane * (int)sizeof(T) + 64 * m * (int)sizeof(T)); lc[m] = ep_load<unsigned short, STREAM>(rlc, lane * 2 + 128 * m); }
        lb = l_ptr[row] - q0;
        nlow = l_ptr[row + 1] - q0 - lb;
        mycolor = row_color[row];
        dg = (T)1.0 / diag[row];
#pragma unroll
        for (int c = 0; c < D; ++c) { bs[c * 64 + lane] = (T)0.0; xs[c * 64 + lane] = (T)0.0; }      // (padding rows of the block: right-hand side zero)
    } else {
#pragma unroll
        for (int m = 0; m < kEpL; ++m) { lv[m] = (T)0.0; lc[m] = 0; }
    }
    __syncthreads();
    {   // ---- the restriction of slice 4 blk + wave (transfer_slice<T, D, 0, 4, C16, XI>)
        const int s = 4 * blk + wave;
        const int srow = s * 16 + lane / 4;
        T acc[D];
        row_dot_sel<T, D, C16, 8, XI>(r_slice_ptr, r_col, r_col16, r_win_base, r_c16_arg, r_val, r_fine, ldx, s, lane, acc);
        quad_reduce<T, D>(acc);
        if ((lane & 3) == 0) {
            const int row = r_row_of ? r_row_of[srow] : srow;
            if (row >= 0) {
#pragma unroll
                for (int c = 0; c < D; ++c) { b_out[row + (int64_t)c * ld] = acc[c]; bs[c * 64 + (row - r0)] = acc[c]; }
            }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // ---- wave 0: the block sweep from the zero iterate (gs_block_ep with x_in == nullptr)
    T rhs[D];
#pragma unroll
    for (int c = 0; c < D; ++c) rhs[c] = bs[c * 64 + lane];
    ep_block_lower<T, D>(xs, zeroT, srec, lane, r0 + lane, blk, q0, nL, lb, nlow, mycolor, dg, rhs, lv, lc, l_col, l_val, blk_ncolors, x_out, ld, (T*)nullptr);
}

// Residual of a blocked level RIGHT AFTER a block-hybrid sweep x_old -> x_new, from the sweep's own explicit part: the sweep solved
//   a_ii x_new_i + sum_{L(i)} a_ij x_new_j + sum_{E(i)} a_ij x_old_j = b_i        (L: earlier rows of the block, E: everything else)
// for every row, so  r_i = b_i - (A x_new)_i = sum_{E(i)} a_ij (x_old_j - x_new_j):  the residual needs neither b nor the diagonal
// nor the lower entries -- 55 % of the operator's entries on the 506 k-row level of the 3 M-vertex torus, on the UNPADDED E storage
// (the residual SpMV reads the merged SELL operator, 25-29 % padding: 31 us; this: see profiles/).  The value differs from
// b - A x_new by the rounding of the sweep's last division, |a_ii x_i| eps -- the size of the rounding error of evaluating b - A x
// itself.  x_old == nullptr: the sweep started from the zero vector.  Launch geometry and entry-parallel reduction as in gs_block_ep.
template <class T, int D, bool STREAM>
__global__ __launch_bounds__(64) void residual_delta_ep(const int* __restrict__ blk_begin, const int* __restrict__ e_ptr, const int* __restrict__ e_col,
                                                        const T* __restrict__ e_val, const T* __restrict__ x_old, const T* __restrict__ x_new,
                                                        T* __restrict__ r, int ld, int n_blocks) {
    extern __shared__ unsigned char smem_raw[];
    T* zeroT = reinterpret_cast<T*>(smem_raw);                         // the zero region, then the products (ep_lds_bytes(0, cap_e, 0))
    T* pbuf = zeroT + kEpZeroBytes / sizeof(T);
    const int chunk = (int)(gridDim.x >> 3);
    const int blk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3));
    if (blk >= n_blocks) return;
    const int lane = threadIdx.x;
    reinterpret_cast<int*>(zeroT)[lane] = 0;
    const int r0 = __builtin_amdgcn_readfirstlane(blk_begin ? blk_begin[blk] : blk << 6);          // a block table (a rank's blocks), or the level's own blocks in order
    const int row = r0 + lane;
    const int e0 = e_ptr[r0], e1 = e_ptr[r0 + 64];
    const int nE = e1 - e0;
    // (chunk loads through buffer resources, unguarded register window, zero-slot reads: as in gs_block_ep)
    const __amdgpu_buffer_rsrc_t rc = ep_chunk(e_col + e0, nE * 4), rv = ep_chunk(e_val + e0, nE * (int)sizeof(T));
    int ec[kEpE];
    T ev[kEpE];
#pragma unroll
    for (int k = 0; k < kEpE; ++k) { ec[k] = ep_load<int, STREAM>(rc, lane * 4 + 256 * k); ev[k] = ep_load<T, STREAM>(rv, lane * (int)sizeof(T) + 64 * k * (int)sizeof(T)); }
    const int eb = e_ptr[row] - e0, ee = e_ptr[row + 1] - e0;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const T* xn = x_new + (int64_t)c * ld;
        const __amdgpu_buffer_rsrc_t rn = ep_chunk(xn, ld * (int)sizeof(T));
        if (x_old) {
            const T* xo = x_old + (int64_t)c * ld;
            const __amdgpu_buffer_rsrc_t ro = ep_chunk(xo, ld * (int)sizeof(T));
            T go[kEpE], gn[kEpE];
#pragma unroll
            for (int k = 0; k < kEpE; ++k) { go[k] = ep_load<T, false>(ro, ec[k] * (int)sizeof(T)); gn[k] = ep_load<T, false>(rn, ec[k] * (int)sizeof(T)); }
#pragma unroll
            for (int k = 0; k < kEpE; ++k) pbuf[64 * k + lane] = ev[k] * (go[k] - gn[k]);
            if (nE > 64 * kEpE)
                for (int e = 64 * kEpE + lane; e < nE; e += 64) { const int cj = e_col[e0 + e]; pbuf[e] = e_val[e0 + e] * (xo[cj] - xn[cj]); }
        } else {
            T gn[kEpE];
#pragma unroll
            for (int k = 0; k < kEpE; ++k) gn[k] = ep_load<T, false>(rn, ec[k] * (int)sizeof(T));
#pragma unroll
            for (int k = 0; k < kEpE; ++k) pbuf[64 * k + lane] = -(ev[k] * gn[k]);
            if (nE > 64 * kEpE)
                for (int e = 64 * kEpE + lane; e < nE; e += 64) pbuf[e] = -(e_val[e0 + e] * xn[e_col[e0 + e]]);
        }
        __syncthreads();
        T acc = (T)0.0;
        const T* pa = pbuf + eb;
        int rem = ee - eb;
        while (__builtin_amdgcn_ballot_w64(rem > 0)) {
            T p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = (rem > j ? pa : zeroT)[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += p[j];
            pa += 8; rem -= 8;
        }
        r[row + (int64_t)c * ld] = acc;
        __syncthreads();
    }
}

// The same sweep on the QUAD layout (4 lanes per row; blocks of <= 256 rows = 1024 threads).  A colour step is the
// critical path of the coarse levels -- 13-16 of them run back to back with one or two wavefronts active -- so the
// row is spread over four lanes: each lane keeps <= WQ in-block entries in registers, gathers them from LDS in one
// batch, and the quad adds its four partial sums with two cross-lane steps.  Same mathematics as gs_block.
constexpr int kQuadBlockRows = 256;
// FR (restriction fused, like restrict_sweep0 for the entry-parallel levels): the level's first pre-sweep starts from zero (x_in == nullptr), and a wave's
// 16 rows are exactly one slice of the restriction INTO this level (quad layout, rows sorted inside 64-row windows, blocks are multiples of 64
// rows): the wave first forms those right-hand sides (transfer<.., 0, 4>'s arithmetic), writes them to b_out and leaves them in LDS for the sweep.
template <class T, int D, int WQ, bool FR = false>
__global__ __launch_bounds__(kBlockRows) void gs_block4(const int* __restrict__ blk_begin, const int* __restrict__ blk_ncolors,
                                                        const unsigned char* __restrict__ row_color,
                                                        const int64_t* __restrict__ in_ptr, const unsigned short* __restrict__ in_col,
                                                        const T* __restrict__ in_val, const int64_t* __restrict__ out_ptr,
                                                        const int* __restrict__ out_col, const T* __restrict__ out_val,
                                                        const T* __restrict__ diag, const T* __restrict__ b,
                                                        const T* __restrict__ x_in, T* __restrict__ x_out, int ld,
                                                        const int64_t* __restrict__ r_slice_ptr = nullptr, const int* __restrict__ r_col = nullptr,
                                                        const T* __restrict__ r_val = nullptr, const int* __restrict__ r_row_of = nullptr,
                                                        const T* __restrict__ r_fine = nullptr, int ldx = 0, T* __restrict__ b_out = nullptr) {
    __shared__ T xs[D][kQuadBlockRows];
    __shared__ T bs[FR ? D : 1][FR ? kQuadBlockRows : 1];
    const int blk = blockIdx.x;
    const int r0 = blk_begin[blk];
    const int nrows = blk_begin[blk + 1] - r0;            // multiple of 64, <= 256
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool active = wave * 16 < nrows;                 // wave-uniform: a wave covers 16 rows
    const int lrow = t >> 2;
    const int row = r0 + lrow;
    const bool writer = (t & 3) == 0;
    T rhs[D], dg = 1.0;
    int mycolor = -1, w = 0;
    int64_t p0 = 0;
#pragma unroll
    for (int c = 0; c < D; ++c) rhs[c] = 0.0;
    T v[WQ];
    unsigned cpk[WQ / 2];
#pragma unroll
    for (int j = 0; j < WQ; ++j) v[j] = 0.0;
#pragma unroll
    for (int j = 0; j < WQ / 2; ++j) cpk[j] = 0u;
    if (active) {
        const int s = (r0 >> 4) + wave;
        if (writer) {
#pragma unroll
            for (int c = 0; c < D; ++c) xs[c][lrow] = x_in ? x_in[row + (int64_t)c * ld] : (T)0.0;
        }
        p0 = in_ptr[s];
        w = (int)((in_ptr[s + 1] - p0) >> 6);
#pragma unroll
        for (int j = 0; j < WQ; ++j)
            if (j < w) {
                // (ordinary loads: the quad layout is for levels of < 65 536 rows, whose operator stays on the chip from one launch to the next)
                v[j] = in_val[p0 + (int64_t)j * 64 + lane];
                cpk[j >> 1] |= (unsigned)in_col[p0 + (int64_t)j * 64 + lane] << ((j & 1) * 16);
            }
        T acc[D];
        if (x_in) { row_dot<T, D>(out_ptr, out_col, out_val, x_in, ld, s, lane, acc); quad_reduce<T, D>(acc); }
        else {
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = 0.0;
        }
        if constexpr (!FR) {
#pragma unroll
            for (int c = 0; c < D; ++c) rhs[c] = b[row + (int64_t)c * ld] - acc[c];
        }
        dg = 1.0 / diag[row];
        mycolor = row_color[row];
    }
    if constexpr (FR) {
        if (active && writer) {
#pragma unroll
            for (int c = 0; c < D; ++c) bs[c][lrow] = (T)0.0;          // (padding rows of the block: right-hand side zero)
        }
        __syncthreads();
        if (active) {
            const int s = (r0 >> 4) + wave;
            T racc[D];
            row_dot_sel<T, D, 0, 8, 0>(r_slice_ptr, r_col, (const unsigned*)nullptr, (const int*)nullptr, 0, r_val, r_fine, ldx, s, lane, racc);
            quad_reduce<T, D>(racc);
            if (writer) {
                const int srow = s * 16 + (lane >> 2);
                const int orow = r_row_of ? r_row_of[srow] : srow;
                if (orow >= 0) {
#pragma unroll
                    for (int c = 0; c < D; ++c) { b_out[orow + (int64_t)c * ld] = racc[c]; bs[c][orow - r0] = racc[c]; }
                }
            }
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int c = 0; c < D; ++c) rhs[c] = bs[c][lrow];
        }
    }
    __syncthreads();
    const int nc = blk_ncolors[blk];
    for (int col = 0; col < nc; ++col) {
        if (mycolor == col) {
            T s_[D];
            T xv[WQ][D];
#pragma unroll
            for (int j = 0; j < WQ; ++j) {
                const int cj = (cpk[j >> 1] >> ((j & 1) * 16)) & 0xffff;
#pragma unroll
                for (int c = 0; c < D; ++c) xv[j][c] = xs[c][cj];
            }
#pragma unroll
            for (int c = 0; c < D; ++c) s_[c] = 0.0;
#pragma unroll
            for (int j = 0; j < WQ; ++j)
#pragma unroll
                for (int c = 0; c < D; ++c) s_[c] += v[j] * xv[j][c];
            for (int j = WQ; j < w; ++j) {                  // rows with more than 4*WQ in-block entries (rare)
                const T vj = in_val[p0 + (int64_t)j * 64 + lane];
                const int cj = in_col[p0 + (int64_t)j * 64 + lane];
#pragma unroll
                for (int c = 0; c < D; ++c) s_[c] += vj * xs[c][cj];
            }
            quad_reduce<T, D>(s_);
            if (writer) {
#pragma unroll
                for (int c = 0; c < D; ++c) xs[c][lrow] = (rhs[c] - s_[c]) * dg;
            }
        }
        __syncthreads();
    }
    if (active && writer) {
#pragma unroll
        for (int c = 0; c < D; ++c) x_out[row + (int64_t)c * ld] = xs[c][lrow];
    }
}

// Weighted Jacobi sweep: x_out = x_in + omega * (b - A x_in) / diag.
template <class T, int D>
__global__ __launch_bounds__(kBlock) void jacobi_sweep(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                       const T* __restrict__ val, const T* __restrict__ diag,
                                                       const T* __restrict__ b, const T* __restrict__ x_in,
                                                       T* __restrict__ x_out, int ld, int n_slices, T omega,
                                                       int xcd_swizzle) {
    const int s = wave_slice(n_slices, xcd_swizzle);
    if (s >= n_slices) return;
    const int lane = threadIdx.x & 63;
    const int row = s * 64 + lane;
    T acc[D];
    row_dot<T, D>(slice_ptr, col, val, x_in, ld, s, lane, acc);
    const T dg = diag[row];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const T xi = x_in[row + (int64_t)c * ld];
        x_out[row + (int64_t)c * ld] = xi + omega * ((b[row + (int64_t)c * ld] - acc[c]) / dg - xi);
    }
}

// MODE 0: y = A x      MODE 1: y = b - A x   (gravomg/src/multigrid_solver.cpp:1066)
// LPR = lanes per row of the SELL layout (1, or 4 on the coarse levels): slices then hold 64 / LPR rows.
template <class T, int D, int MODE, int LPR, int C16 = 0, int YI = 0>
__device__ __forceinline__ void spmv_full_slice(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col, const T* __restrict__ val,
                                                const T* __restrict__ diag, const T* __restrict__ b, const T* __restrict__ x, T* __restrict__ y,
                                                int ld, int s, const unsigned* __restrict__ col16 = nullptr, const int* __restrict__ win_base = nullptr, int c16_arg = 0) {
    const int lane = threadIdx.x & 63;
    const int row = s * (64 / LPR) + lane / LPR;
    T acc[D];
    row_dot_sel<T, D, C16>(slice_ptr, col, col16, win_base, c16_arg, val, x, ld, s, lane, acc);
    if (LPR == 4) { quad_reduce<T, D>(acc); if (lane & 3) return; }
    const T dg = diag[row];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const T ax = acc[c] + dg * x[row + (int64_t)c * ld];
        y[YI ? (int64_t)row * D + c : row + (int64_t)c * ld] = MODE == 1 ? b[row + (int64_t)c * ld] - ax : ax;
    }
}
template <class T, int D, int MODE, int LPR, int C16 = 0, int YI = 0>
__global__ __launch_bounds__(kBlock) void spmv_full(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                    const T* __restrict__ val, const T* __restrict__ diag,
                                                    const T* __restrict__ b, const T* __restrict__ x,
                                                    T* __restrict__ y, int ld, int slice_begin, int slice_end,
                                                    int xcd_swizzle, const unsigned* __restrict__ col16 = nullptr, const int* __restrict__ win_base = nullptr, int c16_arg = 0) {
    const int s = slice_begin + wave_slice(slice_end - slice_begin, xcd_swizzle);
    if (s >= slice_end) return;
    spmv_full_slice<T, D, MODE, LPR, C16, YI>(slice_ptr, col, val, diag, b, x, y, ld, s, col16, win_base, c16_arg);
}
// the same over a LIST of slices (a rank's rows of a level partitioned by blocks: not one contiguous range)
template <class T, int D, int MODE, int LPR>
__global__ __launch_bounds__(kBlock) void spmv_full_list(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                         const T* __restrict__ val, const T* __restrict__ diag,
                                                         const T* __restrict__ b, const T* __restrict__ x,
                                                         T* __restrict__ y, int ld, const int* __restrict__ slices, int n_list) {
    const int i = wave_slice(n_list, 1);
    if (i >= n_list) return;
    spmv_full_slice<T, D, MODE, LPR>(slice_ptr, col, val, diag, b, x, y, ld, __builtin_amdgcn_readfirstlane(slices[i]));
}

// Transfer operators.  ADD = 0: y[out_row] = sum val * x[col]   (restriction rc = U^T r, :1069)
//                       ADD = 1: y[out_row] += sum val * x[col]  (prolongation x += U e, :1082)
// row_of (may be null) maps the slice row to the output row (-1 = none); ldx/ldy are the leading
// dimensions of the source / destination level.  LPR as in spmv_full.
template <class T, int D, int ADD, int LPR, int C16 = 0, int XI = 0>
__device__ __forceinline__ void transfer_slice(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col, const T* __restrict__ val,
                                               const int* __restrict__ row_of, const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int s,
                                               const unsigned* __restrict__ col16 = nullptr, const int* __restrict__ win_base = nullptr, int c16_arg = 0) {
    const int lane = threadIdx.x & 63;
    const int srow = s * (64 / LPR) + lane / LPR;
    T acc[D];
    // (quad layout = restrictions: a lane holds 4-5 of a row's ~18 entries -- one group of 8 in flight instead of 4 + a dependent tail)
    row_dot_sel<T, D, C16, (LPR == 4 ? 8 : DotGroup<D>::value), XI>(slice_ptr, col, col16, win_base, c16_arg, val, x, ldx, s, lane, acc);
    if (LPR == 4) { quad_reduce<T, D>(acc); if (lane & 3) return; }
    // ADD = 1 is a read-modify-write of y: only safe when every output row is produced by exactly one slice row, so that
    // instantiation never takes an output-row map (slice row == output row: unique by construction)
    const int row = (!ADD && row_of) ? row_of[srow] : srow;
    if (row < 0) return;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        if (ADD) y[row + (int64_t)c * ldy] += acc[c];
        else y[row + (int64_t)c * ldy] = acc[c];
    }
}
template <class T, int D, int ADD, int LPR, int C16 = 0, int XI = 0>
__global__ __launch_bounds__(kBlock) void transfer(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                   const T* __restrict__ val, const int* __restrict__ row_of,
                                                   const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                   int slice_begin, int slice_end, int xcd_swizzle, const unsigned* __restrict__ col16 = nullptr,
                                                   const int* __restrict__ win_base = nullptr, int c16_arg = 0) {
    const int s = slice_begin + wave_slice(slice_end - slice_begin, xcd_swizzle);
    if (s >= slice_end) return;
    transfer_slice<T, D, ADD, LPR, C16, XI>(slice_ptr, col, val, row_of, x, ldx, y, ldy, s, col16, win_base, c16_arg);
}
template <class T, int D, int ADD, int LPR>
__global__ __launch_bounds__(kBlock) void transfer_list(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                        const T* __restrict__ val, const int* __restrict__ row_of,
                                                        const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                        const int* __restrict__ slices, int n_list) {
    const int i = wave_slice(n_list, 1);
    if (i >= n_list) return;
    transfer_slice<T, D, ADD, LPR>(slice_ptr, col, val, row_of, x, ldx, y, ldy, __builtin_amdgcn_readfirstlane(slices[i]));
}

// ---- publishing data to another agent (a peer GPU's mailbox, pinned host memory) behind a sequence word ------------------------------
// Portable form (fenced != 0; the default of the multi-GPU entry points): a system-scope RELEASE fence between the data stores and the
// store of the sequence word, a system-scope ACQUIRE fence after the poll that saw it -- the HIP memory model's publication idiom.
// gfx942 / gfx950 form (fenced == 0; opt-in: gmg_p2p_set_fences / GMG_P2P_FENCE_FREE, GMG_PUBLISH_FENCE_FREE): every datum is a
// system-scope (sc0 sc1, write-through) atomic store, so nothing of it sits dirty in an L2 that a release would have to write back
// (buffer_wbl2 -- which also writes back the sweep's own dirty lines: ~3.5 us per exchange launch); `s_waitcnt vmcnt(0)` is the part of
// the release sequence that orders the stores ahead of the word (the LLVM AMDGPU memory model for gfx940+: stores are counted in vmcnt
// and complete at the memory side before it reaches zero; on gfx10+ they count in vscnt -- hence the architecture check); readers take
// the data with system-scope atomic loads, which no cache serves.  Anything else than these two targets does not compile this form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "kernels.hip.hpp: the fence-free publication sequence (publish_order) relies on gfx942 / gfx950 store completion semantics"
#endif
__device__ __forceinline__ void publish_order(int fenced) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
}
__device__ __forceinline__ void consume_order(int fenced) {
    if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// out[c] = sum over blocks of partials[block][c], c < ncomp <= kReduceMaxComp.  One block of kReduceBlock threads, fixed order
// (thread-strided sums, wave shuffles, then the 16 wave sums in index order): one pass over the partials and one barrier.
constexpr int kReduceBlock = 1024;
constexpr int kReduceMaxComp = 8;
__device__ __forceinline__ void block_reduce_partials(const double* __restrict__ partials, int n_blocks, int ncomp, double* __restrict__ out,
                                                      double (*red)[kReduceMaxComp]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v[kReduceMaxComp];
#pragma unroll
    for (int c = 0; c < kReduceMaxComp; ++c) v[c] = 0.0;
    // (four strides of partials in flight per thread -- the loads of a stride are independent, the additions keep the order i, i + 1024, ...: at
    // d = 3 the check's 5 000 x 6 sums took five dependent round trips, 12.5 us of a 1 ms cycle)
    for (int i0 = threadIdx.x; i0 < n_blocks; i0 += 4 * kReduceBlock) {
        double t[4][kReduceMaxComp];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * kReduceBlock;
#pragma unroll
            for (int c = 0; c < kReduceMaxComp; ++c) t[u][c] = (c < ncomp && i < n_blocks) ? partials[(int64_t)i * ncomp + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < kReduceMaxComp; ++c) v[c] += t[u][c];
    }
#pragma unroll
    for (int c = 0; c < kReduceMaxComp; ++c) {
        if (c >= ncomp) break;
        double t = v[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
        if (lane == 0) red[wave][c] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < ncomp) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kReduceBlock / 64; ++w) t += red[w][threadIdx.x];
        // (system scope: a write-through store -- `out` may be host memory that the host polls for, reduce_partials)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + threadIdx.x), (unsigned long long)__double_as_longlong(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// flag != nullptr: `out` is host-visible pinned memory and `seq` is published in *flag after the sums -- the host polls
// that word instead of paying a copy kernel and a stream synchronisation per residual check.  (The sums are written by the
// threads of wave 0; lane 0 of that wave releases them: one system-scope fence, not one per thread.)
// The solve loop's decision, taken where the sums are (`watch.go` != nullptr): the host's arithmetic (engine_cycle.hip.hpp::norm_from_sums and the
// loop of engine.hip::solve_common: sqrt and division are correctly rounded on both sides), so that a launch enqueued behind this one can be
// told whether the iteration goes on.  The host does not decide a second time: it reads the word this kernel publishes beside the sums.
struct SolveWatch {
    int* go;                    // device word the speculative launch reads (nullptr: no decision wanted)
    unsigned long long* host_go;  // the same decision for the host (pinned, beside the sequence word)
    double* least;              // device: smallest residue of this solve so far
    double tol;
    int mode;                   // 0: always go on (a fixed number of cycles, gmg_run_cycles); 1: the solve loop's test
    int norm_type, d, cycles_done;
};
__global__ __launch_bounds__(kReduceBlock) void reduce_partials(const double* __restrict__ partials, int n_blocks, int ncomp,
                                                                double* __restrict__ out, unsigned long long* flag, unsigned long long seq, int fenced,
                                                                SolveWatch watch = SolveWatch{nullptr, nullptr, nullptr, 0.0, 0, 0, 0, 0}) {
    __shared__ double red[kReduceBlock / 64][kReduceMaxComp];
    block_reduce_partials(partials, n_blocks, ncomp, out, red);
    if (watch.go != nullptr && threadIdx.x == 0) {
        int go = 1;
        if (watch.mode == 1) {
            double s[kReduceMaxComp];
            for (int c = 0; c < ncomp; ++c) {                              // (the sums as block_reduce_partials formed them: same order)
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < kReduceBlock / 64; ++w) t += red[w][c];
                s[c] = t;
            }
            double res = 0.0;
            if (watch.norm_type == 3) {
                double t = 0.0;
                for (int c = 0; c < watch.d; ++c) t += s[2 * c];
                res = __builtin_sqrt(t);
            } else {
                for (int c = 0; c < watch.d; ++c) {
                    const double v = watch.norm_type == 0 ? __builtin_sqrt(s[2 * c]) / __builtin_sqrt(s[2 * c + 1]) : __builtin_sqrt(s[2 * c] / s[2 * c + 1]);
                    if (c == 0 || v > res) res = v;
                }
            }
            double least = watch.cycles_done <= 1 ? res : *watch.least;
            if (res < least) least = res;
            *watch.least = least;
            const bool blown = !__builtin_isfinite(res) || (watch.cycles_done >= 3 && res > 1e4 * least);
            go = (res > watch.tol && !blown) ? 1 : 0;
        }
        *watch.go = go;
        __hip_atomic_store(watch.host_go, (unsigned long long)go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (flag && threadIdx.x < 64) {
        // the sums went out as write-through stores: drained, they are ahead of the sequence word (no system fence: that writes the L2 back and
        // invalidates it, microseconds per residual check; publish_order)
        publish_order(fenced);
        if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// halo exchange (multi-GPU): pack / unpack of the published entries
__global__ void gather_entries(const double* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
__global__ void scatter_entries(const double* __restrict__ src, const int64_t* __restrict__ pos, const int64_t* __restrict__ idx, int64_t n,
                                double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[idx[i]] = src[pos[i]];
}

// ---- multi-GPU, device-initiated exchange (one process per GPU; engine_dist.hip.hpp) -------------------------------------
// Every rank owns a MAILBOX (fine-grained device memory, mapped into the other ranks' address spaces through IPC handles)
// with one region per (source rank, exchange kind) and one 64-bit arrival counter per source rank.  An exchange is ONE small
// launch on every rank: block j < n_peers PUSHES -- it stores this rank's values for peer j straight into j's mailbox over
// xGMI, fences at system scope and then publishes the exchange's sequence number in j's counter; block n_peers + j PULLS --
// lane 0 waits until peer j's sequence number has arrived in the local counter (system-scope acquire), then the block copies
// j's region into the local vector.  No collective-library call, no host involvement, ~one xGMI round trip of latency.
// Both directions run every time (with or without payload), so a region is never overwritten before its reader is done with
// it: a rank pushes exchange s + 1 only after it pulled exchange s from that peer, which the peer sent after its own pull.
struct P2POp {
    const int* send_idx;            // local vector entries this rank publishes to the peer (null: contiguous range send_lo ..)
    int n_send, send_lo;
    double* remote_box;             // where they go: the peer's mailbox region for (this rank, this kind), peer-mapped
    unsigned long long* remote_flag;     // the peer's arrival counter for this rank, peer-mapped
    const int* recv_idx;            // where the peer's values go in the local vector (null: contiguous range recv_lo ..)
    int n_recv, recv_lo;
    const double* local_box;        // this rank's mailbox region for (peer, this kind)
    unsigned long long* local_flag; // this rank's arrival counter for the peer
};

// How long a pull waits for its peer, in ticks of the 100 MHz wall clock (default 4 s; engine_dist.hip.hpp sets it from
// GMG_P2P_TIMEOUT_S at gmg_p2p_prepare: first-use code-object loads, paging or a debugger can skew ranks by more than that).
__device__ unsigned long long g_p2p_timeout_ticks = 400000000ull;

// vec: D columns with leading dimension ld.  err (device int): set to 1 when a wait timed out (g_p2p_timeout_ticks of wall clock).
// Grid: 2 * n_peers * B blocks -- per peer B blocks push and B blocks pull, each its strided share of the values (B = 1 for a halo
// of a few thousand entries; a whole vector moves with tens of blocks: one block's stores do not fill an xGMI link).  The last
// push block to finish (done[peer], device memory, zero between launches) publishes the sequence number; no block of this
// launch waits for another block of it, only for the PEER's push.
__global__ __launch_bounds__(256) void p2p_exchange(const P2POp* __restrict__ ops, int n_peers, double* vec, int ld, int D,
                                                    unsigned long long seq, int* err, int B, unsigned int* done, int fenced) {
    const int part = blockIdx.x % B, j = (blockIdx.x / B) % n_peers;
    const bool push = (int)blockIdx.x < n_peers * B;
    const P2POp op = ops[j];
    const int64_t stride = (int64_t)B * blockDim.x;
    if (push) {                                                       // ---- push
        const int64_t total = (int64_t)op.n_send * D;
        for (int64_t i = (int64_t)part * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int c = (int)(i / op.n_send), k = (int)(i - (int64_t)c * op.n_send);
            const int src = op.send_idx ? op.send_idx[k] : op.send_lo + k;
            // write-through stores, drained below: nothing of this exchange sits in an L2 that a release fence would have to write back
            // (buffer_wbl2 + buffer_inv at system scope: ~3.5 us of every exchange launch)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(op.remote_box + i), (unsigned long long)__double_as_longlong(vec[src + (int64_t)c * ld]),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        publish_order(fenced);
        __syncthreads();
        if (threadIdx.x == 0) {
            bool last = true;
            if (B > 1) {
                // (fenced: the tickets carry the happens-before of the blocks that finished earlier to the one that publishes)
                last = (fenced ? __hip_atomic_fetch_add(done + j, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                               : __hip_atomic_fetch_add(done + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == (unsigned)B - 1;
                if (last) __hip_atomic_store(done + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (last) {
                if (fenced) __hip_atomic_store(op.remote_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                else __hip_atomic_store(op.remote_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    } else {                                                          // ---- pull
        __shared__ int timed_out;
        if (threadIdx.x == 0) {
            timed_out = 0;
            const unsigned long long t0 = wall_clock64();            // 100 MHz constant clock
            // (an exchange that already timed out means the peer is gone: the launches still queued behind it give up at once)
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) timed_out = 1;
            // (relaxed polls; the values are read with system-scope loads, which no cache serves)
            while (!timed_out && __hip_atomic_load(op.local_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > g_p2p_timeout_ticks) { timed_out = 1; atomicExch(err, 1); break; }
            }
            consume_order(fenced);          // (relaxed while spinning, one acquire after the poll that saw the number)
        }
        __syncthreads();
        if (timed_out) return;
        const int64_t total = (int64_t)op.n_recv * D;
        const unsigned long long* box = reinterpret_cast<const unsigned long long*>(op.local_box);
        for (int64_t i = (int64_t)part * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int c = (int)(i / op.n_recv), k = (int)(i - (int64_t)c * op.n_recv);
            const int dst = op.recv_idx ? op.recv_idx[k] : op.recv_lo + k;
            // system-scope loads: the region is rewritten by the peer every exchange, no stale cached copy may be served
            const unsigned long long bits = __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            vec[dst + (int64_t)c * ld] = __longlong_as_double((long long)bits);
        }
    }
}

// ---- the exchange of a colour FOLDED into that colour's sweep (gmg_p2p_set_smoother mode 2) ----------------------------------
// No exchange launch: a wave of gs_color_push that has updated its 64 rows stores the ones peers read straight from its registers into the
// peers' mailboxes (pub_ent: the entries of all send lists that fall into the slice), fences at system scope and counts itself done; the LAST
// publishing wave of the launch publishes the sequence number to every peer and then PULLS: waits for every peer's number and copies the
// peers' values into x.  The pull may overlap the rest of the launch: it writes x entries of the peers' rows of THIS colour, which no row
// of this colour reads (that is what a colouring is), and the launch boundary orders it before the next colour's gathers.  A launch without
// any published row (pt.n_pub_waves == 0) hands both jobs to its first wave.  Region reuse is as in p2p_exchange: a rank's pull of exchange
// m is inside its launch m, which precedes its launch (= push) m + 1 in the stream.
struct PushTail {
    const int* pub_ptr;             // [slices of the launch + 1]: entries of pub_ent per slice
    const int2* pub_ent;            // (row, peer << 24 | position in that peer's send list), rows ascending
    const P2POp* ops;               // one per peer: the colour's exchange (kind, parity)
    int n_peers, n_pub_waves;       // n_pub_waves: slices with at least one entry
    unsigned int* done;             // publishing waves that have finished (device memory, zero between launches)
    unsigned long long seq;
    int* err;
    int fenced;                     // publish_order / consume_order
};

template <int D, int FINE>
__global__ __launch_bounds__(kBlock) void gs_color_push(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col, const double* __restrict__ val,
                                                        const double* __restrict__ diag, const double* __restrict__ b, double* x, int ld, int slice_begin,
                                                        int slice_end, double omega, const unsigned* __restrict__ col16, const int* __restrict__ win_base,
                                                        int c16_arg, PushTail pt) {
    const int s = slice_begin + wave_slice(slice_end - slice_begin, 1);
    const int lane = threadIdx.x & 63;
    int finish = pt.n_pub_waves == 0 && blockIdx.x == 0 && threadIdx.x < 64;
    if (s < slice_end) {
        const int row = s * 64 + lane;
        double acc[D], xn[D];
        row_dot_sel<double, D, (FINE >= 2 ? FINE - 1 : 0)>(slice_ptr, col, col16, win_base, c16_arg, val, x, ld, s, lane, acc);
        const double dg = diag[row];
        if (omega == 1.0) {
#pragma unroll
            for (int c = 0; c < D; ++c) { xn[c] = (b[row + (int64_t)c * ld] - acc[c]) / dg; x[row + (int64_t)c * ld] = xn[c]; }
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const double xi = x[row + (int64_t)c * ld];
                xn[c] = xi + omega * ((b[row + (int64_t)c * ld] - acc[c]) / dg - xi);
                x[row + (int64_t)c * ld] = xn[c];
            }
        }
        const int q0 = pt.pub_ptr[s - slice_begin], q1 = pt.pub_ptr[s - slice_begin + 1];       // wave-uniform
        if (q1 > q0) {
            for (int q = q0; q < q1; q += 64) {
                const bool valid = q + lane < q1;
                const int2 e = valid ? pt.pub_ent[q + lane] : make_int2(row, 0);
#pragma unroll
                for (int c = 0; c < D; ++c) {
                    const double v = __shfl(xn[c], e.x & 63, 64);
                    // write-through stores (no L2 line to write back later: a release FENCE here would write back and invalidate the XCD's L2 in the
                    // middle of the sweep -- measured: +8 us per colour launch)
                    if (valid) {
                        const P2POp& op = pt.ops[e.y >> 24];
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(op.remote_box + (e.y & 0xffffff) + (int64_t)c * op.n_send),
                                           (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            publish_order(pt.fenced);      // the stores have arrived before this wave counts itself done
            int last = 0;
            if (lane == 0) last = (pt.fenced ? __hip_atomic_fetch_add(pt.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_fetch_add(pt.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == (unsigned)pt.n_pub_waves - 1u;
            finish = __builtin_amdgcn_readfirstlane(last);
        }
    }
    if (!finish) return;
    // ---- exactly one wave of the launch: publish the sequence number, then pull
    if (lane == 0) __hip_atomic_store(pt.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (every publishing wave drained its write-through stores before it took its ticket, and this wave drew the last one: the number may follow)
    if (lane < pt.n_peers) {
        if (pt.fenced) __hip_atomic_store(pt.ops[lane].remote_flag, pt.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        else __hip_atomic_store(pt.ops[lane].remote_flag, pt.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int j = 0; j < pt.n_peers; ++j) {
        const P2POp op = pt.ops[j];
        int timed_out = 0;
        if (lane == 0) {
            const unsigned long long t0 = wall_clock64();
            if (__hip_atomic_load(pt.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) timed_out = 1;
            // relaxed polls (an acquire per poll would invalidate caches under the waves still sweeping); the values are read with system-scope
            // loads below, which no cache serves
            while (!timed_out && __hip_atomic_load(op.local_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < pt.seq) {
                __builtin_amdgcn_s_sleep(4);
                if (wall_clock64() - t0 > g_p2p_timeout_ticks) { timed_out = 1; atomicExch(pt.err, 1); break; }
            }
        }
        if (__builtin_amdgcn_readfirstlane(timed_out)) return;
        consume_order(pt.fenced);
        const int64_t total = (int64_t)op.n_recv * D;
        const unsigned long long* box = reinterpret_cast<const unsigned long long*>(op.local_box);
        // (one wave copies a few hundred values: eight independent system-scope loads per lane in flight, then the stores)
        for (int64_t i0 = lane; i0 < total; i0 += 64 * 8) {
            unsigned long long bits[8];
            int dst[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = i0 + 64 * u;
                const bool in = i < total;
                const int c = in ? (int)(i / op.n_recv) : 0, k = in ? (int)(i - (int64_t)c * op.n_recv) : 0;
                dst[u] = in ? op.recv_idx[k] + c * ld : -1;
                bits[u] = in ? __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (dst[u] >= 0) x[dst[u]] = __longlong_as_double((long long)bits[u]);
        }
    }
}

// All-reduce of a few doubles (the residual-norm sums): every rank pushes its values to every peer, then adds the world
// contributions in RANK order -- the same sum, bit for bit, on every rank.  One block; peers in ascending rank order.
__global__ __launch_bounds__(64) void p2p_allreduce_small(const P2POp* __restrict__ ops, int n_peers, int rank, const double* __restrict__ mine,
                                                          int n, double* __restrict__ out, unsigned long long seq, int* err) {
    const int t = threadIdx.x;
    __shared__ int timed_out;
    if (t == 0) timed_out = 0;
    for (int j = 0; j < n_peers; ++j)
        if (t < n) ops[j].remote_box[t] = mine[t];
    __threadfence_system();
    __syncthreads();
    if (t < n_peers) {
        __hip_atomic_store(ops[t].remote_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) timed_out = 1;
        while (!timed_out && __hip_atomic_load(ops[t].local_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > g_p2p_timeout_ticks) { timed_out = 1; atomicExch(err, 1); break; }
        }
    }
    __syncthreads();
    if (timed_out) return;
    if (t < n) {
        double sum = 0.0;
        for (int r = 0; r <= n_peers; ++r) {                          // world = n_peers + 1 contributions, rank order
            if (r == rank) { sum += mine[t]; continue; }
            const unsigned long long* box = reinterpret_cast<const unsigned long long*>(ops[r < rank ? r : r - 1].local_box);
            sum += __longlong_as_double((long long)__hip_atomic_load(box + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        }
        out[t] = sum;
    }
}

// ---- multi-GPU, collective exchange: pack -> all-gather -> unpack, enqueued by the engine (engine_dist.hip.hpp, gmg_config::dist_exchange) --
// The north star's exchange -- an all-gather of the x halo -- as three launches on the engine's stream: every rank packs what its peers read
// of the vector it just updated into ONE chunk (a segment per destination rank; or its own rows once, for the whole-rows kinds), the chunks of
// all ranks are all-gathered (ncclAllGather over xGMI; between processes that share a device: stores into the peers' buffers through hipIpc
// mappings, gmgk::p2p_exchange on contiguous ops), and every rank unpacks the segments addressed to it.  A segment holds D columns of n
// entries, column-major.
struct CollSeg {
    const int* idx;                 // local vector entries (device numbering) of the segment
    int n;
    long long off;                  // where the segment starts: doubles from the start of the chunk buffer (pack) / of the gathered buffer (unpack)
};

__global__ __launch_bounds__(256) void coll_pack(const CollSeg* __restrict__ segs, int n_segs, const double* __restrict__ vec, int ld, int D,
                                                 double* __restrict__ chunk, int B) {
    const int part = blockIdx.x % B, j = blockIdx.x / B;
    if (j >= n_segs) return;
    const CollSeg sg = segs[j];
    const int64_t total = (int64_t)sg.n * D, stride = (int64_t)B * blockDim.x;
    for (int64_t i = (int64_t)part * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i / sg.n), k = (int)(i - (int64_t)c * sg.n);
        chunk[sg.off + i] = vec[sg.idx[k] + (int64_t)c * ld];
    }
}

// SYS: the gathered buffer was written by OTHER processes through peer mappings (the emulated collective): system-scope loads, no stale line
template <bool SYS>
__global__ __launch_bounds__(256) void coll_unpack(const CollSeg* __restrict__ segs, int n_segs, const double* __restrict__ gathered, double* __restrict__ vec,
                                                   int ld, int D, int B) {
    const int part = blockIdx.x % B, j = blockIdx.x / B;
    if (j >= n_segs) return;
    const CollSeg sg = segs[j];
    const int64_t total = (int64_t)sg.n * D, stride = (int64_t)B * blockDim.x;
    for (int64_t i = (int64_t)part * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i / sg.n), k = (int)(i - (int64_t)c * sg.n);
        double v;
        if (SYS) v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(gathered) + sg.off + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        else v = gathered[sg.off + i];
        vec[sg.idx[k] + (int64_t)c * ld] = v;
    }
}

// out[t] = sum over the ranks, IN RANK ORDER, of their n values (the residual-norm sums: the same bits on every rank).  The own contribution is
// read from `mine` (an emulated all-gather does not deliver a rank's chunk to itself).
template <bool SYS>
__global__ __launch_bounds__(64) void coll_sum_ranks(const double* __restrict__ gathered, long long chunk, int world, int rank, const double* __restrict__ mine,
                                                     int n, double* __restrict__ out) {
    const int t = threadIdx.x;
    if (t >= n) return;
    double sum = 0.0;
    for (int r = 0; r < world; ++r) {
        if (r == rank) { sum += mine[t]; continue; }
        if (SYS) sum += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(gathered) + (long long)r * chunk + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        else sum += gathered[(long long)r * chunk + t];
    }
    out[t] = sum;
}

// n doubles -> host-visible pinned memory, then the sequence word (one block; the coarsest right-hand side)
__global__ __launch_bounds__(kBlock) void publish_to_host(const double* __restrict__ src, double* __restrict__ dst, int n,
                                                          unsigned long long* flag, unsigned long long seq, int fenced) {
    for (int i = threadIdx.x; i < n; i += kBlock)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + i), (unsigned long long)__double_as_longlong(src[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    publish_order(fenced);      // write-through stores, drained: ahead of the sequence word (fence-free form: without a system fence)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// n doubles from host-visible pinned memory (the host's coarsest solution; the stream was held until the host had written it)
__global__ __launch_bounds__(kBlock) void fetch_from_host(const double* src, double* __restrict__ dst, int n) {
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src);
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        dst[i] = __longlong_as_double((long long)__hip_atomic_load(s + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// Residual norms (gravomg/src/multigrid_solver.cpp:1228-1277): per block, partial sums of
// w_i r_i^2 and w_i b_i^2 for r = A x - b and w = weight ? weight[i] : 1.
// partials layout: [block][2*D].  Reduced by reduce_partials (deterministic order).
template <int D>
__global__ __launch_bounds__(kBlock) void residual_norm_partials(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                                 const double* __restrict__ val, const double* __restrict__ diag,
                                                                 const double* __restrict__ b, const double* __restrict__ x,
                                                                 const double* __restrict__ weight, int ld, int slice_begin,
                                                                 int slice_end, double* __restrict__ partials) {
    __shared__ double red[kWavesPerBlock][2 * D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double sums[2 * D];
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) sums[c] = 0.0;
    // fixed-size grid, each wave strides over the slices: few partials, fixed summation order
    for (int s = slice_begin + __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + wave); s < slice_end;
         s += gridDim.x * kWavesPerBlock) {
        const int row = s * 64 + lane;
        double acc[D];
        row_dot<double, D>(slice_ptr, col, val, x, ld, s, lane, acc);
        const double dg = diag[row];
        const double w = weight ? weight[row] : 1.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double bi = b[row + (int64_t)c * ld];
            const double r = acc[c] + dg * x[row + (int64_t)c * ld] - bi;
            sums[2 * c] += (r * w) * r;
            sums[2 * c + 1] += (bi * w) * bi;
        }
    }
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) {
        double v = sums[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        double v = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kWavesPerBlock; ++w2) v += red[w2][threadIdx.x];
        partials[(int64_t)blockIdx.x * (2 * D) + threadIdx.x] = v;
    }
}

// The same sums with the launch geometry of the residual SpMV -- one slice per wave, XCD-aware slice map, one partial per
// block of kNormWaves slices (n_slices / 16 of them; reduce_partials adds them in index order, so the result does not depend on
// which block ran where).  The grid-stride kernel above walks ~6 slices per wave one after the other, each with its own
// load -> gather round trips: 59 us against 47 us for the SpMV over the same matrix at 3 M vertices.  MODE 1 also writes the
// residual b - A x as the fp32 right-hand side of the mixed-precision inner cycle (the defect-correction loop of BASELINE config 5).
constexpr int kNormWaves = 16;      // slices (waves) per block of the norm kernel: 1024 threads, one partial per 16 slices
template <int D, int MODE, int C16 = 0>
// (second launch bound: 8 waves per SIMD, i.e. TWO of these 1024-thread blocks per CU -- the d = 3 instantiation with the 16-bit codes
// took 66 registers without it, one block per CU, 94 -> 116 us)
__global__ __launch_bounds__(kNormWaves * 64, 8) void residual_norm_slices(const int64_t* __restrict__ slice_ptr, const int* __restrict__ col,
                                                               const double* __restrict__ val, const double* __restrict__ diag,
                                                               const double* __restrict__ b, const double* __restrict__ x,
                                                               const double* __restrict__ weight, int ld, int n_slices,
                                                               float* __restrict__ r32, double* __restrict__ partials,
                                                               const unsigned* __restrict__ col16 = nullptr, const int* __restrict__ win_base = nullptr, int c16_arg = 0) {
    __shared__ double red[kNormWaves][2 * D];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // partial index = the position of this block's slices in the slice range, not blockIdx (the XCD map permutes blocks)
    int nblk = gridDim.x, bb = blockIdx.x;
    if ((nblk & 7) == 0) bb = (bb & 7) * (nblk >> 3) + (bb >> 3);
    const int s = __builtin_amdgcn_readfirstlane(bb * kNormWaves + wave);
    double sums[2 * D];
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) sums[c] = 0.0;
    if (s < n_slices) {
        const int row = s * 64 + lane;
        double acc[D];
        row_dot_sel<double, D, C16>(slice_ptr, col, col16, win_base, c16_arg, val, x, ld, s, lane, acc);
        const double dg = diag[row];
        const double w = weight ? weight[row] : 1.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            const double bi = b[row + (int64_t)c * ld];
            double r;
            if (MODE == 1) { r = bi - (acc[c] + dg * x[row + (int64_t)c * ld]); r32[row + (int64_t)c * ld] = (float)r; }
            else r = acc[c] + dg * x[row + (int64_t)c * ld] - bi;
            sums[2 * c] = (r * w) * r;
            sums[2 * c + 1] = (bi * w) * bi;
        }
    }
#pragma unroll
    for (int c = 0; c < 2 * D; ++c) {
        double v = sums[c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        double v = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kNormWaves; ++w2) v += red[w2][threadIdx.x];
        partials[(int64_t)bb * (2 * D) + threadIdx.x] = v;
    }
}

// ---- mixed precision (fp32 inner V-cycle inside an fp64 defect-correction loop, BASELINE config 5) ----------------
// x (fp64) += e (fp32): the correction of one inner V-cycle
__global__ void add_correction(const float* __restrict__ e, double* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += (double)e[i];
}

__global__ void cvt_f64_to_f32(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

__global__ void cvt_f32_to_f64(const float* __restrict__ src, double* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (double)src[i];
}

// Natural (host) numbering <-> device numbering.  src natural: column-major n x D.
__global__ void permute_in(const double* __restrict__ src, int n, const int* __restrict__ new2old, double* __restrict__ dst,
                           int ld, int n_pad, int D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    const int old = new2old[r];
    for (int c = 0; c < D; ++c) dst[r + (int64_t)c * ld] = old >= 0 ? src[old + (int64_t)c * n] : 0.0;
}

__global__ void permute_out(const double* __restrict__ src, int ld, int n_pad, const int* __restrict__ new2old,
                            double* __restrict__ dst, int n, int D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    const int old = new2old[r];
    if (old < 0) return;
    for (int c = 0; c < D; ++c) dst[old + (int64_t)c * n] = src[r + (int64_t)c * ld];
}

// Lumped mass and its inverse in device numbering (weights of the M / M^-1 residual norms); padding rows get 1.
__global__ void permute_mass(const double* __restrict__ mass, const int* __restrict__ new2old, int n_pad, double* __restrict__ m,
                             double* __restrict__ minv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_pad) return;
    const int old = new2old[r];
    const double v = old >= 0 ? mass[old] : 1.0;
    m[r] = v;
    minv[r] = 1.0 / v;
}

// Coarsest level on the device (gmg_config::coarse_mode): e = A_L^-1 rc with the dense symmetric inverse, n x n (ld = n), level numbering (built by
// setup_kernels.hip.hpp::coarse_inverse_tiles).  One wave per row, lanes stride the row (coalesced; the matrix is the traffic: 8 n^2 bytes per
// application, streamed -- the vectors are a few KB and stay in the caches), wave reduction in a fixed order.  x/y leading dimension ldv.
// R rows per wave: the R rows share every load of x (with d = 3 and one row per wave the vectors were read three times as often as the matrix -- from the
// L2, whose bandwidth then bounded the product: 288 MB in 95 us at n = 6 005); a row's sum keeps its order (lane-strided, then the shuffle tree).
template <int D, int R, int U>
__global__ __launch_bounds__(kBlock) void dense_symv(const double* __restrict__ Ainv, int n, int lda, const double* __restrict__ x,
                                                     double* __restrict__ y, int ldv) {
    const int row0 = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * R;
    const int lane = threadIdx.x & 63;
    if (row0 >= n) return;
    double acc[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) acc[r][c] = 0.0;
    const double* a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = Ainv + (int64_t)(row0 + r < n ? row0 + r : row0) * lda;    // (rows beyond the last: recomputed, not stored)
    // U strides of 64 columns per trip: R * U matrix loads in flight per lane (a lane adds its products in ascending column order whatever U is: the
    // same bits) -- few rows mean few waves, which then have to keep more bytes in flight each to cover the memory latency
    int j = lane;
    for (; j + 64 * (U - 1) < n; j += 64 * U) {
        double v[R][U], xv[U][D];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r][u] = __builtin_nontemporal_load(a[r] + j + 64 * u);
#pragma unroll
            for (int c = 0; c < D; ++c) xv[u][c] = x[j + 64 * u + (int64_t)c * ldv];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) acc[r][c] += v[r][u] * xv[u][c];
    }
    for (; j < n; j += 64) {
        double xv[D];
#pragma unroll
        for (int c = 0; c < D; ++c) xv[c] = x[j + (int64_t)c * ldv];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double v = a[r][j];
#pragma unroll
            for (int c = 0; c < D; ++c) acc[r][c] += v * xv[c];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v = acc[r][c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && row0 + r < n) y[row0 + r + (int64_t)c * ldv] = v;
        }
}

// The same product with 16-byte loads: lane l takes columns 2 l and 2 l + 1 (+ 128 per stride), so a row of the matrix goes by in half as many load
// instructions -- with one or two rows per wave that is what the product waits for (n_L = 2 968: 21.4 -> 16.3 us, n_L = 4 046 at d = 3: 40.4 -> 29.2 us;
// with four rows per wave the loads of the vectors are shared widely enough and nothing changes).  Rows start every lda doubles, lda even (the inverse is
// stored with lda = n rounded up to 8), vectors 16-byte aligned (level buffers are).  A lane adds its columns in ascending order: deterministic, not the bits
// of the 8-byte kernel.
template <int D, int R, int U>
__global__ __launch_bounds__(kBlock) void dense_symv_v2(const double* __restrict__ Ainv, int n, int lda, const double* __restrict__ x, double* __restrict__ y, int ldv) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int row0 = (blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * R;
    const int lane = threadIdx.x & 63;
    if (row0 >= n) return;
    double acc[R][D];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) acc[r][c] = 0.0;
    const double* a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = Ainv + (int64_t)(row0 + r < n ? row0 + r : row0) * lda;
    int j = 2 * lane;
    for (; j + 128 * (U - 1) + 1 < n; j += 128 * U) {
        d2 v[R][U], xv[U][D];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[r][u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a[r] + j + 128 * u));
#pragma unroll
            for (int c = 0; c < D; ++c) xv[u][c] = *reinterpret_cast<const d2*>(x + j + 128 * u + (int64_t)c * ldv);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < D; ++c) { acc[r][c] += v[r][u].x * xv[u][c].x; acc[r][c] += v[r][u].y * xv[u][c].y; }
    }
    for (; j < n; j += 128) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < D; ++c) { acc[r][c] += a[r][j] * x[j + (int64_t)c * ldv]; if (j + 1 < n) acc[r][c] += a[r][j + 1] * x[j + 1 + (int64_t)c * ldv]; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            double v = acc[r][c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && row0 + r < n) y[row0 + r + (int64_t)c * ldv] = v;
        }
}

}  // namespace gmgk
