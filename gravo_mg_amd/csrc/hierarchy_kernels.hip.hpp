// hierarchy_kernels.hip.hpp -- device stage of the hierarchy builder (part of libgravomg_hip.so's single translation unit).
// The per-point parent selection of the Graph-Voronoi prolongation (gravomg/src/multigrid_solver.cpp:291-452; host routine:
// HierarchyBuilder::select_point, host_hierarchy.hpp) is independent per fine point: one thread per point.  The arithmetic is
// the host routine's, operation for operation -- no fused multiply-adds (the host code is compiled for baseline x86-64, which
// has none), IEEE division and square root -- so that the prolongation operators come out bit-identical
// (tests/test_gpu_hierarchy.py).  Everything sequential (sampling, clustering) and the cell / triangle bookkeeping stays on the host.
#pragma once

#include <hip/hip_runtime.h>

namespace gmgh {

#pragma clang fp contract(off)

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ D3 scale(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double norm(D3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ D3 normalized(D3 a) { const double z = dot(a, a); return z > 0 ? scale(1.0 / sqrt(z), a) : a; }
__device__ __forceinline__ D3 load3(const double* p, int i) { return {p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2]}; }

constexpr int kMaxEdgeKeys = 32;      // distinct neighbour cells one point's triangle tests can touch (two per triangle)

struct EdgeMap {                      // the host's SmallIntFloatMap: insertion order here, "lowest key" taken at the end
    int key[kMaxEdgeKeys];
    float val[kMaxEdgeKeys];
    int n = 0;
    bool overflow = false;
    __device__ int find(int k) const { for (int i = 0; i < n; ++i) if (key[i] == k) return i; return -1; }
    __device__ void set_if_absent(int k, float v) {
        if (find(k) >= 0) return;
        if (n == kMaxEdgeKeys) { overflow = true; return; }
        key[n] = k; val[n] = v; ++n;
    }
    __device__ void set(int k, float v) {
        const int i = find(k);
        if (i >= 0) { val[i] = v; return; }
        if (n == kMaxEdgeKeys) { overflow = true; return; }
        key[n] = k; val[n] = v; ++n;
    }
};

__device__ __forceinline__ void inv_dist_weights(const double* Pc, D3 p, const int* ids, int cnt, double* w) {
    double s = 0.0;
    for (int j = 0; j < cnt; ++j) { w[j] = 1.0 / fmax(1e-8, norm(sub(p, load3(Pc, ids[j])))); s += w[j]; }
    for (int j = 0; j < cnt; ++j) w[j] /= s;
}

__device__ __forceinline__ void edge_weights(int c, int other, D3 p, D3 pc, const double* Pc, int weighting, double& w1, double& w2) {
    if (weighting == 0) {
        const D3 e = sub(load3(Pc, other), pc);
        const double len = fmax(norm(e), 1e-8);
        w2 = dot(sub(p, pc), normalized(e)) / len;
        w2 = fmin(fmax(w2, 0.), 1.);
        w1 = 1. - w2;
    } else if (weighting == 1) {
        w1 = w2 = 0.5;
    } else {
        int ids[2] = {c, other};
        double w[2];
        inv_dist_weights(Pc, p, ids, 2, w);
        w1 = w[0]; w2 = w[1];
    }
}

__global__ __launch_bounds__(128) void select_parents(int nf, int Kc, int weighting, int nested, const double* __restrict__ P,
                                                      const double* __restrict__ Pc, const int* __restrict__ nearest,
                                                      const int* __restrict__ sample, const int* __restrict__ cadj_ptr,
                                                      const int* __restrict__ cadj, const int* __restrict__ tris,
                                                      const int* __restrict__ tof_ptr,
                                                      const int* __restrict__ tof, const int* __restrict__ NBc,
                                                      unsigned char* __restrict__ out_cnt, unsigned char* __restrict__ out_kind,
                                                      int* __restrict__ out_col, double* __restrict__ out_w) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    int* col = out_col + 3 * (size_t)f;
    double* w = out_w + 3 * (size_t)f;
    const D3 p = load3(P, f);
    const int c = nearest[f];
    const D3 pc = load3(Pc, c);
    if (nested && sample[c] == f) { col[0] = c; w[0] = 1.0; out_cnt[f] = 1; out_kind[f] = 4; return; }
    const int a0 = cadj_ptr[c], deg = cadj_ptr[c + 1] - a0;
    if (deg == 0) { col[0] = c; w[0] = 1.0; out_cnt[f] = 1; out_kind[f] = 3; return; }
    if (deg == 1) {
        const int nb = cadj[a0];
        edge_weights(c, nb, p, pc, Pc, weighting, w[0], w[1]);
        col[0] = c; col[1] = nb; out_cnt[f] = 2; out_kind[f] = 3;
        return;
    }
    EdgeMap inside;
    bool found = false;
    int best[3] = {0, 0, 0};
    double bary[3] = {0, 0, 0};
    for (int tq = tof_ptr[c]; tq < tof_ptr[c + 1] && !found; ++tq) {
        const int t = tof[tq];
        int tri[3] = {tris[3 * (size_t)t], tris[3 * (size_t)t + 1], tris[3 * (size_t)t + 2]};
        while (tri[0] != c) { const int h = tri[0]; tri[0] = tri[1]; tri[1] = tri[2]; tri[2] = h; }
        // the triangle's unit normal from its STORED vertex order (the host's tri_normal[t], :266-269), before the rotation to c
        const D3 nrm = normalized(cross(sub(load3(Pc, tris[3 * (size_t)t + 1]), load3(Pc, tris[3 * (size_t)t])),
                                        sub(load3(Pc, tris[3 * (size_t)t + 2]), load3(Pc, tris[3 * (size_t)t]))));
        const D3 v1 = load3(Pc, tri[0]), v2 = load3(Pc, tri[1]), v3 = load3(Pc, tri[2]);
        const D3 v1p = sub(p, v1), e12 = sub(v2, v1), e13 = sub(v3, v1);
        const double plane_dist = dot(sub(p, v1), nrm);
        const D3 q = sub(p, scale(plane_dist, nrm));
        const double area2 = dot(cross(sub(v2, v1), sub(v3, v1)), nrm);
        double b[3];
        b[0] = dot(cross(sub(v3, v2), sub(q, v2)), nrm) / area2;
        b[1] = dot(cross(sub(v1, v3), sub(q, v3)), nrm) / area2;
        b[2] = 1.0 - b[0] - b[1];
        inside.set_if_absent(tri[1], (float)norm(sub(v1p, scale(dot(v1p, e12), e12))));
        inside.set_if_absent(tri[2], (float)norm(sub(v1p, scale(dot(v1p, e13), e13))));
        if (b[0] < 0. || b[1] < 0.) inside.set(tri[1], -1.f);
        if (b[0] < 0. || b[2] < 0.) inside.set(tri[2], -1.f);
        if (b[0] >= 0. && b[1] >= 0. && b[2] >= 0.) {                // (the host returns |plane_dist| >= 0 here)
            found = true;
            best[0] = tri[0]; best[1] = tri[1]; best[2] = tri[2];
            bary[0] = b[0]; bary[1] = b[1]; bary[2] = b[2];
        }
    }
    if (inside.overflow) { out_cnt[f] = 255; out_kind[f] = 0; return; }      // the host redoes this point
    if (found) {
        if (weighting == 0) { w[0] = bary[0]; w[1] = bary[1]; w[2] = bary[2]; }
        else if (weighting == 1) { w[0] = w[1] = w[2] = 1.0 / 3; }
        else inv_dist_weights(Pc, p, best, 3, w);
        col[0] = best[0]; col[1] = best[1]; col[2] = best[2];
        out_cnt[f] = 3; out_kind[f] = 0;
        return;
    }
    int edge_to = -1;                                                  // the lowest cell index whose edge the point projects into
    for (int i = 0; i < inside.n; ++i)
        if (inside.val[i] >= 0.f && (edge_to < 0 || inside.key[i] < edge_to)) edge_to = inside.key[i];
    if (edge_to >= 0) {
        edge_weights(c, edge_to, p, pc, Pc, weighting, w[0], w[1]);
        col[0] = c; col[1] = edge_to; out_cnt[f] = 2; out_kind[f] = 1;
        return;
    }
    // closest three: the cell itself + its two nearest table neighbours ((distance, index) ascending, like std::sort of pairs)
    int from[3] = {c, -1, -1};
    double d1 = 0, d2 = 0;
    int cnt = 1;
    for (int j = 0; j < Kc; ++j) {
        const int nb = NBc[(size_t)c * Kc + j];
        if (nb < 0 || nb == c) continue;
        const double d = norm(sub(p, load3(Pc, nb)));
        if (cnt < 2 || d < d1 || (d == d1 && nb < from[1])) {
            if (cnt >= 2) { from[2] = from[1]; d2 = d1; }
            from[1] = nb; d1 = d;
            cnt = cnt < 2 ? 2 : 3;
        } else if (cnt < 3 || d < d2 || (d == d2 && nb < from[2])) {
            from[2] = nb; d2 = d;
            cnt = 3;
        }
    }
    double ww[3];
    inv_dist_weights(Pc, p, from, cnt, ww);
    for (int j = 0; j < cnt; ++j) { col[j] = from[j]; w[j] = ww[j]; }
    out_cnt[f] = (unsigned char)cnt; out_kind[f] = 2;
}

}  // namespace gmgh
