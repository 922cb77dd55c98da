// host_hierarchy.hpp -- Graph-Voronoi prolongation hierarchy (host; the greedy sampling and the Dijkstra clustering are
// sequential by definition, the per-point / per-cell stages run on all cores with the sequential stages' exact output).
//
// Restates, with its own data structures, what the reference's constructor does on its default path:
//   MGBS::MultigridSolver::buildHierarchy / constructProlongation
//       gravomg/src/multigrid_solver.cpp:43-60, 62-469
//   computeAverageEdgeLength     :695-711
//   fastDiskSample               :975-1013
//   constructDijkstraWithCluster :1015-1056
//   inTriangle                   :471-507
//   inverseDistanceWeights       :515-526,  uniformWeights :509-513
// Supported options: Sampling::FASTDISK only; Weighting::{BARYCENTRIC, UNIFORM, INVDIST}; nested;
// check_voronoi.  The other samplers / SIG06 / SIG21 / ablation hierarchies are paper baselines and are
// out of scope; they are rejected with GMG_ERR_UNSUPPORTED upstream.
//
// Behavioural details that are kept on purpose (SURVEY.md A.2/A.3 and the lines cited inline):
//   * a level is accepted only if the sample set has >= lower_bound points, at most 10 levels (:103,:156);
//   * the coarse neighbour table keeps the point itself in column 0 and at most maxNeigh-1 neighbours (:196-205);
//   * the first containing triangle wins (:343-349, the `break`), the first non-negative "inside edge" in
//     ascending key order wins (:375-383), edge distances are stored as float (std::map<int,float>, :336).
#pragma once
#include <algorithm>
#include <future>
#include <memory>
#include <atomic>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <queue>
#include <string>
#include <vector>

#include "host_sparse.hpp"

namespace gmg {

struct HierarchyOptions {
    double ratio = 8.0;          // gravomg_bindings/src/gravomg/core.py:10
    int lower_bound = 1000;
    bool check_voronoi = true;
    bool nested = false;
    int weighting = 0;           // 0 BARYCENTRIC, 1 UNIFORM, 2 INVDIST (multigrid_solver.h:48-52)
    bool keep_triangles = false; // the reference's `debug`: keep every level's candidate triangles (allTriangles, multigrid_solver.cpp:281)
    bool full_clustering = false; // run the Dijkstra clustering sweep (:1015-1056) literally instead of the shortcut that provably equals it (voronoi_dijkstra)
    // Optional accelerator for the per-point parent selection (:291-452) of a level: every fine point is independent there.  Host
    // pointers in, per-point results out; returns false when it did not run (the host loop does the level then).  A point it
    // could not handle comes back with cnt = 255 and is redone by the host routine.  Must produce the host routine's bits.
    struct SelectJob {
        int nf = 0, nc = 0, Kc = 0, ntri = 0, weighting = 0, nested = 0, device = -1;
        const double* P = nullptr;            // nf x 3 (row-major)
        const double* Pc = nullptr;           // nc x 3
        const int* nearest = nullptr;         // nf
        const int* sample = nullptr;          // nc
        const int* cadj_ptr = nullptr;        // nc + 1
        const int* cadj = nullptr;            // sorted neighbour cells of every cell
        const int* tris = nullptr;            // ntri x 3
        const double* tri_normal = nullptr;   // ntri x 3
        const int* tof_ptr = nullptr;         // nc + 1
        const int* tof = nullptr;             // triangles of every cell, ascending
        const int* NBc = nullptr;             // nc x Kc
        unsigned char* cnt = nullptr;         // nf: entries of the row (1..3), 255 = not handled
        unsigned char* kind = nullptr;        // nf: 0 triangle, 1 edge, 2 closest three, 3 single / one neighbour, 4 nested sample
        int* col = nullptr;                   // 3 nf
        double* w = nullptr;                  // 3 nf
    };
    bool (*device_select)(const SelectJob&) = nullptr;
    int device = -1;                          // the HIP device the hook shall use (the builder itself knows no devices)
    int device_select_min_points = 200000;    // smaller levels stay on the host (transfer set-up costs more than the loop)
};

struct HierarchyResult {
    std::vector<Compressed> U;               // U[k]: n_k x n_{k+1}, CSC (outer = coarse columns)
    std::vector<int> dof;                    // n_0, n_1, ..., n_L
    std::vector<std::vector<int>> samples;   // per level: fine index of each coarse sample
    std::vector<std::vector<int>> nearest;   // per level: the coarse sample (cluster) every fine point belongs to (nearestSource, :115,:171)
    std::vector<std::vector<double>> points; // per level: positions of the coarse points, n_{k+1} x 3 row-major (levelV, :216-241)
    std::vector<std::vector<std::array<int, 3>>> triangles;   // per level: the candidate triangles of the coarse points (allTriangles, :247-281); only with keep_triangles
    std::map<std::string, double> timing;    // the reference's hierarchyTiming keys
    // per level counts of prolongation row kinds: triangle / edge / fallback / single
    std::vector<std::array<int, 4>> row_kinds;
};

namespace detail {
struct V3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) { double z = dot(a, a); return z > 0 ? (1.0 / std::sqrt(z)) * a : a; }
struct HeapItem { int v; double dist; bool operator>(const HeapItem& o) const { return dist > o.dist; } };
// std::map<int, float> for a handful of keys (the "inside edge" bookkeeping of one fine point: two keys per tested
// triangle): same interface and the same ascending-key iteration order, but a sorted array instead of a heap node per
// key -- the map's allocations were a third of the triangle selection.
struct SmallIntFloatMap {
    std::vector<std::pair<int, float>> kv;
    void clear() { kv.clear(); }
    size_t find_pos(int key) const { size_t i = 0; while (i < kv.size() && kv[i].first < key) ++i; return i; }
    int count(int key) const { const size_t i = find_pos(key); return i < kv.size() && kv[i].first == key; }
    float& operator[](int key) {
        const size_t i = find_pos(key);
        if (i == kv.size() || kv[i].first != key) kv.insert(kv.begin() + i, {key, 0.f});
        return kv[i].second;
    }
    std::vector<std::pair<int, float>>::const_iterator begin() const { return kv.begin(); }
    std::vector<std::pair<int, float>>::const_iterator end() const { return kv.end(); }
};

// Read-only window on a contiguous array: the caller's positions / neighbour table on level 0 (no 170 MB copy), the
// builder's own vectors on the coarser levels.
template <class T>
struct View {
    const T* p = nullptr;
    size_t n = 0;
    View() {}
    View(const T* p_, size_t n_) : p(p_), n(n_) {}
    View(const std::vector<T>& v) : p(v.data()), n(v.size()) {}
    const T& operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    const T* data() const { return p; }
};
static_assert(sizeof(V3) == 3 * sizeof(double), "V3 must overlay a row of the n x 3 position array");
}  // namespace detail

class HierarchyBuilder {
    using V3 = detail::V3;
public:
    // pos: n x 3 row-major; neigh: n x K row-major, rows padded with -1 (scanned left to right).
    static HierarchyResult build(const double* pos, int n, const int* neigh, int K, const HierarchyOptions& opt) {
        using clk = std::chrono::steady_clock;
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        HierarchyResult R;
        auto t_all = clk::now();
        for (const char* key : {"PDS", "sampling", "cluster", "next_neighborhood", "next_positions", "triangle_finding", "triangle_selection",
                                "prepare", "edge_length", "assemble", "selection_on_device"}) R.timing[key] = 0.0;      // the last four: not in the reference's list
        R.timing["n_vertices"] = n;
        // Storage of the coarser levels (level 0 reads the caller's arrays in place).  Shared: the last stage of a level -- parent
        // selection + assembly of U_k -- runs as a task beside the sequential stages of the next levels and reads them too.
        std::shared_ptr<std::vector<V3>> P_hold;
        std::shared_ptr<std::vector<int>> NB_hold;
        struct Work {                          // what the last stage of one level reads (and the level's by-products)
            std::vector<int> nearest, sample, tof_ptr, tof;
            std::vector<std::vector<int>> cadj;
            std::vector<std::array<int, 3>> tris;
            std::vector<V3> tri_normal;
        };
        struct LevelOut { Compressed U; std::array<int, 4> kinds{0, 0, 0, 0}; double ms_select = 0, ms_assemble = 0, on_device = 0; };
        std::vector<std::shared_ptr<Work>> works;
        std::vector<std::future<LevelOut>> pending;
        detail::View<V3> P(reinterpret_cast<const V3*>(pos), (size_t)n);
        detail::View<int> NB(neigh, (size_t)n * K);
        int nbK = K;
        R.dof.push_back(n);
        R.timing["prepare"] = ms(t_all, clk::now());
        int level = 0;
        while ((int)P.size() > opt.lower_bound && level < 10) {                      // :103
            const int nf = (int)P.size();
            auto te = clk::now();
            ValueVec EL;
            const double radius = std::cbrt(opt.ratio) * average_edge_length(P, NB, nbK, EL);   // :104
            R.timing["edge_length"] += ms(te, clk::now());
            std::vector<double> D(nf, std::numeric_limits<double>::max());
            auto W = std::make_shared<Work>();
            std::vector<int>& nearest = W->nearest;
            nearest.assign(nf, 0);
            auto t0 = clk::now();
            std::vector<int>& sample = W->sample;
            sample = fast_disk_sample(P, NB, nbK, radius, D, nearest, EL);                  // :128
            if ((int)sample.size() < opt.lower_bound) break;                                // :156-159
            const int nc = (int)sample.size();
            auto t1 = clk::now();
            R.timing["sampling"] += ms(t0, t1);
            voronoi_dijkstra(P, sample, NB, nbK, D, nearest, EL, opt.full_clustering);                           // :170
            auto t2 = clk::now();
            R.timing["cluster"] += ms(t1, t2);

            // coarse adjacency: clusters that touch through a fine edge (:178-187); sorted unique lists
            // (the lists are SETS -- sorted, unique -- so they can be collected cluster by cluster on all cores: the fine
            // points of every cluster first, by a counting sort of `nearest`)
            std::vector<std::vector<int>>& cadj = W->cadj;
            cadj.assign(nc, std::vector<int>());
            std::vector<int> mem_ptr((size_t)nc + 1, 0), members(nf);
            for (int f = 0; f < nf; ++f) mem_ptr[nearest[f] + 1]++;
            for (int c = 0; c < nc; ++c) mem_ptr[c + 1] += mem_ptr[c];
            {
                std::vector<int> fillp(mem_ptr.begin(), mem_ptr.end() - 1);
                for (int f = 0; f < nf; ++f) members[fillp[nearest[f]]++] = f;
            }
            const int T = std::max(1, std::min(hw_threads(), 64));
            parallel_ranges(nc, T, [&](int lo, int hi, int) {
                for (int c = lo; c < hi; ++c) {
                    std::vector<int>& a = cadj[c];
                    a.reserve(32);                      // one allocation per cluster instead of one per doubling
                    for (int m = mem_ptr[c]; m < mem_ptr[c + 1]; ++m) {
                        const int f = members[m];
                        for (int j = 0; j < nbK; ++j) {
                            int g = NB[(size_t)f * nbK + j];
                            if (g < 0) break;
                            if (nearest[g] != c) a.push_back(nearest[g]);
                        }
                    }
                    std::sort(a.begin(), a.end());
                    a.erase(std::unique(a.begin(), a.end()), a.end());
                }
            }, 256);
            int max_nb = 0;
            for (auto& a : cadj) max_nb = std::max(max_nb, (int)a.size());
            // homogeneous table for the next level (:196-205): self first, then at most max_nb-1 neighbours
            auto NBc_sp = std::make_shared<std::vector<int>>((size_t)nc * std::max(max_nb, 1), -1);
            std::vector<int>& NBc = *NBc_sp;
            const int Kc = std::max(max_nb, 1);
            if (max_nb > 0)
                for (int i = 0; i < nc; ++i) {
                    NBc[(size_t)i * Kc] = i;
                    int cnt = 1;
                    for (int node : cadj[i]) {
                        if (node == i) continue;
                        if (cnt >= max_nb) break;
                        NBc[(size_t)i * Kc + cnt++] = node;
                    }
                }
            auto t3 = clk::now();
            R.timing["next_neighborhood"] += ms(t2, t3);

            // coarse positions (:216-240)
            auto Pc_sp = std::make_shared<std::vector<V3>>(nc, V3{0, 0, 0});
            std::vector<V3>& Pc = *Pc_sp;
            if (opt.nested) {
                for (int c = 0; c < nc; ++c) Pc[c] = P[sample[c]];
            } else {
                // (cell by cell on all threads: `members` lists a cell's points in ascending index, the order in which the reference's
                // single loop over the fine points adds them -- the sums have the same bits)
                parallel_ranges(nc, T, [&](int lo, int hi, int) {
                    for (int c = lo; c < hi; ++c) {
                        V3 acc{0, 0, 0};
                        for (int m = mem_ptr[c]; m < mem_ptr[c + 1]; ++m) acc = acc + P[members[m]];
                        const int csize = mem_ptr[c + 1] - mem_ptr[c];
                        if (csize == 1) {
                            V3 s = P[sample[c]];
                            for (int nb : cadj[c]) s = s + P[sample[nb]];
                            Pc[c] = (1.0 / (cadj[c].size() + 1.0)) * s;
                        } else {
                            Pc[c] = (1.0 / csize) * acc;
                        }
                    }
                }, 1024);
            }
            auto t4 = clk::now();
            R.timing["next_positions"] += ms(t3, t4);

            // candidate triangles from mutually adjacent Voronoi cells (:247-281)
            // Triangle t is created by its lowest cell; ids follow the cells' order, and every cell lists its triangles in
            // ascending id -- the order the sequential construction produces and the "first containing triangle" rule
            // below depends on.  Built on all cores: per-cell lists, ids by a prefix sum, cell lists by a stable counting sort.
            std::vector<std::vector<std::array<int, 3>>> local(nc);
            parallel_ranges(nc, T, [&](int lo, int hi, int) {
                for (int c = lo; c < hi; ++c) {
                    const auto& a = cadj[c];
                    for (size_t i2 = 0; i2 < a.size(); ++i2) {
                        int v2 = a[i2];
                        if (v2 < c) continue;
                        for (size_t i3 = i2 + 1; i3 < a.size(); ++i3) {
                            int v3 = a[i3];
                            if (v3 < c) continue;
                            if (!opt.check_voronoi || std::binary_search(cadj[v2].begin(), cadj[v2].end(), v3)) local[c].push_back({c, v2, v3});
                        }
                    }
                }
            }, 256);
            std::vector<int> tri_first((size_t)nc + 1, 0);
            for (int c = 0; c < nc; ++c) tri_first[c + 1] = tri_first[c] + (int)local[c].size();
            const int ntri = tri_first[nc];
            std::vector<std::array<int, 3>>& tris = W->tris;
            tris.assign(ntri, std::array<int, 3>{0, 0, 0});
            std::vector<V3>& tri_normal = W->tri_normal;
            tri_normal.assign(ntri, V3{0, 0, 0});
            parallel_ranges(nc, T, [&](int lo, int hi, int) {
                for (int c = lo; c < hi; ++c)
                    for (size_t q = 0; q < local[c].size(); ++q) {
                        const auto& tr = local[c][q];
                        tris[tri_first[c] + q] = tr;
                        tri_normal[tri_first[c] + q] = detail::normalized(detail::cross(Pc[tr[1]] - Pc[tr[0]], Pc[tr[2]] - Pc[tr[0]]));
                    }
            }, 256);
            std::vector<int>& tof_ptr = W->tof_ptr;
            tof_ptr.assign((size_t)nc + 1, 0);
            for (int t = 0; t < ntri; ++t) for (int q = 0; q < 3; ++q) tof_ptr[tris[t][q] + 1]++;
            for (int c = 0; c < nc; ++c) tof_ptr[c + 1] += tof_ptr[c];
            std::vector<int>& tof = W->tof;
            tof.assign(tof_ptr[nc], 0);
            {
                std::vector<int> fillp(tof_ptr.begin(), tof_ptr.end() - 1);
                for (int t = 0; t < ntri; ++t) for (int q = 0; q < 3; ++q) tof[fillp[tris[t][q]]++] = t;
            }
            auto t5 = clk::now();
            R.timing["triangle_finding"] += ms(t4, t5);

            // ---- last stage of the level, as a task: it needs nothing the next levels change and nothing they need waits for it
            auto P_keep = P_hold;                                  // (keeps this level's positions alive; null on level 0)
            const detail::View<V3> Pv = P;
            const HierarchyOptions optc = opt;
            auto finish = [W, Pc_sp, NBc_sp, P_keep, Pv, optc, nf, nc, Kc, ntri, T]() -> LevelOut {
            auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            LevelOut out;
            auto t5s = clk::now();
            const HierarchyOptions& opt = optc;
            const detail::View<V3>& P = Pv;
            const std::vector<V3>& Pc = *Pc_sp;
            const std::vector<int>& NBc = *NBc_sp;
            const std::vector<int>&nearest = W->nearest, &sample = W->sample, &tof_ptr = W->tof_ptr, &tof = W->tof;
            const std::vector<std::vector<int>>& cadj = W->cadj;
            const std::vector<std::array<int, 3>>& tris = W->tris;
            const std::vector<V3>& tri_normal = W->tri_normal;
            // per fine point: pick coarse parents and weights (:291-452)
            // Every fine point is independent: chunks of points on all cores, each with its own triplet list, concatenated in
            // chunk order afterwards (= the sequential emission order).
            const int nchunk = nf >= 8192 ? T : 1;
            const int chunk_len = (nf + nchunk - 1) / nchunk;
            struct Chunk { IndexVec row, col; ValueVec val; std::array<int, 4> kinds{0, 0, 0, 0}; };
            std::vector<Chunk> chunks(nchunk);
            // Big levels: the selection runs on the GPU when the caller supplied one (same bits); what comes back is one record
            // per point, turned into the same per-chunk triplet lists below.
            std::unique_ptr<unsigned char[]> dev_cnt, dev_kind;
            std::unique_ptr<int[]> dev_col;
            std::unique_ptr<double[]> dev_w;
            bool on_device = false;
            if (opt.device_select && nf >= opt.device_select_min_points) {
                std::vector<int> cadj_ptr((size_t)nc + 1, 0), cadj_flat;
                for (int c = 0; c < nc; ++c) cadj_ptr[c + 1] = cadj_ptr[c] + (int)cadj[c].size();
                cadj_flat.resize(cadj_ptr[nc]);
                parallel_ranges(nc, T, [&](int lo, int hi, int) { for (int c = lo; c < hi; ++c) std::copy(cadj[c].begin(), cadj[c].end(), cadj_flat.begin() + cadj_ptr[c]); }, 4096);
                dev_cnt.reset(new unsigned char[nf]); dev_kind.reset(new unsigned char[nf]);
                dev_col.reset(new int[3 * (size_t)nf]); dev_w.reset(new double[3 * (size_t)nf]);
                HierarchyOptions::SelectJob job;
                job.nf = nf; job.nc = nc; job.Kc = Kc; job.ntri = ntri; job.weighting = opt.weighting; job.nested = opt.nested ? 1 : 0; job.device = opt.device;
                job.P = reinterpret_cast<const double*>(P.data()); job.Pc = reinterpret_cast<const double*>(Pc.data());
                job.nearest = nearest.data(); job.sample = sample.data(); job.cadj_ptr = cadj_ptr.data(); job.cadj = cadj_flat.data();
                job.tris = reinterpret_cast<const int*>(tris.data()); job.tri_normal = reinterpret_cast<const double*>(tri_normal.data());
                job.tof_ptr = tof_ptr.data(); job.tof = tof.data(); job.NBc = NBc.data();
                job.cnt = dev_cnt.get(); job.kind = dev_kind.get(); job.col = dev_col.get(); job.w = dev_w.get();
                on_device = opt.device_select(job);
            }
            out.on_device = on_device ? 1.0 : 0.0;
            parallel_ranges(nchunk, nchunk, [&](int q0, int q1, int) {
            for (int q = q0; q < q1; ++q) {
            Chunk& ck = chunks[q];
            IndexVec& trow = ck.row; IndexVec& tcol = ck.col; ValueVec& tval = ck.val;
            std::array<int, 4>& kinds = ck.kinds;
            const int f_lo = q * chunk_len, f_hi = std::min(nf, f_lo + chunk_len);
            trow.reserve((size_t)(f_hi - f_lo) * 3); tcol.reserve((size_t)(f_hi - f_lo) * 3); tval.reserve((size_t)(f_hi - f_lo) * 3);
            auto emit = [&](int f, int c, double w) { trow.push_back(f); tcol.push_back(c); tval.push_back(w); };
            detail::SmallIntFloatMap inside_edge;
            for (int f = f_lo; f < f_hi; ++f) {
                if (on_device && dev_cnt[f] != 255) {
                    for (int j = 0; j < dev_cnt[f]; ++j) emit(f, dev_col[3 * (size_t)f + j], dev_w[3 * (size_t)f + j]);
                    if (dev_kind[f] < 4) ++kinds[dev_kind[f]];
                    continue;
                }
                select_point(f, P, Pc, nearest, sample, cadj, tof_ptr, tof, tris, tri_normal, NBc, Kc, opt, inside_edge, emit, kinds);
            }
            }
            }, 1);
            IndexVec trow, tcol; ValueVec tval;           // (not zero-filled, huge pages: 144 MB at 3 M vertices)
            std::array<int, 4> kinds{0, 0, 0, 0};
            {
                std::vector<size_t> at((size_t)nchunk + 1, 0);
                for (int q = 0; q < nchunk; ++q) at[q + 1] = at[q] + chunks[q].row.size();
                const size_t total = at[nchunk];
                trow.resize(total); tcol.resize(total); tval.resize(total);
                parallel_ranges(nchunk, nchunk, [&](int q0, int q1, int) {
                    for (int q = q0; q < q1; ++q) {
                        const Chunk& ck = chunks[q];
                        std::copy(ck.row.begin(), ck.row.end(), trow.begin() + at[q]);
                        std::copy(ck.col.begin(), ck.col.end(), tcol.begin() + at[q]);
                        std::copy(ck.val.begin(), ck.val.end(), tval.begin() + at[q]);
                    }
                }, 1);
                for (auto& ck : chunks) for (int z = 0; z < 4; ++z) kinds[z] += ck.kinds[z];
            }
            auto t6 = clk::now();
            out.ms_select = ms(t5s, t6);
            out.kinds = kinds;
            out.U = from_triplets(nf, nc, trow, tcol, tval);
            out.ms_assemble = ms(t6, clk::now());
            return out;
            };
            // big levels: beside the next levels' sequential sweeps; small ones: right here
            works.push_back(W);
            if (nf >= 100000) pending.push_back(std::async(std::launch::async, finish));
            else { std::promise<LevelOut> done; done.set_value(finish()); pending.push_back(done.get_future()); }
            { std::vector<double> xyz((size_t)nc * 3); for (int c = 0; c < nc; ++c) { xyz[3 * (size_t)c] = Pc[c].x; xyz[3 * (size_t)c + 1] = Pc[c].y; xyz[3 * (size_t)c + 2] = Pc[c].z; } R.points.push_back(std::move(xyz)); }
            R.dof.push_back(nc);
            P_hold = Pc_sp;
            NB_hold = NBc_sp;
            P = detail::View<V3>(*P_hold);
            NB = detail::View<int>(*NB_hold);
            nbK = Kc;
            ++level;
        }
        for (size_t k = 0; k < pending.size(); ++k) {               // collect the levels' last stages, in order
            LevelOut out = pending[k].get();
            R.U.push_back(std::move(out.U));
            R.row_kinds.push_back(out.kinds);
            R.timing["triangle_selection"] += out.ms_select;
            R.timing["assemble"] += out.ms_assemble;
            R.timing["selection_on_device"] += out.on_device;
            R.samples.push_back(std::move(works[k]->sample));
            R.nearest.push_back(std::move(works[k]->nearest));
            if (opt.keep_triangles) R.triangles.push_back(std::move(works[k]->tris));
        }
        R.timing["levels"] = (double)R.U.size();
        R.timing["hierarchy"] = ms(t_all, clk::now());
        return R;
    }

private:
    // EL keeps every edge length (slot i*K + j = |P[i] - P[NB[i*K + j]]|): the sampler and the clustering need the same
    // lengths again (56 per sample, one per relaxation) and read them instead of recomputing gathers + square roots.
    static double average_edge_length(detail::View<V3> P, detail::View<int> NB, int K, ValueVec& EL) {   // :695-711
        const int n = (int)P.size();
        EL.resize((size_t)n * K);
        // the lengths (gathers + square roots) on all threads ...
        parallel_ranges(n, hw_threads(), [&](int lo, int hi, int) {
            for (int i = lo; i < hi; ++i)
                for (int j = 0; j < K; ++j) {
                    const int g = NB[(size_t)i * K + j];
                    EL[(size_t)i * K + j] = g < 0 ? 0.0 : detail::norm(P[i] - P[g]);
                }
        }, 1 << 14);
        // ... their sum in the reference's order (i outer, j inner): absent and zero-length edges hold +0.0 and are skipped
        // like there, so the running sum has the bits of the sequential loop
        double sum = 0.0; long cnt = 0;
        const size_t m = (size_t)n * K;
        const double* el = EL.data();
        for (size_t q = 0; q < m; ++q)
            if (el[q] > 0) { sum += el[q]; ++cnt; }
        return sum / (double)cnt;
    }

    // :975-1013  greedy disk sampling over the one- and two-ring, first come first served in index order
    static std::vector<int> fast_disk_sample(detail::View<V3> P, detail::View<int> NB, int K, double radius,
                                             std::vector<double>& D, std::vector<int>& nearest, const ValueVec& EL) {
        const int n = (int)P.size();
        std::vector<char> visited(n, 0);
        std::vector<int> sel;
        for (int i = 0; i < n; ++i) {
            if (visited[i]) continue;
            const int sidx = (int)sel.size();
            sel.push_back(i);
            nearest[i] = sidx;
            for (int j = 0; j < K; ++j) {
                int g = NB[(size_t)i * K + j];
                if (g < 0) break;
                double d1 = EL[(size_t)i * K + j];                      // = norm(P[i] - P[g])
                if (!(d1 < radius)) continue;
                visited[g] = 1;
                if (d1 < D[g]) { D[g] = d1; nearest[g] = sidx; }
                for (int j2 = 0; j2 < K; ++j2) {
                    int g2 = NB[(size_t)g * K + j2];
                    if (g2 < 0) break;
                    double d2 = d1 + EL[(size_t)g * K + j2];            // = d1 + norm(P[g] - P[g2])
                    if (d2 < radius) {
                        visited[g2] = 1;
                        if (d2 < D[g2]) { D[g2] = d2; nearest[g2] = sidx; }
                    }
                }
            }
        }
        return sel;
    }

    // :1015-1056  multi-source Dijkstra; D/nearest arrive pre-seeded by the sampler and are only ever lowered.
    //
    // After fast_disk_sample that sweep cannot lower anything -- it only resets the samples themselves (D = 0, owner = own index; a
    // later sample's ring may have overwritten a sample's owner):
    //   * every point that is not a sample was "visited", which happens only together with D[g] = min(D[g], d) for a d < radius;
    //   * the heap starts with the samples at distance 0.  Popping sample s offers each table neighbour g the candidate 0 + |p_s - p_g|
    //     = d1, bit for bit the d1 the sampler formed for the same pair.  d1 < radius: the sampler already took the minimum with
    //     it, and D never rises.  d1 >= radius (or NaN): D[g] < radius <= d1, or g is a sample with D[g] = 0.  Either way
    //     `cand < D[g]` is false, nothing is pushed, and the heap runs empty after the samples.
    // So the result is known after the initialisation loop: 35-50 ms of heap traffic at 3 M points become 0.3 ms, with the
    // reference's output by construction (multi-threading or a GPU could only have reproduced that no-op faster).  The one
    // assumption is the sampler's early exit: it stops at a row's first -1 where the sweep skips over it, so a sample whose row
    // holds a neighbour BEHIND a -1 (the reference's tables never do: rows are padded at the end) makes the full sweep run.
    // gmg_hierarchy_options::full_clustering forces it (tests/test_hierarchy_restatement.py compares the two).
    static void voronoi_dijkstra(detail::View<V3> P, const std::vector<int>& src, detail::View<int> NB, int K,
                                 std::vector<double>& D, std::vector<int>& nearest, const ValueVec& EL, bool full) {
        std::priority_queue<detail::HeapItem, std::vector<detail::HeapItem>, std::greater<detail::HeapItem>> heap;
        const int ns = (int)src.size();
        std::atomic<int> gaps{0};
        // (one pass over the samples on all threads: each touches three scattered cache lines)
        parallel_ranges(ns, hw_threads(), [&](int lo, int hi, int) {
            bool any = false;
            for (int i = lo; i < hi; ++i) {
                const int* row = NB.data() + (size_t)src[i] * K;
                bool gap = false;
                for (int j = 0; j < K; ++j) { if (row[j] < 0) gap = true; else if (gap) { any = true; break; } }
                D[src[i]] = 0.0;
                nearest[src[i]] = i;
            }
            if (any) gaps.fetch_add(1);
        }, 1 << 14);
        full = full || gaps.load() > 0;
        if (full) for (int i = 0; i < ns; ++i) heap.push({src[i], 0.0});
        while (!heap.empty()) {
            detail::HeapItem it = heap.top();
            const int owner = nearest[it.v];
            heap.pop();
            for (int j = 0; j < K; ++j) {
                int g = NB[(size_t)it.v * K + j];
                if (g < 0) continue;
                double cand = it.dist + EL[(size_t)it.v * K + j];      // = norm(P[g] - P[it.v]): the same squares, the same sum
                if (cand < D[g]) { D[g] = cand; heap.push({g, cand}); nearest[g] = owner; }
            }
        }
    }

    // Parents and weights of ONE fine point (:291-452): the containing candidate triangle of its cell, else the edge it projects
    // into, else the cell and its two nearest table neighbours.  emit(f, coarse, weight) in the order the row is stored.
    template <class PosView, class Emit>
    static void select_point(int f, const PosView& P, const std::vector<V3>& Pc, const std::vector<int>& nearest, const std::vector<int>& sample,
                             const std::vector<std::vector<int>>& cadj, const std::vector<int>& tof_ptr, const std::vector<int>& tof,
                             const std::vector<std::array<int, 3>>& tris, const std::vector<V3>& tri_normal, const std::vector<int>& NBc, int Kc,
                             const HierarchyOptions& opt, detail::SmallIntFloatMap& inside_edge, Emit& emit, std::array<int, 4>& kinds) {
        const V3 p = P[f];
        const int c = nearest[f];
        const V3 pc = Pc[c];
        if (opt.nested && sample[c] == f) { emit(f, c, 1.0); return; }
        if (cadj[c].empty()) { emit(f, c, 1.0); ++kinds[3]; return; }
        if (cadj[c].size() == 1) {
            int nb = cadj[c][0];
            emit_edge(f, c, nb, p, pc, Pc, opt.weighting, emit);
            ++kinds[3];
            return;
        }
        inside_edge.clear();
        bool found = false;
        std::array<int, 3> best{0, 0, 0};
        double bary[3] = {0, 0, 0};
        for (int tq = tof_ptr[c]; tq < tof_ptr[c + 1]; ++tq) {
            const int t = tof[tq];
            std::array<int, 3> tri = tris[t];
            while (tri[0] != c) std::rotate(tri.begin(), tri.begin() + 1, tri.end());
            double b[3];
            double dist = in_triangle(p, tri, tri_normal[t], Pc, b, inside_edge);
            if (dist >= 0.0) { found = true; best = tri; bary[0] = b[0]; bary[1] = b[1]; bary[2] = b[2]; break; }
        }
        if (found) {
            ++kinds[0];
            double w[3];
            if (opt.weighting == 0) { w[0] = bary[0]; w[1] = bary[1]; w[2] = bary[2]; }
            else if (opt.weighting == 1) { w[0] = w[1] = w[2] = 1.0 / 3; }
            else inv_dist_weights(Pc, p, best.data(), 3, w);
            for (int j = 0; j < 3; ++j) emit(f, best[j], w[j]);
            return;
        }
        int edge_to = -1;
        for (const auto& kv : inside_edge)
            if (kv.second >= 0.f) { edge_to = kv.first; break; }              // :375-383
        if (edge_to >= 0) {
            ++kinds[1];
            emit_edge(f, c, edge_to, p, pc, Pc, opt.weighting, emit);
            return;
        }
        // closest three (:415-435): the cell itself + its two nearest table neighbours, inverse distance
        ++kinds[2];
        std::vector<std::pair<double, int>> cand;
        for (int j = 0; j < Kc; ++j) {
            int nb = NBc[(size_t)c * Kc + j];
            if (nb < 0 || nb == c) continue;
            cand.emplace_back(detail::norm(p - Pc[nb]), nb);
        }
        std::sort(cand.begin(), cand.end());
        int from[3] = {c, -1, -1};
        int cnt = 1;
        for (size_t j = 0; j < cand.size() && cnt < 3; ++j) from[cnt++] = cand[j].second;
        double w[3];
        inv_dist_weights(Pc, p, from, cnt, w);
        for (int j = 0; j < cnt; ++j) emit(f, from[j], w[j]);
    }

    // :471-507  barycentric test of the projection of p onto the triangle's plane; returns |distance to plane|
    // when inside, -1 otherwise, and records the "inside edge" bookkeeping.
    static double in_triangle(V3 p, const std::array<int, 3>& tri, V3 nrm, const std::vector<V3>& pos, double bary[3],
                              detail::SmallIntFloatMap& inside_edge) {
        const V3 v1 = pos[tri[0]], v2 = pos[tri[1]], v3 = pos[tri[2]];
        const V3 v1p = p - v1, e12 = v2 - v1, e13 = v3 - v1;
        const double plane_dist = detail::dot(p - v1, nrm);
        const V3 q = p - plane_dist * nrm;
        const double area2 = detail::dot(detail::cross(v2 - v1, v3 - v1), nrm);
        bary[0] = detail::dot(detail::cross(v3 - v2, q - v2), nrm) / area2;
        bary[1] = detail::dot(detail::cross(v1 - v3, q - v3), nrm) / area2;
        bary[2] = 1.0 - bary[0] - bary[1];
        if (!inside_edge.count(tri[1])) inside_edge[tri[1]] = (float)detail::norm(v1p - detail::dot(v1p, e12) * e12);
        if (!inside_edge.count(tri[2])) inside_edge[tri[2]] = (float)detail::norm(v1p - detail::dot(v1p, e13) * e13);
        if (bary[0] < 0. || bary[1] < 0.) inside_edge[tri[1]] = -1.f;
        if (bary[0] < 0. || bary[2] < 0.) inside_edge[tri[2]] = -1.f;
        if (bary[0] >= 0. && bary[1] >= 0. && bary[2] >= 0.) return std::fabs(plane_dist);
        return -1.0;
    }

    static void inv_dist_weights(const std::vector<V3>& pos, V3 p, const int* ids, int cnt, double* w) {   // :515-526
        double s = 0.0;
        for (int j = 0; j < cnt; ++j) { w[j] = 1.0 / std::max(1e-8, detail::norm(p - pos[ids[j]])); s += w[j]; }
        for (int j = 0; j < cnt; ++j) w[j] /= s;
    }

    // two-parent rows (:308-331 and :386-411): clamp the projection parameter onto the segment
    template <class Emit>
    static void emit_edge(int f, int c, int other, V3 p, V3 pc, const std::vector<V3>& Pc, int weighting, Emit& emit) {
        double w1, w2;
        if (weighting == 0) {
            V3 e = Pc[other] - pc;
            double len = std::max(detail::norm(e), 1e-8);
            w2 = detail::dot(p - pc, detail::normalized(e)) / len;
            w2 = std::min(std::max(w2, 0.), 1.);
            w1 = 1. - w2;
        } else if (weighting == 1) {
            w1 = w2 = 0.5;
        } else {
            int ids[2] = {c, other};
            double w[2];
            inv_dist_weights(Pc, p, ids, 2, w);
            w1 = w[0]; w2 = w[1];
        }
        emit(f, c, w1);
        emit(f, other, w2);
    }

    // Eigen's setFromTriplets semantics: duplicates are summed, explicit zeros are kept, inner indices sorted.  The
    // triplets arrive grouped by row (ascending) with at most a handful per row, so duplicates can only be neighbours in
    // the list: they are merged there, then the columns are filled on all cores (atomic slot counters) and each column
    // is sorted by row -- every (row, column) is unique by then, so the result does not depend on the fill order.
    // (the triplet values are consumed: duplicates inside a row are merged into the first occurrence in place)
    static Compressed from_triplets(int nrows, int ncols, const IndexVec& r, const IndexVec& c, ValueVec& vm) {
        const size_t nt = r.size();
        std::unique_ptr<char[]> keep(new char[std::max<size_t>(nt, 1)]);
        const int Tm = std::max(1, std::min(hw_threads(), 64));
        std::vector<size_t> kept_of((size_t)Tm, 0);
        // rows are short runs of consecutive triplets and independent of each other: every thread takes a range of
        // triplets moved forward to the next row boundary (quadratic merge inside a row, as in the sequential loop)
        parallel_ranges(Tm, Tm, [&](int q0, int q1, int) {
            for (int q = q0; q < q1; ++q) {
                size_t t = nt * (size_t)q / Tm, te = nt * (size_t)(q + 1) / Tm;
                while (t > 0 && t < nt && r[t] == r[t - 1]) ++t;
                while (te > 0 && te < nt && r[te] == r[te - 1]) ++te;
                size_t kept = 0;
                while (t < te) {
                    size_t e = t;
                    while (e < nt && r[e] == r[t]) ++e;
                    for (size_t i = t; i < e; ++i) keep[i] = 1;
                    for (size_t i = t; i < e; ++i)
                        if (keep[i]) {
                            ++kept;
                            for (size_t j = i + 1; j < e; ++j)
                                if (keep[j] && c[j] == c[i]) { vm[i] += vm[j]; keep[j] = 0; }
                        }
                    t = e;
                }
                kept_of[q] = kept;
            }
        }, 1);
        size_t total = 0;
        for (size_t k : kept_of) total += k;
        Compressed M;
        M.n_outer = ncols; M.n_inner = nrows;
        M.idx.resize(total); M.val.resize(total);      // not zero-filled (default_init_allocator): first touched by the threaded fill
        const int T = std::max(1, std::min(hw_threads(), 64));
        std::unique_ptr<std::atomic<int>[]> cnt(new std::atomic<int>[(size_t)ncols + 1]);
        parallel_ranges(ncols + 1, T, [&](int lo, int hi, int) { for (int j = lo; j < hi; ++j) cnt[j].store(0, std::memory_order_relaxed); });
        parallel_ranges((int)nt, T, [&](int lo, int hi, int) {
            for (int t = lo; t < hi; ++t) if (keep[t]) cnt[c[t]].fetch_add(1, std::memory_order_relaxed);
        });
        M.ptr.assign((size_t)ncols + 1, 0);
        for (int j = 0; j < ncols; ++j) M.ptr[j + 1] = M.ptr[j] + cnt[j].load(std::memory_order_relaxed);
        parallel_ranges(ncols, T, [&](int lo, int hi, int) { for (int j = lo; j < hi; ++j) cnt[j].store(M.ptr[j], std::memory_order_relaxed); });
        parallel_ranges((int)nt, T, [&](int lo, int hi, int) {
            for (int t = lo; t < hi; ++t)
                if (keep[t]) { const int q = cnt[c[t]].fetch_add(1, std::memory_order_relaxed); M.idx[q] = r[t]; M.val[q] = vm[t]; }
        });
        parallel_ranges(ncols, T, [&](int lo, int hi, int) {
            std::vector<std::pair<int, double>> tmp;
            for (int j = lo; j < hi; ++j) {
                const int b0 = M.ptr[j], e0 = M.ptr[j + 1];
                bool sorted = true;
                for (int q = b0 + 1; q < e0; ++q) if (M.idx[q] < M.idx[q - 1]) { sorted = false; break; }
                if (sorted) continue;
                tmp.clear();
                for (int q = b0; q < e0; ++q) tmp.emplace_back(M.idx[q], M.val[q]);
                std::sort(tmp.begin(), tmp.end(), [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
                for (int q = b0; q < e0; ++q) { M.idx[q] = tmp[q - b0].first; M.val[q] = tmp[q - b0].second; }
            }
        }, 256);
        return M;
    }
};

}  // namespace gmg
