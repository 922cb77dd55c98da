// multigrid_solver.cpp -- MGBS::MultigridSolver over the C-ABI of libgravomg_hip.so (see multigrid_solver.h).
#include "multigrid_solver.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <algorithm>
#include <vector>
#include <thread>

namespace MGBS {

namespace {
// [0, n) cut into contiguous ranges run on up to 16 threads (plain std::thread: these loops run once per hierarchy, over 10^6..10^7
// items each touching fresh memory -- page faults and strided reads that one core takes tens of milliseconds for)
template <class F>
void parallelRanges(int n, F&& f) {
    const int hw = (int)std::thread::hardware_concurrency();
    const int T = std::max(1, std::min({16, hw > 0 ? hw : 1, n / 65536 + 1}));
    if (T == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back([&, t] { f((int)((long long)n * t / T), (int)((long long)n * (t + 1) / T)); });
    f(0, (int)((long long)n / T));
    for (auto& x : th) x.join();
}

bool configEqual(const gmg_config& a, const gmg_config& b) { return std::memcmp(&a, &b, sizeof(gmg_config)) == 0; }

// FNV-1a style digest of a byte range taken 8 bytes at a time, in up to 16 chunks hashed concurrently and combined in
// order (the chunking is fixed by the length, so the digest is a function of the content only).
uint64_t digestBytes(const void* data, size_t bytes, uint64_t seed) {
    const size_t words = bytes / 8;
    const int chunks = (int)std::min<size_t>(16, words / (1u << 16) + 1);
    std::vector<uint64_t> part(chunks, 0);
    auto work = [&](int c) {
        const uint64_t* p = (const uint64_t*)data;
        const size_t lo = words * c / chunks, hi = words * (c + 1) / chunks;
        // four independent multiply chains (one chain runs at ~5 cycles per word: 250 MB would take 40 ms of one core)
        const uint64_t K = 1099511628211ull;
        uint64_t h = 1469598103934665603ull ^ seed ^ (0x9e3779b97f4a7c15ull * (uint64_t)(c + 1));
        uint64_t h1 = h ^ 0xa0761d6478bd642full, h2 = h ^ 0xe7037ed1a0b428dbull, h3 = h ^ 0x8ebc6af09c88c6e3ull;
        size_t i = lo;
        for (; i + 4 <= hi; i += 4) {
            uint64_t w[4];
            std::memcpy(w, p + i, 32);
            h = (h ^ w[0]) * K; h ^= h >> 31;
            h1 = (h1 ^ w[1]) * K; h1 ^= h1 >> 31;
            h2 = (h2 ^ w[2]) * K; h2 ^= h2 >> 31;
            h3 = (h3 ^ w[3]) * K; h3 ^= h3 >> 31;
        }
        for (; i < hi; ++i) { uint64_t w; std::memcpy(&w, p + i, 8); h = (h ^ w) * K; h ^= h >> 31; }
        h = (h ^ h1) * K; h = (h ^ h2) * K; h = (h ^ h3) * K;
        part[c] = h;
    };
    if (chunks == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int c = 1; c < chunks; ++c) th.emplace_back(work, c);
        work(0);
        for (auto& t : th) t.join();
    }
    uint64_t h = 1469598103934665603ull ^ seed ^ (uint64_t)bytes;
    for (int c = 0; c < chunks; ++c) h = (h ^ part[c]) * 1099511628211ull;
    const unsigned char* tail = (const unsigned char*)data + words * 8;
    for (size_t i = 0; i < bytes - words * 8; ++i) h = (h ^ tail[i]) * 1099511628211ull;
    return h;
}
}  // namespace

std::pair<uint64_t, uint64_t> SparseMatrix::digest() const {
    const size_t nnz = (size_t)nonZeros();
    uint64_t a = digestBytes(outerPtr(), sizeof(int) * ((size_t)cols_ + 1), (uint64_t)rows_ * 0x100000001b3ull + (uint64_t)cols_);
    a ^= digestBytes(innerPtr(), sizeof(int) * nnz, 0x51ed270b7f4a7c15ull) * 0x9e3779b97f4a7c15ull;
    const uint64_t b = digestBytes(valuePtr(), sizeof(double) * nnz, 0xc2b2ae3d27d4eb4full);
    return {a, b};
}

// Where this class departs from the engine's defaults.  The reference's callers hand solve() a NEW system matrix per call (a new tau per frame,
// demos/smoothing.py:43-52), and solve() sets everything up again every time (multigrid_solver.cpp:1387-1401): a caller like that solves each
// system ONCE, with 4-6 V-cycles, and the engine's default -- a dense inverse of the coarsest operator built per gmg_set_system, which pays off
// from ~50 cycles per system on -- would cost it 2-7 ms per solve for 0.2-0.7 ms saved.  The coarsest solve therefore stays where
// the reference has it, on the host; set_engine_option("coarse_mode", 2) selects the engine's default for callers that keep a system.
static void dropInDefaults(gmg_config& c) {
    // (this file and libgravomg_hip.so are built separately: a library with another edition of gmg_config would misread every field)
    if (gmg_config_size() != (int)sizeof(gmg_config))
        throw std::runtime_error("libgravomg_hip.so was built with another gmg_config than this module (include/gravomg_hip.h changed): rebuild both (gravo_mg_amd/csrc/build.sh, build_bindings.sh)");
    c.coarse_mode = GMG_COARSE_HOST_LDLT;
}

MultigridSolver::MultigridSolver(MatrixXd& V_, MatrixXi& neigh_, SparseMatrix& M_) : V(V_), neigh(neigh_), M(M_) {
    hierarchyTiming["n_vertices"] = V.rows();               // multigrid_solver.cpp:21
    gmg_config_default(&engineConfig);
    dropInDefaults(engineConfig);
    std::memset(&createdWith_, 0, sizeof(createdWith_));
}

MultigridSolver::MultigridSolver(MatrixXd&& V_, MatrixXi&& neigh_, SparseMatrix&& M_) : V(std::move(V_)), neigh(std::move(neigh_)), M(std::move(M_)) {
    hierarchyTiming["n_vertices"] = V.rows();
    gmg_config_default(&engineConfig);
    dropInDefaults(engineConfig);
    std::memset(&createdWith_, 0, sizeof(createdWith_));
}

MultigridSolver::~MultigridSolver() {
    if (engine_) gmg_destroy(engine_);
    if (parked_.engine) gmg_destroy(parked_.engine);
}

const char* MultigridSolver::lastError() const { return err_.c_str(); }

// multigrid_solver.cpp:43-60.  Only the default FASTDISK hierarchy is in scope; SIG06 / ablation hierarchies and the
// other samplers are paper baselines and are refused loudly.
void MultigridSolver::buildHierarchy() {
    if (sig06 || ablation) {
        err_ = "SIG06 / ablation hierarchies are out of scope of the MI355X hot-path build";
        std::cout << "ERROR! " << err_ << std::endl;
        U.clear();
        return;
    }
    gmg_hierarchy_options opt;
    gmg_hierarchy_options_default(&opt);
    opt.ratio = ratio; opt.lower_bound = lowBound; opt.check_voronoi = checkVoronoi; opt.nested = nested;
    opt.sampling = (int)samplingStrategy; opt.weighting = (int)weightingScheme; opt.debug = debug ? 1 : 0;
    // positions: column-major n x 3 -> row-major
    const int n = V.rows();
    std::unique_ptr<double[]> pos(new double[(size_t)n * 3]);             // (not value-initialised: every entry is written below, on all cores)
    parallelRanges(n, [&](int lo, int hi) {
        for (int i = lo; i < hi; ++i) for (int c = 0; c < 3; ++c) pos[(size_t)i * 3 + c] = V(i, c);
    });
    const bool trace = ctorTrace();
    auto tt = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[gravomg ctor]   %-26s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tt).count());
        tt = now;
    };
    lap("row-major positions");
    gmg_hierarchy hh = nullptr;
    int rc = gmg_hierarchy_build(pos.get(), n, neigh.data.data(), neigh.cols(), &opt, &hh);
    if (rc != GMG_OK) {
        err_ = rc == GMG_ERR_UNSUPPORTED ? "only Sampling::FASTDISK is supported by the MI355X hot-path build" : "gmg_hierarchy_build failed";
        std::cout << "ERROR! " << err_ << std::endl;
        U.clear();
        return;
    }
    lap("gmg_hierarchy_build");
    const int L = gmg_hierarchy_num_levels(hh);
    U.assign(L, SparseMatrix());
    DoF.clear();
    DoF.push_back(n);
    for (int k = 0; k < L; ++k) {
        int nf, nc, nnz;
        gmg_hierarchy_level_shape(hh, k, &nf, &nc, &nnz);
        SparseMatrix& u = U[k];
        u.rows_ = nf; u.cols_ = nc;
        u.outer.resize(nc + 1); u.inner.resize(nnz); u.values.resize(nnz);
        gmg_hierarchy_get_prolongation(hh, k, u.outer.data(), u.inner.data(), u.values.data());
        DoF.push_back(nc);
    }
    lap("fetch U");
    // what the reference keeps beside U (samples and nearestSource always, levelV with debug only: :128, :171, :241)
    samples.assign(L, std::vector<int>());
    nearestSource.assign(L, std::vector<size_t>());
    levelV.clear();
    // the remaining debug members of the reference (multigrid_solver.h:99-102): levelE is only filled by the SIG06 hierarchy
    // (:663-681) and levelN by nothing at all, so both stay empty here as there; with `debug` the reference keeps every level's
    // candidate triangles (allTriangles, :281) and a zero vector per level in noTriFoundMap (:291, never written afterwards)
    levelE.clear(); levelN.clear(); allTriangles.clear(); noTriFoundMap.clear();
    for (int k = 0; k < L; ++k) {
        const int nf = U[k].rows_, nc = U[k].cols_;
        if (debug) {
            int cnt = 0;
            gmg_hierarchy_get_triangles(hh, k, nullptr, &cnt);
            std::vector<int> flat((size_t)cnt * 3);
            if (cnt > 0) gmg_hierarchy_get_triangles(hh, k, flat.data(), &cnt);
            std::vector<std::vector<int>> tris((size_t)cnt);
            for (int t = 0; t < cnt; ++t) tris[t] = {flat[3 * (size_t)t], flat[3 * (size_t)t + 1], flat[3 * (size_t)t + 2]};
            allTriangles.push_back(std::move(tris));
            noTriFoundMap.push_back(std::vector<int>((size_t)nf, 0));
        }
        samples[k].resize(nc);
        gmg_hierarchy_get_samples(hh, k, samples[k].data());
        std::unique_ptr<int[]> near(new int[(size_t)nf]);
        gmg_hierarchy_get_nearest(hh, k, near.get());
        nearestSource[k].resize(nf);
        parallelRanges(nf, [&](int lo, int hi) { for (int i = lo; i < hi; ++i) nearestSource[k][i] = (size_t)near[i]; });
        if (debug) {
            std::vector<double> xyz((size_t)nc * 3);
            gmg_hierarchy_get_points(hh, k, xyz.data());
            MatrixXd P(nc, 3);
            for (int c = 0; c < nc; ++c) for (int a = 0; a < 3; ++a) P.data[(size_t)a * nc + c] = xyz[3 * (size_t)c + a];      // column-major n x 3
            levelV.push_back(P);
        }
    }
    for (const char* key : {"hierarchy", "sampling", "cluster", "next_neighborhood", "next_positions", "triangle_finding", "triangle_selection", "PDS", "levels"}) {
        double v = 0;
        if (gmg_hierarchy_get_timing(hh, key, &v) == GMG_OK) hierarchyTiming[key] = v;
    }
    lap("samples / nearest / timing");
    {   // by-product of the construction for the engine (not in the reference): breadth-first order of the points, made for inputs
        // whose numbering has no locality -- belongs to the `U` built here (prepareEngine drops it when `U` was replaced)
        int cnt = 0;
        fineOrder_.clear();
        if (gmg_hierarchy_get_fine_order(hh, nullptr, &cnt) == GMG_OK && cnt > 0) { fineOrder_.resize((size_t)cnt); gmg_hierarchy_get_fine_order(hh, fineOrder_.data(), &cnt); }
        fineOrderFor_.clear();
        for (size_t k = 0; k < U.size(); ++k) fineOrderFor_.push_back(U[k].digest());
    }
    gmg_hierarchy_destroy(hh);
    lap("destroy");
}

void MultigridSolver::swapParked() {
    std::swap(engine_, parked_.engine);
    std::swap(createdWith_, parked_.createdWith);
    std::swap(uploadedU_, parked_.uploadedU);
    std::swap(uploadedLHS_, parked_.uploadedLHS);
    std::swap(systemReady_, parked_.systemReady);
}

// the exact-GS fallback is for the engine's DEFAULT smoothers only: a smoother the caller chose (weighted Jacobi) is never replaced, and
// Gauss-Seidel in colour order on every level has nothing to fall back to
bool MultigridSolver::fallbackAllowed() const {
    return engineConfig.smoother == GMG_SMOOTHER_MULTICOLOR_GS && (engineConfig.block_rows != 0 || engineConfig.gs_omega != 1.0);
}

// what a remembered "needs the fallback" verdict was reached under: the engine options and the sweep counts
uint64_t MultigridSolver::configKey() const {
    uint64_t k = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { k ^= b[i]; k *= 1099511628211ull; } };
    gmg_config c = engineConfig;
    c.verbose = 0;
    mix(&c, sizeof(c)); mix(&preIters, sizeof(preIters)); mix(&postIters, sizeof(postIters));
    return k;
}

int MultigridSolver::ensureEngine() {
    gmg_config want = engineConfig;
    want.pre_iters = preIters; want.post_iters = postIters; want.verbose = 0;
    if (exactGsActive_) { want.smoother = GMG_SMOOTHER_MULTICOLOR_GS; want.block_rows = 0; want.gs_omega = 1.0; }      // scoped fallback, see solve()
    if (engine_ && !configEqual(want, createdWith_)) {
        // An engine is PARKED only across the switch configured <-> exact-GS fallback (an application alternating between systems switches
        // engines, it does not rebuild them).  Any other change of configuration destroys the old engines: a parked one keeps its whole
        // device state -- hierarchy, level operators, vectors: several GB at 3 M vertices.
        auto fallbackPair = [](gmg_config a, gmg_config b) {
            a.smoother = b.smoother = GMG_SMOOTHER_MULTICOLOR_GS; a.block_rows = b.block_rows = 0; a.gs_omega = b.gs_omega = 1.0;
            return configEqual(a, b);
        };
        if (parked_.engine && !configEqual(want, parked_.createdWith)) { gmg_destroy(parked_.engine); parked_ = ParkedEngine(); }
        if (parked_.engine || fallbackPair(want, createdWith_)) swapParked();
        else { gmg_destroy(engine_); engine_ = nullptr; }
    }
    if (!engine_) {
        int rc = gmg_create(&want, &engine_);
        if (rc != GMG_OK) { engine_ = nullptr; err_ = "gmg_create failed (invalid engine configuration)"; return rc; }
        createdWith_ = want;
        uploadedU_.clear();
        systemReady_ = false;
        partRank_ = 0; partWorld_ = 1;
    }
    if (partRank_ != distRank || partWorld_ != distWorld) {
        // (before the hierarchy goes in: a partitioned handle prepares and lays out this rank's share only)
        int rc = gmg_dist_partition(engine_, distRank, distWorld);
        if (rc != GMG_OK) { err_ = gmg_last_error(engine_); return rc; }
        partRank_ = distRank; partWorld_ = distWorld;
        uploadedU_.clear();
        systemReady_ = false;
    }
    return GMG_OK;
}

int MultigridSolver::prepareEngine() {
    int rc = ensureEngine();
    if (rc) return rc;
    // `U` is a public member the caller may replace (set_prolongation_matrices): compare content digests
    std::vector<std::pair<uint64_t, uint64_t>> digU(U.size());
    for (size_t k = 0; k < U.size(); ++k) digU[k] = U[k].digest();
    if (digU != uploadedU_) {
        if ((rc = gmg_set_num_levels(engine_, (int)U.size()))) { err_ = gmg_last_error(engine_); return rc; }
        for (size_t k = 0; k < U.size(); ++k)
            if ((rc = gmg_set_prolongation(engine_, (int)k, U[k].rows(), U[k].cols(), U[k].outerPtr(), U[k].innerPtr(), U[k].valuePtr()))) {
                err_ = gmg_last_error(engine_);
                return rc;
            }
        if (!U.empty() && !fineOrder_.empty() && digU == fineOrderFor_ && (int)fineOrder_.size() == U[0].rows())
            (void)gmg_set_fine_order(engine_, (int)fineOrder_.size(), fineOrder_.data());       // (optional: a refusal only costs locality)
        // the point graph the hierarchy was built from: the engine prepares the structure of the systems to come (gmg_set_fine_graph).  Only
        // for the hierarchy this object built itself -- a caller's own prolongations (set_prolongation_matrices) say nothing about `neigh`
        if (!U.empty() && digU == fineOrderFor_ && neigh.rows() == U[0].rows() && neigh.cols() > 0)
            (void)gmg_set_fine_graph(engine_, neigh.rows(), neigh.cols(), neigh.data.data());
        if (!U.empty() && (rc = gmg_finalize_hierarchy(engine_))) { err_ = gmg_last_error(engine_); return rc; }
        uploadedU_ = digU;
        systemReady_ = false;
    }
    return GMG_OK;
}

int MultigridSolver::ensureSystem(const SparseMatrix& LHS) {
    int rc = prepareEngine();
    if (rc) return rc;
    const std::pair<uint64_t, uint64_t> digLHS = LHS.digest();
    if (exactGsActive_ && (digLHS != exactGsFor_ || !fallbackAllowed())) {       // another system, or the caller chose a smoother since: back to the configured engine
        exactGsActive_ = false;
        if ((rc = prepareEngine())) return rc;
    }
    // known not to contract with the default smoothers UNDER THESE OPTIONS: no second attempt (options changed since: the verdict does not carry over)
    if (!exactGsActive_ && fallbackAllowed() && needsExactGs_.count(std::make_pair(digLHS, configKey()))) {
        exactGsActive_ = true;
        exactGsFor_ = digLHS;
        if ((rc = prepareEngine())) return rc;
    }
    if (!systemReady_ || uploadedLHS_ != digLHS) {
        // mass diagonal for the M / M^-1 norms (multigrid_solver.cpp:1248-1264)
        std::vector<double> md(M.cols(), 0.0);
        for (int j = 0; j < M.cols(); ++j)
            for (int p = M.outerPtr()[j]; p < M.outerPtr()[j + 1]; ++p) if (M.innerPtr()[p] == j) md[j] = M.valuePtr()[p];
        if ((rc = gmg_set_mass(engine_, (int)md.size(), md.data()))) { err_ = gmg_last_error(engine_); return rc; }
        if ((rc = gmg_set_system(engine_, LHS.rows(), LHS.outerPtr(), LHS.innerPtr(), LHS.valuePtr()))) { err_ = gmg_last_error(engine_); systemReady_ = false; return rc; }
        uploadedLHS_ = digLHS;
        systemReady_ = true;
        ++systemGeneration_;
    }
    return GMG_OK;
}

int MultigridSolver::prepareSystem(const SparseMatrix& LHS, gmg_handle* handle, long* generation) {
    if (U.empty()) { err_ = "the hierarchy has no levels (mesh smaller than lower_bound?)"; return GMG_ERR_STATE; }
    int rc = ensureSystem(LHS);
    if (rc) return rc;
    if (handle) *handle = engine_;
    if (generation) *generation = systemGeneration_;
    return GMG_OK;
}

// multigrid_solver.cpp:1059-1088.  A and k other than (the current LHS, 0) would need their own upload; the solve loop
// only ever calls it with (LHS, ..., 0), which is what this supports.  Returns 0.0 like the reference.
double MultigridSolver::multiGridVCycleGS(SparseMatrix& A, MatrixXd& b, MatrixXd& x, int k, bool) {
    if (k != 0) { err_ = "multiGridVCycleGS: only the top-level call (k = 0) is exposed"; std::cout << "ERROR! " << err_ << std::endl; return 0.0; }
    if (ensureSystem(A) != GMG_OK) { std::cout << "ERROR! " << err_ << std::endl; return 0.0; }
    if (gmg_vcycle(engine_, b.data.data(), x.data.data(), b.cols()) != GMG_OK) { err_ = gmg_last_error(engine_); std::cout << "ERROR! " << err_ << std::endl; }
    return 0.0;
}

// multigrid_solver.cpp:1194-1226 (tol / isDebug unused there as well).  Runs the engine's level-0 smoother:
// multicolour Gauss-Seidel, i.e. the reference sweep in a colour-permuted order.
void MultigridSolver::GaussSeidelSmoother(SparseMatrix& LHS, MatrixXd& rhs, MatrixXd& x, int maxIter_, double, bool) {
    if (ensureSystem(LHS) != GMG_OK) { std::cout << "ERROR! " << err_ << std::endl; return; }
    if (gmg_smooth(engine_, 0, rhs.data.data(), x.data.data(), rhs.cols(), maxIter_) != GMG_OK) { err_ = gmg_last_error(engine_); std::cout << "ERROR! " << err_ << std::endl; }
}

// multigrid_solver.cpp:1228-1277
double MultigridSolver::residualCheck(const SparseMatrix& A, const MatrixXd& b, const MatrixXd& x, int type) {
    double out = std::numeric_limits<double>::quiet_NaN();
    if (ensureSystem(A) != GMG_OK) { std::cout << "ERROR! " << err_ << std::endl; return out; }
    if (gmg_residual_norm(engine_, b.data.data(), x.data.data(), b.cols(), type, &out) != GMG_OK) { err_ = gmg_last_error(engine_); std::cout << "ERROR! " << err_ << std::endl; }
    return out;
}

// multigrid_solver.cpp:1279-1485
void MultigridSolver::solve(SparseMatrix& LHS, MatrixXd& rhs, MatrixXd& x, int solverType) {
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    convergence.reserve(maxIter);                                        // :1283 (never cleared upstream either)
    if (solverType == 0) {
        // direct LDL^T on the host (the reference's SimplicialLLT/LDLT comparison branch, :1287-1320)
        if (verbose) std::cout << "Solve the linear system using direct solver";
        auto t0 = clk::now();
        int64_t nnzL = 0;
        std::vector<double> out(x.data.size());
        int rc = gmg_host_ldlt_solve(LHS.rows(), LHS.outerPtr(), LHS.innerPtr(), LHS.valuePtr(), rhs.data.data(), rhs.cols(), out.data(), &nnzL);
        if (rc != GMG_OK) { err_ = "direct solve failed (zero pivot)"; std::cout << "ERROR! " << err_ << std::endl; return; }
        x.data = out;
        solverTiming["direct_total"] = ms(t0);
        return;
    }
    if (solverType == 1) { std::cout << "Pardiso is not available in this build" << std::endl; return; }   // :1364
    if (solverType != 2) return;
    if (verbose) std::cout << "Solve the linear system using OUR Multigrid solver \n";
    if (!isSmootherGaussSeidel) return;                                   // :1380
    if (cycleType != 0) {                                                 // F-/W-cycles index U out of range upstream
        err_ = "only cycle_type = 0 (V-cycle) is supported";
        std::cout << "ERROR! " << err_ << std::endl;
        return;
    }
    if (U.empty()) { err_ = "the hierarchy has no levels (mesh smaller than lower_bound?)"; std::cout << "ERROR! " << err_ << std::endl; return; }
    auto t_total = clk::now();
    if (verbose) std::cout << "Reducing system" << std::endl;
    if (ensureSystem(LHS) != GMG_OK) { std::cout << "ERROR! " << err_ << std::endl; return; }
    for (const char* key : {"reduction", "coarsest_solve"}) { double v = 0; if (gmg_get_timing(engine_, key, &v) == GMG_OK) solverTiming[key] = v; }
    if (verbose) std::cout << "V-CYCLE \n";
    std::vector<double> conv(2 * (size_t)std::max(maxIter, 1));
    int iters = 0;
    double residue = std::numeric_limits<double>::max();
    // The initial guess, in case the solve has to be repeated (gmg_solve overwrites x with the last iterate, as the reference does).
    // The reference's binding always passes x0 = rhs (core.cpp:69; initialGuessIsRhs says so and costs nothing); any other caller's guess is
    // compared in full (a guess that differs from rhs in a few entries only -- a point source, a local perturbation -- must survive a retry)
    // and copied when it differs.
    bool x0IsRhs = x.data.size() == rhs.data.size();
    const bool knownRhs = initialGuessIsRhs && x0IsRhs;        // the caller says so: x is output only
    if (x0IsRhs && !knownRhs) x0IsRhs = std::memcmp(x.data.data(), rhs.data.data(), sizeof(double) * x.data.size()) == 0;
    std::vector<double> x0;
    if (!x0IsRhs) x0 = x.data;
    auto run = [&]() {
        return knownRhs ? gmg_solve_x0_rhs(engine_, rhs.data.data(), x.data.data(), rhs.cols(), accuracy, stoppingCriteria, maxIter, &iters, &residue, conv.data())
                        : gmg_solve(engine_, rhs.data.data(), x.data.data(), rhs.cols(), accuracy, stoppingCriteria, maxIter, &iters, &residue, conv.data());
    };
    int rc = run();
    if (rc != GMG_OK && rc != GMG_DIVERGED) { err_ = gmg_last_error(engine_); std::cout << "ERROR! " << err_ << std::endl; return; }
#ifdef GMG_TESTING
    // test build of the pybind module only (tests/_native/, build_bindings.sh): pretend the default smoothers did not contract, so that the
    // handling below can be exercised on a system every smoother solves.  The production module and libgravomg_hip.so hold no such hook.
    if (testReportDiverged > 0 && rc == GMG_OK && !exactGsActive_ && fallbackAllowed()) { --testReportDiverged; rc = GMG_DIVERGED; }
#endif
    // The engine's default smoothers (over-relaxed multicolour sweep on level 0, block-hybrid sweeps below) are not the reference's
    // Gauss-Seidel and carry no convergence guarantee for every SPD matrix.  If the iteration did not contract (GMG_DIVERGED), solve
    // again from the same initial guess with Gauss-Seidel in colour order on EVERY level -- the reference's update in a permuted
    // order, convergent for every SPD matrix.  The switch is scoped to THIS system (keyed by its content digest: the next solve on
    // another matrix runs the configured engine again), never overrides a smoother the caller chose (weighted Jacobi), never touches
    // the public `engineConfig`, and is always reported (message + solverTiming["fallback_exact_gs"]).
    solverTiming["fallback_exact_gs"] = exactGsActive_ ? 1.0 : 0.0;
    solverTiming["diverged"] = 0.0;
    const bool canFallBack = !exactGsActive_ && fallbackAllowed();
    if (rc == GMG_DIVERGED && canFallBack) {
        std::cout << "gravomg: the default smoothers did not contract on this system (residue " << residue << " after " << iters
                  << " cycles): solving again with Gauss-Seidel in colour order on every level" << std::endl;
        exactGsActive_ = true;
        exactGsFor_ = uploadedLHS_;
        needsExactGs_.insert(std::make_pair(uploadedLHS_, configKey()));
        if (!knownRhs) { if (x0IsRhs) x.data = rhs.data; else x.data = x0; }
        if (ensureSystem(LHS) != GMG_OK) { std::cout << "ERROR! " << err_ << std::endl; return; }
        rc = run();
        if (rc != GMG_OK && rc != GMG_DIVERGED) { err_ = gmg_last_error(engine_); std::cout << "ERROR! " << err_ << std::endl; return; }
        solverTiming["fallback_exact_gs"] = 1.0;
    }
    if (rc == GMG_DIVERGED) {
        // the reference would hand back its last iterate without a word; so does this, with a word -- unless the iteration blew up
        // (residue not finite or 1e4 x the smallest one seen: the engine stopped it), which no caller can use
        solverTiming["diverged"] = 1.0;
        std::cout << "gravomg: the V-cycle iteration did not contract (residue " << residue << " after " << iters << " cycles); x holds the last iterate" << std::endl;
        double blown = 0;
        (void)gmg_get_timing(engine_, "blown_up", &blown);
        if (blown != 0.0 || !std::isfinite(residue)) { err_ = "the V-cycle iteration diverged (residue " + std::to_string(residue) + ")"; std::cout << "ERROR! " << err_ << std::endl; }
    }
    for (int i = 0; i < iters; ++i) {
        convergence.push_back({conv[2 * i], conv[2 * i + 1]});          // :1414
        if (verbose) printf("%d,%f,%.14f \n", i + 1, conv[2 * i], conv[2 * i + 1]);
    }
    double cyc = 0;
    gmg_get_timing(engine_, "cycles", &cyc);
    solverTiming["cycles"] = cyc;                                         // :1445-1448
    solverTiming["solver_total"] = ms(t_total);
    solverTiming["iterations"] = iters * 1.0;
    solverTiming["residue"] = residue;
}

// gravomg/src/utility.cpp:106-131: header row "experiment,<keys...>" (map order), then one row per call
void writeTiming(const std::map<std::string, double>& timing, const std::string& experiment, const std::string& filename, const bool& writeHeaders) {
    std::ofstream f;
    if (writeHeaders) f.open(filename.c_str()); else f.open(filename.c_str(), std::ios_base::app);
    if (!f.is_open()) { std::cout << "Unable to open timing file: " << filename << std::endl; return; }
    if (writeHeaders) {
        f << "experiment";
        for (auto const& t : timing) f << ',' << t.first;
        f << "\n";
    }
    f << experiment;
    for (auto const& t : timing) f << ',' << t.second;
    f << "\n";
}

// gravomg/src/utility.cpp:133-149
void writeConvergence(const std::vector<std::tuple<double, double>>& convergence, const std::string& filename) {
    std::ofstream f(filename.c_str());
    if (!f.is_open()) { std::cout << "Unable to open convergence file: " << filename << std::endl; return; }
    f << "time,residue\n";
    for (auto const& it : convergence) f << std::get<0>(it) << ',' << std::get<1>(it) << "\n";
}

}  // namespace MGBS
