// bindings.cpp -- pybind11 module `gravomg_bindings`: the reference's Python-facing class, same constructor
// signature (22 positional arguments) and method list as gravomg_bindings/src/cpp/core.cpp:13-180, implemented
// over MGBS::MultigridSolver (multigrid_solver.h) -> C-ABI -> HIP kernels.
//
// pybind11's Eigen casters are not usable (no Eigen), so scipy sparse matrices are unpacked by hand the way
// pybind11/eigen/matrix.h does it (obj -> csc_matrix -> indptr / indices / data) and numpy arrays are taken
// column-major (forcecast), i.e. all arguments are copied in and results copied out, like upstream.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "multigrid_solver.h"

namespace py = pybind11;
using DenseIn = py::array_t<double, py::array::f_style | py::array::forcecast>;
using IntIn = py::array_t<int, py::array::c_style | py::array::forcecast>;

namespace {

MGBS::MatrixXd to_dense(const DenseIn& a) {
    if (a.ndim() != 1 && a.ndim() != 2) throw std::invalid_argument("expected a 1-D or 2-D float array");
    MGBS::MatrixXd m;
    m.rows_ = (int)a.shape(0); m.cols_ = a.ndim() == 2 ? (int)a.shape(1) : 1;
    m.data.assign(a.data(), a.data() + (size_t)m.rows_ * m.cols_);        // one pass (no zero fill first)
    return m;
}

py::array_t<double> from_dense(const MGBS::MatrixXd& m) {
    py::array_t<double, py::array::f_style> out({(py::ssize_t)m.rows(), (py::ssize_t)m.cols()});
    std::memcpy(out.mutable_data(), m.data.data(), sizeof(double) * m.data.size());
    return std::move(out);
}

MGBS::SparseMatrix to_sparse(const py::object& obj) {
    py::object sp = py::module_::import("scipy.sparse");
    py::object csc = sp.attr("csc_matrix")(obj);
    csc.attr("sum_duplicates")();
    csc.attr("sort_indices")();
    auto shape = csc.attr("shape").cast<std::pair<py::ssize_t, py::ssize_t>>();
    auto indptr = csc.attr("indptr").cast<py::array_t<int, py::array::forcecast>>();
    auto indices = csc.attr("indices").cast<py::array_t<int, py::array::forcecast>>();
    auto data = csc.attr("data").cast<py::array_t<double, py::array::forcecast>>();
    MGBS::SparseMatrix m;
    m.rows_ = (int)shape.first; m.cols_ = (int)shape.second;
    m.outer.assign(indptr.data(), indptr.data() + indptr.size());
    m.inner.assign(indices.data(), indices.data() + indices.size());
    m.values.assign(data.data(), data.data() + data.size());
    return m;
}

// A system matrix mapped in place (no copy).  The reference's API receives scipy CSR (gravomg/core.py:74-77) and pybind11
// converts it to Eigen's column-major storage on every call (core.cpp:68) -- ~100 ms of single-threaded work at 3 M
// vertices, twice the whole GPU solve.  The engine consumes the OUTER vectors of the compressed storage it is handed as the
// rows of the operator (for the symmetric matrices the algorithm is defined for, multigrid_solver.cpp:1200-1208, rows and
// columns coincide).  So: CSR input is mapped as it is -- its outer vectors ARE the rows, for any matrix, and residual() is then
// exact for unsymmetric input too, like upstream's A*x; CSC input is mapped as it is when it is symmetric -- verified on
// EVERY entry the first time a matrix content is seen (threaded pass, verdict cached by content digest) -- and converted to
// CSR otherwise (a matrix unsymmetric in a few rows, e.g. overwritten Dirichlet rows, must not be worked on as its
// transpose); every other format goes through scipy's conversion to CSR.
struct MappedSparse {
    MGBS::SparseMatrix m;
    std::vector<py::object> keep;      // the numpy arrays the view points into
};

// EVERY entry (i, j) has a partner (j, i) with the same value (to 1e-10 relative): one threaded pass, a binary search per
// entry (linear where a row's indices are not sorted).  A random probe would pass a matrix that is unsymmetric in a few
// rows -- Dirichlet rows overwritten without their columns -- and the solver would then work on the transpose.
bool is_symmetric(int n, const int* ptr, const int* idx, const double* val) {
    const int64_t nnz = ptr[n];
    if (nnz == 0) return true;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int T = (int)std::min<int64_t>(hw, nnz / 65536 + 1);
    std::atomic<bool> ok{true};
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi && ok.load(std::memory_order_relaxed); ++i)
            for (int p = ptr[i]; p < ptr[i + 1]; ++p) {
                const int j = idx[p];
                if (j < 0 || j >= n) { ok = false; return; }
                if (j == i) continue;
                const int* b = idx + ptr[j];
                const int* e = idx + ptr[j + 1];
                const int* q = std::lower_bound(b, e, i);
                if (q == e || *q != i) q = std::find(b, e, i);          // unsorted row: linear
                if (q == e) { ok = false; return; }
                const double x = val[p], y = val[q - idx];
                if (std::abs(x - y) > 1e-10 * std::max(std::abs(x), std::abs(y))) { ok = false; return; }
            }
    };
    if (T <= 1) work(0, n);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, (int)((int64_t)n * t / T), (int)((int64_t)n * (t + 1) / T));
        for (auto& t : th) t.join();
    }
    return ok.load();
}

// verdict of the last matrix checked, keyed by its content digest: a repeated solve() on the same lhs pays the check once
struct SymmetryCache {
    std::pair<uint64_t, uint64_t> key{0, 0};
    bool valid = false, symmetric = false;
};

MappedSparse map_system_matrix(const py::object& obj, SymmetryCache& cache) {
    py::object sp = py::module_::import("scipy.sparse");
    py::object mat = obj;
    bool is_csc = sp.attr("isspmatrix_csc")(mat).cast<bool>();
    const bool is_csr = !is_csc && sp.attr("isspmatrix_csr")(mat).cast<bool>();
    if (!is_csc && !is_csr) mat = sp.attr("csr_matrix")(mat);
    for (int attempt = 0; attempt < 2; ++attempt) {
        MappedSparse out;
        auto shape = mat.attr("shape").cast<std::pair<py::ssize_t, py::ssize_t>>();
        auto indptr = mat.attr("indptr").cast<py::array_t<int, py::array::c_style | py::array::forcecast>>();
        auto indices = mat.attr("indices").cast<py::array_t<int, py::array::c_style | py::array::forcecast>>();
        auto data = mat.attr("data").cast<py::array_t<double, py::array::c_style | py::array::forcecast>>();
        if (attempt == 0 && is_csc) {
            bool sym = shape.first == shape.second;
            if (sym) {
                MGBS::SparseMatrix view;
                view.rows_ = (int)shape.first; view.cols_ = (int)shape.second;
                view.outerView = indptr.data(); view.innerView = indices.data(); view.valuesView = data.data();
                const auto key = view.digest();
                if (!(cache.valid && cache.key == key)) {
                    cache.symmetric = is_symmetric((int)shape.first, indptr.data(), indices.data(), data.data());
                    cache.key = key; cache.valid = true;
                }
                sym = cache.symmetric;
            }
            if (!sym) {
                mat = sp.attr("csr_matrix")(mat);      // columns are not rows here: the real conversion
                is_csc = false;
                continue;
            }
        }
        out.m.rows_ = (int)shape.first; out.m.cols_ = (int)shape.second;
        out.m.outerView = indptr.data(); out.m.innerView = indices.data(); out.m.valuesView = data.data();
        out.keep = {mat, indptr, indices, data};
        return out;
    }
    throw std::runtime_error("could not map the system matrix");
}

py::object from_sparse(const MGBS::SparseMatrix& m) {
    py::object sp = py::module_::import("scipy.sparse");
    py::array_t<double> data(m.values.size(), m.values.data());
    py::array_t<int> indices(m.inner.size(), m.inner.data());
    py::array_t<int> indptr(m.outer.size(), m.outer.data());
    return sp.attr("csc_matrix")(py::make_tuple(data, indices, indptr), py::arg("shape") = py::make_tuple(m.rows(), m.cols()));
}

MGBS::MatrixXi to_int(const IntIn& a) {
    if (a.ndim() != 2) throw std::invalid_argument("neighbors must be an n x K integer array");
    MGBS::MatrixXi m;
    m.rows_ = (int)a.shape(0); m.cols_ = (int)a.shape(1);
    m.data.assign(a.data(), a.data() + a.size());
    return m;
}

}  // namespace

class MultigridSolver {
public:
    // Same argument list as gravomg_bindings/src/cpp/core.cpp:20-26.
    MultigridSolver(DenseIn positions, IntIn neighbors, py::object mass, double ratio, int low_bound, int cycle_type, double tolerance,
                    int stopping_criteria, int pre_iters, int post_iters, int max_iter, bool check_voronoi, bool nested,
                    Sampling sampling_strategy, Weighting weighting, bool sig06, DenseIn normals, bool verbose, bool debug, bool ablation,
                    int ablation_num_points, bool ablation_random) {
        const bool trace = MGBS::ctorTrace();      // GMG_TRACE=ctor: phases of the construction on stderr
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (!trace) return;
            auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[gravomg ctor] %-28s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
            t0 = now;
        };
        MGBS::MatrixXd V = to_dense(positions);
        lap("positions");
        MGBS::MatrixXi N = to_int(neighbors);
        lap("neighbours");
        MGBS::SparseMatrix M = to_sparse(mass);
        lap("mass");
        if (V.cols() != 3 || N.rows() != V.rows() || M.rows() != V.rows()) throw std::invalid_argument("positions must be n x 3, neighbors n x K, mass n x n");
        solver.reset(new MGBS::MultigridSolver(std::move(V), std::move(N), std::move(M)));
        solver->checkVoronoi = check_voronoi;
        solver->nested = nested;
        solver->samplingStrategy = sampling_strategy;
        solver->weightingScheme = weighting;
        solver->maxIter = max_iter;
        solver->sig06 = sig06;
        solver->ablation = ablation;
        solver->ablationNumPoints = ablation_num_points;
        solver->ablationRandom = ablation_random;
        if (sig06) solver->normals = to_dense(normals);      // only the (out-of-scope) SIG06 hierarchy reads them: no 72 MB copy otherwise
        solver->verbose = verbose;
        solver->debug = debug;
        solver->ratio = ratio;
        solver->lowBound = low_bound;
        lap("solver object");
        solver->buildHierarchy();
        lap("buildHierarchy");
        if (solver->U.empty() && *solver->lastError()) throw std::runtime_error(solver->lastError());
        solver->cycleType = cycle_type;
        solver->accuracy = tolerance;
        solver->stoppingCriteria = stopping_criteria;
        solver->preIters = pre_iters;
        solver->postIters = post_iters;
        solver->isSmootherGaussSeidel = true;
        (void)solver->prepareEngine();       // hierarchy -> device now (part of the construction phase); errors resurface in solve()
        lap("prepareEngine");
    }

    void construct_sig21_hierarchy(py::object) { throw std::runtime_error("the SIG21 comparison hierarchy is out of scope of the MI355X hot-path build"); }
    void toggle_hierarchy(Hierarchy hierarchy) {
        if (hierarchy != OURS) throw std::runtime_error("only Hierarchy.OURS is available in the MI355X hot-path build");
    }

    // core.cpp:68-72: x0 = rhs
    py::array_t<double> solve(py::object lhs, DenseIn rhs) {
        MappedSparse mapped = map_system_matrix(lhs, symCache);
        MGBS::SparseMatrix& A = mapped.m;
        MGBS::MatrixXd b = to_dense(rhs);
        if (A.rows() != A.cols() || A.rows() != b.rows()) throw std::invalid_argument("lhs must be n x n and rhs n x d");
        MGBS::MatrixXd x;                                   // x0 = rhs, formed on the device: no host copy of b here (24 MB per column set at 3 M)
        x.rows_ = b.rows_; x.cols_ = b.cols_;
        x.data.resize(b.data.size());
        solver->clearError();
        solver->initialGuessIsRhs = true;
        solver->solve(A, b, x, 2);
        solver->initialGuessIsRhs = false;
        check();
        return from_dense(x);
    }

    py::array_t<double> direct_solve(py::object lhs, DenseIn rhs, bool pardiso) {
        MGBS::SparseMatrix A = to_sparse(lhs);
        MGBS::MatrixXd b = to_dense(rhs);
        MGBS::MatrixXd x = b;
        solver->clearError();
        if (pardiso) throw std::runtime_error("direct_solve(pardiso=True): Pardiso (MKL) is not part of this build; use pardiso=False (sparse LDL^T)");
        solver->solve(A, b, x, 0);
        check();
        return from_dense(x);
    }

    py::list prolongation_matrices() {
        py::list out;
        for (const auto& u : solver->U) out.append(from_sparse(u));
        return out;
    }

    void set_prolongation_matrices(py::list U) {
        std::vector<MGBS::SparseMatrix> v;
        for (auto item : U) v.push_back(to_sparse(py::reinterpret_borrow<py::object>(item)));
        solver->U = v;
    }

    // core.cpp:90-116.  samples / nearestSource are filled by every build upstream, levelV, allTriangles and noTriFoundMap only
    // with debug = True; levelE belongs to the SIG06 hierarchy and levelN is never filled upstream: both come back empty, as there.
    std::vector<std::vector<int>> sampling_indices() { return solver->samples; }
    std::vector<std::vector<size_t>> nearest_source() { return solver->nearestSource; }
    py::list level_points() {
        py::list out;
        for (const auto& P : solver->levelV) out.append(from_dense(P));
        return out;
    }
    py::list level_edges() { return py::list(); }
    std::vector<std::vector<int>> notrimap() { return solver->noTriFoundMap; }
    std::vector<std::vector<std::vector<int>>> all_triangles() { return solver->allTriangles; }
    py::list coarse_normals() { return py::list(); }

    void write_hierarchy_timing(std::string experiment, std::string file, bool write_headers) { MGBS::writeTiming(solver->hierarchyTiming, experiment, file, write_headers); }
    void write_solver_timing(std::string experiment, std::string file, bool write_headers) { MGBS::writeTiming(solver->solverTiming, experiment, file, write_headers); }
    void write_convergence(std::string file) { MGBS::writeConvergence(solver->convergence, file); }

    double residual(py::object lhs, DenseIn rhs, DenseIn solution, int type = 2) {
        MappedSparse mapped = map_system_matrix(lhs, symCache);
        MGBS::SparseMatrix& A = mapped.m;
        solver->clearError();
        double r = solver->residualCheck(A, to_dense(rhs), to_dense(solution), type);
        check();
        return r;
    }

    // extras (not upstream): timers as dicts, engine knobs
    std::map<std::string, double> solver_timing() { return solver->solverTiming; }
    std::map<std::string, double> hierarchy_timing() { return solver->hierarchyTiming; }
    std::vector<std::tuple<double, double>> convergence() { return solver->convergence; }
    // multi-GPU hook (not upstream; gravomg.MultigridSolver.enable_distributed): system set on this rank's engine, no solve.
    // Returns (C-ABI handle of the engine as an integer -- owned by this object --, layout generation).
    std::tuple<uintptr_t, long> prepare_system(py::object lhs) {
        MappedSparse mapped = map_system_matrix(lhs, symCache);
        gmg_handle h = nullptr;
        long gen = 0;
        solver->clearError();
        if (solver->prepareSystem(mapped.m, &h, &gen) != GMG_OK) { check(); throw std::runtime_error("prepare_system failed"); }
        return std::make_tuple((uintptr_t)h, gen);
    }
    void set_engine_option(const std::string& key, double value) {
        gmg_config& c = solver->engineConfig;
        if (key == "smoother") c.smoother = (int)value;
        else if (key == "jacobi_omega") c.jacobi_omega = value;
        else if (key == "gs_omega") c.gs_omega = value;
        else if (key == "coarse_mode") c.coarse_mode = (int)value;
        else if (key == "use_graph") c.use_graph = (int)value;
        else if (key == "block_rows") c.block_rows = (int)value;
        else if (key == "block_from_level") c.block_from_level = (int)value;
        else if (key == "device") c.device = (int)value;
        else if (key == "row_align") c.row_align = (int)value;
        else if (key == "block_lanes") c.block_lanes = (int)value;
        else if (key == "dist_shard_levels") c.dist_shard_levels = (int)value;
        else if (key == "block_fine") c.block_fine = (int)value;
        else if (key == "dist_rank") solver->distRank = (int)value;
        else if (key == "dist_world") solver->distWorld = (int)value;
        else if (key == "fine_col16") c.fine_col16 = (int)value;
        else if (key == "stream_gate") c.stream_gate = (int)value;
        else if (key == "fuse_restrict_sweep") c.fuse_restrict_sweep = (int)value;
        else if (key == "speculate_head") c.speculate_head = (int)value;
        else if (key == "uniform_slices") c.uniform_slices = (int)value;
        else if (key == "color_ahead") c.color_ahead = (int)value;
        else if (key == "dist_exchange") c.dist_exchange = (int)value;
        else if (key == "inner_precision") c.inner_precision = (int)value;
        else throw std::invalid_argument("unknown engine option: " + key);
    }

#ifdef GMG_TESTING
    void test_report_diverged(int n) { solver->testReportDiverged = n; }
#endif

private:
    // The reference reports problems with printed messages only; the drop-in additionally raises, so that a missing
    // GPU / an unsupported option can never pass silently.
    void check() {
        std::string now = solver->lastError();
        if (!now.empty()) throw std::runtime_error(now);
    }
    std::unique_ptr<MGBS::MultigridSolver> solver;
    SymmetryCache symCache;
};

PYBIND11_MODULE(gravomg_bindings, m) {
    m.doc() = "Multigrid solver bindings (MI355X-native V-cycle hot path)";

    py::enum_<Hierarchy>(m, "Hierarchy").value("OURS", OURS).value("SIG21", SIG21);
    py::enum_<Sampling>(m, "Sampling").value("FASTDISK", FASTDISK).value("POISSONDISK", POISSONDISK).value("FPS", FPS).value("RANDOM", RANDOM).value("MIS", MIS);
    py::enum_<Weighting>(m, "Weighting").value("BARYCENTRIC", BARYCENTRIC).value("UNIFORM", UNIFORM).value("INVDIST", INVDIST);

    py::class_<MultigridSolver>(m, "MultigridSolver")
        .def(py::init<DenseIn, IntIn, py::object, double, int, int, double, int, int, int, int, bool, bool, Sampling, Weighting, bool, DenseIn, bool, bool, bool, int, bool>())
        .def("construct_sig21_hierarchy", &MultigridSolver::construct_sig21_hierarchy, py::arg("F"))
        .def("toggle_hierarchy", &MultigridSolver::toggle_hierarchy, py::arg("hierarchy"))
        .def("solve", &MultigridSolver::solve, py::arg("lhs"), py::arg("rhs"))
        .def("direct_solve", &MultigridSolver::direct_solve, py::arg("lhs"), py::arg("rhs"), py::arg("pardiso"))
        .def("prolongation_matrices", &MultigridSolver::prolongation_matrices)
        .def("set_prolongation_matrices", &MultigridSolver::set_prolongation_matrices, py::arg("U"))
        .def("sampling_indices", &MultigridSolver::sampling_indices)
        .def("level_points", &MultigridSolver::level_points)
        .def("level_edges", &MultigridSolver::level_edges)
        .def("notrimap", &MultigridSolver::notrimap)
        .def("all_triangles", &MultigridSolver::all_triangles)
        .def("coarse_normals", &MultigridSolver::coarse_normals)
        .def("nearest_source", &MultigridSolver::nearest_source)
        .def("write_hierarchy_timing", &MultigridSolver::write_hierarchy_timing, py::arg("experiment"), py::arg("file"), py::arg("write_headers"))
        .def("write_solver_timing", &MultigridSolver::write_solver_timing, py::arg("experiment"), py::arg("file"), py::arg("write_headers"))
        .def("write_convergence", &MultigridSolver::write_convergence, py::arg("file"))
        .def("residual", &MultigridSolver::residual, py::arg("lhs"), py::arg("rhs"), py::arg("solution"), py::arg("type") = 2)
        .def("solver_timing", &MultigridSolver::solver_timing)
        .def("hierarchy_timing", &MultigridSolver::hierarchy_timing)
        .def("convergence", &MultigridSolver::convergence)
        .def("prepare_system", &MultigridSolver::prepare_system, py::arg("lhs"))
        .def("set_engine_option", &MultigridSolver::set_engine_option, py::arg("key"), py::arg("value"))
#ifdef GMG_TESTING
        .def("_test_report_diverged", &MultigridSolver::test_report_diverged, py::arg("n"))
#endif
        ;
}
