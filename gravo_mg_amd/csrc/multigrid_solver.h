// multigrid_solver.h -- host-side mirror of the reference's C++ interface on the V-cycle path.
//
// Same class name, namespace, public method names, argument meaning and public tunables as
// gravomg/include/gravomg/multigrid_solver.h:54-170 (MGBS::MultigridSolver), so the pybind11 shim
// (bindings.cpp, mirroring gravomg_bindings/src/cpp/core.cpp) and a C++ caller read the same.  Eigen is not
// available, so the matrix types are minimal stand-ins with Eigen's storage conventions:
//   MGBS::SparseMatrix  == Eigen::SparseMatrix<double>  (CSC, int32, sorted inner indices)
//   MGBS::MatrixXd      == Eigen::MatrixXd              (column-major n x d)
//   MGBS::MatrixXi      == Eigen::MatrixXi, but ROW-major here (numpy's default; only `neigh` uses it)
// All numerics of the hot path run in libgravomg_hip.so through include/gravomg_hip.h.
#ifndef GRAVOMG_AMD_MULTIGRID_SOLVER_H
#define GRAVOMG_AMD_MULTIGRID_SOLVER_H

#include <set>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/gravomg_hip.h"

/* Enums of the reference, gravomg/include/gravomg/multigrid_solver.h:35-52 */
enum Hierarchy { OURS = 0, SIG21 = 1 };
enum Sampling { FASTDISK = 0, POISSONDISK = 1, FPS = 2, RANDOM = 3, MIS = 4 };
enum Weighting { BARYCENTRIC = 0, UNIFORM = 1, INVDIST = 2 };

namespace MGBS {

// environment GMG_TRACE=ctor (the library's one trace switch, csrc/host_sparse.hpp::EnvSwitches): phases of the construction on stderr
inline bool ctorTrace() { static const bool v = [] { const char* t = std::getenv("GMG_TRACE"); return t && std::strstr(t, "ctor"); }(); return v; }

struct SparseMatrix {
    int rows_ = 0, cols_ = 0;
    std::vector<int> outer;      // cols_+1
    std::vector<int> inner;      // nnz
    std::vector<double> values;  // nnz
    // Optional non-owning view of caller storage (like Eigen::Map<SparseMatrix>): when set, the three arrays above are
    // unused.  The pybind11 shim maps the caller's scipy arrays this way instead of copying 250 MB per solve.
    const int* outerView = nullptr;
    const int* innerView = nullptr;
    const double* valuesView = nullptr;
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    const int* outerPtr() const { return outerView ? outerView : outer.data(); }
    const int* innerPtr() const { return outerView ? innerView : inner.data(); }
    const double* valuePtr() const { return outerView ? valuesView : values.data(); }
    int nonZeros() const { return (outerView || !outer.empty()) ? outerPtr()[cols_] : 0; }
    // 128-bit content digest (shape, pattern, values), threaded: how the solver recognises an unchanged matrix without
    // keeping a copy of it.
    std::pair<uint64_t, uint64_t> digest() const;
};

struct MatrixXd {
    int rows_ = 0, cols_ = 0;
    std::vector<double> data;    // column-major
    MatrixXd() {}
    MatrixXd(int r, int c) : rows_(r), cols_(c), data((size_t)r * c, 0.0) {}
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    double& operator()(int i, int j) { return data[(size_t)j * rows_ + i]; }
    double operator()(int i, int j) const { return data[(size_t)j * rows_ + i]; }
};

struct MatrixXi {
    int rows_ = 0, cols_ = 0;
    std::vector<int> data;       // row-major
    int rows() const { return rows_; }
    int cols() const { return cols_; }
};

class MultigridSolver {
public:
    // gravomg/include/gravomg/multigrid_solver.h:58; M is the (diagonal) mass matrix
    MultigridSolver(MatrixXd& V, MatrixXi& neigh, SparseMatrix& M);
    // the same, taking over the caller's storage (the pybind shim's converted copies: 160 MB at 3 M vertices)
    MultigridSolver(MatrixXd&& V, MatrixXi&& neigh, SparseMatrix&& M);
    ~MultigridSolver();
    MultigridSolver(const MultigridSolver&) = delete;
    MultigridSolver& operator=(const MultigridSolver&) = delete;

    /* Hierarchy-related methods (multigrid_solver.cpp:43-60) */
    void buildHierarchy();

    /* Multigrid solver pieces, same names as multigrid_solver.h:77-83 */
    double multiGridVCycleGS(SparseMatrix& A, MatrixXd& b, MatrixXd& x, int k, bool isDebug = true);
    void GaussSeidelSmoother(SparseMatrix& LHS, MatrixXd& rhs, MatrixXd& x, int maxIter, double tol, bool isDebug = true);
    double residualCheck(const SparseMatrix& A, const MatrixXd& b, const MatrixXd& x, int type);

    /* Core solver function (multigrid_solver.h:90): solverType 2 = multigrid (the hot path), 0 = direct LDL^T,
       1 = Pardiso (not available: message, as an upstream build without MKL prints). */
    void solve(SparseMatrix& LHS, MatrixXd& rhs, MatrixXd& x, int solverType = 2);

    /* Data, same names as the reference's public members */
    MatrixXd V;
    MatrixXd normals;
    MatrixXi neigh;
    SparseMatrix M;
    std::vector<size_t> DoF;
    std::vector<SparseMatrix> U;                 // prolongation operators (what the solver sees)
    std::vector<std::vector<int>> samples;       // samples[k]: fine index of every point of level k+1 (multigrid_solver.cpp:128)
    std::vector<std::vector<size_t>> nearestSource;   // nearestSource[k]: the level-(k+1) point every level-k point clusters to (:171)
    std::vector<MatrixXd> levelV;                // positions of the points of level k+1 (the reference fills it with debug only, :241)
    std::vector<MatrixXi> levelE;                // (multigrid_solver.h:99) filled by the SIG06 hierarchy only upstream: always empty here
    std::vector<std::vector<int>> noTriFoundMap; // (:100) with debug: one zero vector per level, as upstream (:291 is its only write)
    std::vector<std::vector<std::vector<int>>> allTriangles;   // (:101) with debug: the candidate triangles of every level (:281)
    std::vector<MatrixXd> levelN;                // (:102) never filled upstream: always empty
    int cycleType = 0;                           // 0: V-cycle (1 F / 2 W are rejected: broken upstream)
    bool isSmootherGaussSeidel = false;          // set by the shim (core.cpp:57); solve() does nothing without it
    bool sig06 = false;
    bool checkVoronoi = true;
    bool nested = false;
    int stoppingCriteria = 0;                    // multigrid_solver.h:131
    int maxIter = 50;
    int lowBound = 1000;
    double ratio = 8;
    Sampling samplingStrategy = FASTDISK;
    Weighting weightingScheme = BARYCENTRIC;
    int preIters = 2;
    int postIters = 2;
    double accuracy = 5e-4;                      // multigrid_solver.h:144
    bool verbose = true;
    bool debug = false;
    bool ablation = false;
    int ablationNumPoints = 3;
    bool ablationRandom = false;

    /* Logging and timing (multigrid_solver.h:157-159) */
    std::map<std::string, double> hierarchyTiming;
    std::map<std::string, double> solverTiming;
    std::vector<std::tuple<double, double>> convergence;

    /* MI355X engine knobs (not in the reference) */
    gmg_config engineConfig;
    /* Set by a caller that KNOWS the initial guess of the next solve() is the right-hand side (the binding: core.cpp:69): x is then
       output only -- it need not hold a copy of rhs -- and the engine copies rhs to x on the device (gmg_solve_x0_rhs). */
    bool initialGuessIsRhs = false;
    /* One process per GPU (not in the reference): this object is rank distRank of distWorld ranks of a row-partitioned job.  With distWorld > 1
       the engine lays out and keeps only this rank's rows of levels 0-1 (gmg_dist_partition) and only the collective solve of
       gravomg.MultigridSolver.enable_distributed() runs on it; engineConfig.row_align must be 64 * distWorld. */
    int distRank = 0, distWorld = 1;
    /* Creates the device engine and hands it the hierarchy (`U`) now rather than inside the first solve(); optional. */
    int prepareEngine();
    const char* lastError() const;
    void clearError() { err_.clear(); }
    /* Multi-GPU hook (not in the reference): brings the engine to "system set" for LHS without solving and hands out the engine's
       C-ABI handle (owned by this object) and a counter that changes whenever the device layout was rebuilt, so that a caller can
       drive the engine-driven multi-GPU cycle (gmg_p2p_*, include/gravomg_hip.h) on it.  Returns a gmg status. */
    int prepareSystem(const SparseMatrix& LHS, gmg_handle* handle, long* generation);
#ifdef GMG_TESTING
    int testReportDiverged = 0;      // test build of the pybind module only (tests/_native/): the next N default-engine solves report GMG_DIVERGED
#endif

private:
    int ensureEngine();
    bool fallbackAllowed() const;
    uint64_t configKey() const;
    int ensureSystem(const SparseMatrix& LHS);
    gmg_handle engine_ = nullptr;
    std::vector<std::pair<uint64_t, uint64_t>> uploadedU_;     // digests of what the engine holds
    std::vector<int> fineOrder_;                               // breadth-first order of the points from buildHierarchy (may be empty) ...
    std::vector<std::pair<uint64_t, uint64_t>> fineOrderFor_;  // ... and the digests of the U buildHierarchy made (what fineOrder_ and `neigh` belong to)
    std::pair<uint64_t, uint64_t> uploadedLHS_{0, 0};
    bool systemReady_ = false;
    long systemGeneration_ = 0;                                // bumped by every gmg_set_system (prepareSystem)
    bool exactGsActive_ = false;                               // Gauss-Seidel on every level instead of the configured smoothers ...
    std::pair<uint64_t, uint64_t> exactGsFor_{0, 0};           // ... for the system with this digest only (solve())
    gmg_config createdWith_;
    int partRank_ = 0, partWorld_ = 1;                         // what the engine was told (gmg_dist_partition)
    // The engine that is NOT in use -- the exact-GS one while the configured one runs, or the other way round -- with its state: an
    // application that alternates between a system needing the fallback and others switches engines instead of rebuilding them.
    struct ParkedEngine {
        gmg_handle engine = nullptr;
        gmg_config createdWith;
        std::vector<std::pair<uint64_t, uint64_t>> uploadedU;
        std::pair<uint64_t, uint64_t> uploadedLHS{0, 0};
        bool systemReady = false;
    } parked_;
    void swapParked();
    std::set<std::pair<std::pair<uint64_t, uint64_t>, uint64_t>> needsExactGs_;     // (system content digest, configKey()) pairs whose default iteration did not contract: straight to the fallback next time
    std::string err_;
};

/* gravomg/src/utility.cpp:106-149, same CSV layout */
void writeTiming(const std::map<std::string, double>& timing, const std::string& experiment, const std::string& filename, const bool& writeHeaders = false);
void writeConvergence(const std::vector<std::tuple<double, double>>& convergence, const std::string& filename);

}  // namespace MGBS

#endif
