// eigen_baseline.cpp -- TEST / MEASUREMENT INFRASTRUCTURE (only bench.py's cpu_baseline leg loads it).
//
// "the reference's own Eigen CPU path timed on the GPU box's host cores".  The reference cannot be built here
// (Eigen + libigl from the network), and this image ships no Eigen -- but a host that does have the Eigen headers can at least
// time the SAME EXPRESSIONS the reference's hot path evaluates (Eigen::SparseMatrix<double> column-major, InnerIterator /
// coeffRef Gauss-Seidel, `b - A * x`, `U.transpose() * res`, `x + U * eps`, SimplicialLDLT), which is what the plain-C port in
// gravomg_oracle.c can only imitate.  This file is that hook: with <Eigen/Sparse> on the include path it is a restatement of
//   gravomg/src/multigrid_solver.cpp:1059-1088 (V-cycle), :1194-1226 (Gauss-Seidel), :1228-1277 (residual norms, type 0 / 2 / 3),
//   :1387-1392 (Galerkin products), :1401 (coarsest factorisation), :1408-1419 (solve loop)
// written against Eigen's API; without it the library only answers orc_eigen_available() = 0.  One thread, like the reference
// (omp_set_num_threads(1), :86-87).  Not a checker: parity is established against gravomg_oracle.c.
#include <chrono>
#include <cmath>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Sparse>)
#define GMG_HAVE_EIGEN 1
#endif
#endif

extern "C" int orc_eigen_available() {
#ifdef GMG_HAVE_EIGEN
    return 1;
#else
    return 0;
#endif
}

#ifdef GMG_HAVE_EIGEN
#include <Eigen/Sparse>
#include <Eigen/SparseCholesky>

namespace {
using SpMat = Eigen::SparseMatrix<double>;      // column-major, int indices: the reference's type (multigrid_solver.h:105-108)
using Mat = Eigen::MatrixXd;

struct State {
    std::vector<SpMat> U, Abar;                 // Abar[0] unused, like the reference
    Eigen::SimplicialLDLT<SpMat> coarsest;
    Eigen::VectorXd mass;
    int pre = 2, post = 2;
};

SpMat from_csc(int rows, int cols, const int* colptr, const int* rowidx, const double* val) {
    return Eigen::Map<const SpMat>(rows, cols, colptr[cols], colptr, rowidx, val);
}

void gauss_seidel(SpMat& A, const Mat& rhs, Mat& x, int sweeps) {             // :1194-1226
    for (int s = 0; s < sweeps; ++s)
        for (int c = 0; c < x.cols(); ++c)
            for (int k = 0; k < A.outerSize(); ++k) {
                double sum = 0.0;
                for (SpMat::InnerIterator it(A, k); it; ++it)
                    if (it.row() != k) sum += it.value() * x(it.row(), c);
                x(k, c) = (rhs(k, c) - sum) / A.coeffRef(k, k);
            }
}

void vcycle(State& S, SpMat& A, const Mat& b, Mat& x, int k) {                // :1059-1088
    gauss_seidel(A, b, x, S.pre);
    Mat res = b - A * x;
    Mat resRest = S.U[k].transpose() * res;
    Mat eps = Mat::Zero(resRest.rows(), resRest.cols());
    if (k == (int)S.U.size() - 1) eps = S.coarsest.solve(resRest);
    else vcycle(S, S.Abar[k + 1], resRest, eps, k + 1);
    x = x + S.U[k] * eps;
    gauss_seidel(A, b, x, S.post);
}

double residual_check(const State& S, const SpMat& A, const Mat& b, const Mat& x, int type) {      // :1228-1277
    double worst = 0.0, frob = 0.0;
    for (int c = 0; c < b.cols(); ++c) {
        Eigen::VectorXd r = A * x.col(c) - b.col(c);
        double v = 0.0;
        if (type == 0) v = r.norm() / b.col(c).norm();
        else if (type == 1) v = std::sqrt(r.dot(r.cwiseQuotient(S.mass)) / b.col(c).dot(b.col(c).cwiseQuotient(S.mass)));
        else if (type == 2) v = std::sqrt(r.dot(S.mass.cwiseProduct(r)) / b.col(c).dot(S.mass.cwiseProduct(b.col(c))));
        else frob += r.squaredNorm();
        if (v > worst) worst = v;
    }
    return type == 3 ? std::sqrt(frob) : worst;
}
}  // namespace

extern "C" {
void* orc_eigen_create(int L) { State* S = new State(); S->U.resize(L); S->Abar.resize(L + 1); return S; }
void orc_eigen_destroy(void* h) { delete (State*)h; }
void orc_eigen_set_prolongation(void* h, int k, int nf, int nc, const int* colptr, const int* rowidx, const double* val) { ((State*)h)->U[k] = from_csc(nf, nc, colptr, rowidx, val); }
void orc_eigen_set_mass(void* h, int n, const double* m) { ((State*)h)->mass = Eigen::Map<const Eigen::VectorXd>(m, n); }

// Galerkin products + coarsest factorisation of one system; times[0] = "reduction", times[1] = "coarsest_solve" (ms).  The LHS is
// kept as Abar[0].
int orc_eigen_galerkin(void* h, int n, const int* colptr, const int* rowidx, const double* val, double* times) {
    State& S = *(State*)h;
    using clk = std::chrono::steady_clock;
    S.Abar[0] = from_csc(n, n, colptr, rowidx, val);
    auto t0 = clk::now();
    for (size_t k = 0; k < S.U.size(); ++k) S.Abar[k + 1] = S.U[k].transpose() * S.Abar[k] * S.U[k];       // :1387-1392
    auto t1 = clk::now();
    S.coarsest.compute(S.Abar[S.U.size()]);                                                                // :1401
    auto t2 = clk::now();
    if (times) { times[0] = std::chrono::duration<double, std::milli>(t1 - t0).count(); times[1] = std::chrono::duration<double, std::milli>(t2 - t1).count(); }
    return S.coarsest.info() == Eigen::Success ? 0 : -1;
}

// do { V-cycle; residualCheck } while (residue > tol && it < max_iter), :1408-1419; conv: (elapsed ms, residue) pairs
int orc_eigen_solve(void* h, const double* rhs, double* x, int d, double tol, int stop_type, int max_iter, double* conv, double* residue_out) {
    State& S = *(State*)h;
    using clk = std::chrono::steady_clock;
    const int n = (int)S.Abar[0].rows();
    Mat b = Eigen::Map<const Mat>(rhs, n, d), X = Eigen::Map<const Mat>(x, n, d);
    int it = 0;
    double residue = 0.0;
    auto t0 = clk::now();
    do {
        vcycle(S, S.Abar[0], b, X, 0);
        residue = residual_check(S, S.Abar[0], b, X, stop_type);
        if (conv) { conv[2 * it] = std::chrono::duration<double, std::milli>(clk::now() - t0).count(); conv[2 * it + 1] = residue; }
        ++it;
    } while (residue > tol && it < max_iter);
    Eigen::Map<Mat>(x, n, d) = X;
    if (residue_out) *residue_out = residue;
    return it;
}
}  // extern "C"
#endif
