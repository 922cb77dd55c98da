/*
 * gravomg_oracle.c -- CPU restatement of the Gravo MG V-cycle hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gravo_mg_amd/ may link, import or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * there only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" by the reference -- the reference ships no tests, no
 * golden vectors and cannot be compiled here (Eigen + libigl are absent).
 * This file follows the reference line by line (citations below are into /root/reference)
 * and is itself cross-checked against an independent scipy implementation
 * (tests/test_oracle.py, tests/golden/make_golden.py).
 *
 * Data conventions follow the reference: sparse matrices are CSC (Eigen::SparseMatrix<double>
 * default, int32 indices, sorted inner indices), dense multi-vectors are column-major n x d
 * (Eigen::MatrixXd).
 *
 *   orc_gauss_seidel      gravomg/src/multigrid_solver.cpp:1194-1226
 *   orc_residual          gravomg/src/multigrid_solver.cpp:1066
 *   orc_restrict          gravomg/src/multigrid_solver.cpp:1069
 *   orc_prolong_add       gravomg/src/multigrid_solver.cpp:1082
 *   orc_vcycle            gravomg/src/multigrid_solver.cpp:1059-1088
 *   orc_residual_check    gravomg/src/multigrid_solver.cpp:1228-1277
 *   orc_galerkin          gravomg/src/multigrid_solver.cpp:1387-1392 (+ coarsest factor :1401)
 *   orc_solve             gravomg/src/multigrid_solver.cpp:1408-1419, 1445-1448
 *                         (x0 = rhs comes from gravomg_bindings/src/cpp/core.cpp:69, caller's job)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    int nrows, ncols;
    int *colptr;   /* ncols+1 */
    int *rowidx;   /* nnz, ascending within a column */
    double *val;   /* nnz */
} orc_csc;

typedef struct {
    int n;
    int *perm;      /* new -> old (RCM) */
    int *first;     /* first stored column of row i in the skyline (new numbering) */
    int64_t *rptr;  /* start of row i's skyline storage */
    double *l;      /* unit lower skyline, row-wise: l[rptr[i] + (j-first[i])] for first[i] <= j < i */
    double *dg;     /* D */
    double *work;   /* n */
} orc_ldlt;

typedef struct {
    int L;                 /* number of transfer levels = U.size() */
    orc_csc *A;            /* A[0] = LHS (set per solve), A[1..L] = Galerkin operators (Abar) */
    orc_csc *U;            /* U[0..L-1], U[k] is n_k x n_{k+1} */
    double *mass;          /* lumped mass diagonal, n_0 (may be NULL until set) */
    orc_ldlt coarse;
    int pre_iters, post_iters;
} orc_hier;

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void csc_free(orc_csc *m) {
    free(m->colptr); free(m->rowidx); free(m->val);
    memset(m, 0, sizeof(*m));
}

static void csc_copy_in(orc_csc *m, int nrows, int ncols, const int *colptr, const int *rowidx,
                        const double *val) {
    csc_free(m);
    int nnz = colptr[ncols];
    m->nrows = nrows; m->ncols = ncols;
    m->colptr = (int *)malloc(sizeof(int) * (size_t)(ncols + 1));
    m->rowidx = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    m->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    memcpy(m->colptr, colptr, sizeof(int) * (size_t)(ncols + 1));
    memcpy(m->rowidx, rowidx, sizeof(int) * (size_t)nnz);
    memcpy(m->val, val, sizeof(double) * (size_t)nnz);
}

/* ---------------------------------------------------------------- sparse helpers */

/* B = A^T (CSC -> CSC); rows of the result come out sorted. */
static void csc_transpose(const orc_csc *a, orc_csc *t) {
    int nnz = a->colptr[a->ncols];
    t->nrows = a->ncols; t->ncols = a->nrows;
    t->colptr = (int *)calloc((size_t)a->nrows + 1, sizeof(int));
    t->rowidx = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    t->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    for (int p = 0; p < nnz; ++p) t->colptr[a->rowidx[p] + 1]++;
    for (int i = 0; i < a->nrows; ++i) t->colptr[i + 1] += t->colptr[i];
    int *next = (int *)malloc(sizeof(int) * (size_t)(a->nrows > 0 ? a->nrows : 1));
    memcpy(next, t->colptr, sizeof(int) * (size_t)a->nrows);
    for (int j = 0; j < a->ncols; ++j)
        for (int p = a->colptr[j]; p < a->colptr[j + 1]; ++p) {
            int q = next[a->rowidx[p]]++;
            t->rowidx[q] = j;
            t->val[q] = a->val[p];
        }
    free(next);
}

static int cmp_int(const void *a, const void *b) {
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* C = A * B, column by column (Gustavson): C[:,j] = sum_k A[:,k] * B[k,j], k ascending as stored.
 * This is the accumulation order of Eigen's conservative sparse*sparse product. */
static void csc_matmul(const orc_csc *a, const orc_csc *b, orc_csc *c) {
    int m = a->nrows, n = b->ncols;
    double *acc = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    int *mark = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int *list = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; ++i) mark[i] = -1;
    size_t cap = (size_t)a->colptr[a->ncols] + (size_t)b->colptr[b->ncols] + 16;
    c->nrows = m; c->ncols = n;
    c->colptr = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    c->rowidx = (int *)malloc(sizeof(int) * cap);
    c->val = (double *)malloc(sizeof(double) * cap);
    size_t nnz = 0;
    c->colptr[0] = 0;
    for (int j = 0; j < n; ++j) {
        int cnt = 0;
        for (int pb = b->colptr[j]; pb < b->colptr[j + 1]; ++pb) {
            int k = b->rowidx[pb];
            double bkj = b->val[pb];
            for (int pa = a->colptr[k]; pa < a->colptr[k + 1]; ++pa) {
                int i = a->rowidx[pa];
                if (mark[i] != j) { mark[i] = j; list[cnt++] = i; acc[i] = 0.0; }
                acc[i] += a->val[pa] * bkj;
            }
        }
        qsort(list, (size_t)cnt, sizeof(int), cmp_int);
        if (nnz + (size_t)cnt > cap) {
            cap = (nnz + (size_t)cnt) * 2;
            c->rowidx = (int *)realloc(c->rowidx, sizeof(int) * cap);
            c->val = (double *)realloc(c->val, sizeof(double) * cap);
        }
        for (int t = 0; t < cnt; ++t) { c->rowidx[nnz] = list[t]; c->val[nnz] = acc[list[t]]; ++nnz; }
        c->colptr[j + 1] = (int)nnz;
    }
    free(acc); free(mark); free(list);
}

/* ---------------------------------------------------------------- coarsest direct solver
 * The reference uses Eigen::SimplicialLDLT (multigrid_solver.h:145, multigrid_solver.cpp:1075,1401).
 * Eigen is a third-party dependency absent from /root/reference; this is a plain LDL^T (skyline
 * storage, reverse Cuthill-McKee ordering) -- a direct solve, identical up to rounding. */

static void rcm_order(const orc_csc *a, int *perm /* new->old */) {
    int n = a->ncols;
    int *deg = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    char *seen = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; ++j) deg[j] = a->colptr[j + 1] - a->colptr[j];
    int head = 0, tail = 0;
    while (tail < n) {
        int start = -1;
        for (int j = 0; j < n; ++j)
            if (!seen[j] && (start < 0 || deg[j] < deg[start])) start = j;
        seen[start] = 1; perm[tail++] = start;
        while (head < tail) {
            int v = perm[head++];
            int cnt = 0;
            for (int p = a->colptr[v]; p < a->colptr[v + 1]; ++p) {
                int w = a->rowidx[p];
                if (!seen[w]) { seen[w] = 1; tmp[cnt++] = w; }
            }
            /* insertion sort by degree */
            for (int i = 1; i < cnt; ++i) {
                int w = tmp[i], k = i - 1;
                while (k >= 0 && deg[tmp[k]] > deg[w]) { tmp[k + 1] = tmp[k]; --k; }
                tmp[k + 1] = w;
            }
            for (int i = 0; i < cnt; ++i) perm[tail++] = tmp[i];
        }
    }
    for (int i = 0; i < n / 2; ++i) { int t = perm[i]; perm[i] = perm[n - 1 - i]; perm[n - 1 - i] = t; }
    free(deg); free(seen); free(tmp);
}

static void ldlt_free(orc_ldlt *f) {
    free(f->perm); free(f->first); free(f->rptr); free(f->l); free(f->dg); free(f->work);
    memset(f, 0, sizeof(*f));
}

static int ldlt_factor(const orc_csc *a, orc_ldlt *f) {
    ldlt_free(f);
    int n = a->ncols;
    f->n = n;
    f->perm = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    f->first = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    f->rptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    f->dg = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    f->work = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    rcm_order(a, f->perm);
    int *inv = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) inv[f->perm[i]] = i;
    for (int i = 0; i < n; ++i) {
        int old = f->perm[i], fi = i;
        for (int p = a->colptr[old]; p < a->colptr[old + 1]; ++p) {
            int j = inv[a->rowidx[p]];
            if (j < fi) fi = j;
        }
        f->first[i] = fi;
    }
    f->rptr[0] = 0;
    for (int i = 0; i < n; ++i) f->rptr[i + 1] = f->rptr[i] + (i - f->first[i]);
    f->l = (double *)calloc((size_t)(f->rptr[n] > 0 ? f->rptr[n] : 1), sizeof(double));
    /* scatter the lower triangle of P A P^T (A symmetric: column old == row old) */
    for (int i = 0; i < n; ++i) {
        int old = f->perm[i];
        f->dg[i] = 0.0;
        for (int p = a->colptr[old]; p < a->colptr[old + 1]; ++p) {
            int j = inv[a->rowidx[p]];
            if (j < i) f->l[f->rptr[i] + (j - f->first[i])] = a->val[p];
            else if (j == i) f->dg[i] = a->val[p];
        }
    }
    /* row-wise skyline LDL^T */
    for (int i = 0; i < n; ++i) {
        double *li = f->l + f->rptr[i];
        int fi = f->first[i];
        for (int j = fi; j < i; ++j) {
            double *lj = f->l + f->rptr[j];
            int fj = f->first[j];
            int k0 = fi > fj ? fi : fj;
            double s = li[j - fi];
            for (int k = k0; k < j; ++k) s -= li[k - fi] * lj[k - fj];   /* li[k] holds L_ik * d_k here */
            li[j - fi] = s;                                            /* = L_ij * d_j */
        }
        double d = f->dg[i];
        for (int j = fi; j < i; ++j) {
            double t = li[j - fi];
            double lij = t / f->dg[j];
            d -= t * lij;
            li[j - fi] = lij;
        }
        f->dg[i] = d;
    }
    free(inv);
    return 0;
}

/* NOTE on the inner loop above: while row i is being eliminated li[k] temporarily stores
 * L_ik*d_k for k<j (the "s" values), which is what the dot product with the finished row j
 * (true L_jk) needs: (L_ij d_j) = a_ij - sum_k (L_ik d_k) L_jk. */

static void ldlt_solve(const orc_ldlt *f, const double *b, double *x) {
    int n = f->n;
    double *y = f->work;
    for (int i = 0; i < n; ++i) y[i] = b[f->perm[i]];
    for (int i = 0; i < n; ++i) {
        const double *li = f->l + f->rptr[i];
        int fi = f->first[i];
        double s = y[i];
        for (int j = fi; j < i; ++j) s -= li[j - fi] * y[j];
        y[i] = s;
    }
    for (int i = 0; i < n; ++i) y[i] /= f->dg[i];
    for (int i = n - 1; i >= 0; --i) {
        const double *li = f->l + f->rptr[i];
        int fi = f->first[i];
        double yi = y[i];
        for (int j = fi; j < i; ++j) y[j] -= li[j - fi] * yi;
    }
    for (int i = 0; i < n; ++i) x[f->perm[i]] = y[i];
}

/* ---------------------------------------------------------------- hot-path operators */

/* multigrid_solver.cpp:1194-1226.  Forward lexicographic Gauss-Seidel that walks COLUMN k of the
 * CSC matrix (== row k because LHS is symmetric), sums the off-diagonal terms in stored order,
 * then divides by coeffRef(k,k).  For d > 1 the sweep is repeated per right-hand-side column with
 * the columns outermost (:1213).  `tol` / `isDebug` of the reference are unused there too. */
void orc_gauss_seidel(int n, const int *colptr, const int *rowidx, const double *val,
                      const double *rhs, double *x, int d, int max_iter) {
    for (int it = 0; it < max_iter; ++it)
        for (int c = 0; c < d; ++c) {
            const double *bc = rhs + (size_t)c * n;
            double *xc = x + (size_t)c * n;
            for (int k = 0; k < n; ++k) {
                double sum = 0.0, diag = 0.0;
                for (int p = colptr[k]; p < colptr[k + 1]; ++p) {
                    int r = rowidx[p];
                    if (r != k) sum += val[p] * xc[r];
                    else diag = val[p];           /* coeffRef(k,k): 0 if absent (then x -> inf, as upstream) */
                }
                xc[k] = (bc[k] - sum) / diag;
            }
        }
}

/* y = A*x with Eigen's column-major sparse * dense kernel: per rhs column, j outer, scatter-axpy. */
static void csc_mult(int nrows, int ncols, const int *colptr, const int *rowidx, const double *val,
                     const double *x, int ldx, double *y, int ldy, int d) {
    for (int c = 0; c < d; ++c) {
        double *yc = y + (size_t)c * ldy;
        const double *xc = x + (size_t)c * ldx;
        for (int i = 0; i < nrows; ++i) yc[i] = 0.0;
        for (int j = 0; j < ncols; ++j) {
            double xj = xc[j];
            for (int p = colptr[j]; p < colptr[j + 1]; ++p) yc[rowidx[p]] += val[p] * xj;
        }
    }
}

/* multigrid_solver.cpp:1066   res = b - A*x  (product evaluated into a temporary, then subtracted) */
void orc_residual(int n, const int *colptr, const int *rowidx, const double *val,
                  const double *b, const double *x, int d, double *res) {
    csc_mult(n, n, colptr, rowidx, val, x, n, res, n, d);
    for (size_t i = 0; i < (size_t)n * d; ++i) res[i] = b[i] - res[i];
}

/* multigrid_solver.cpp:1069   resRest = U^T * res : row c of U^T is column c of the CSC U. */
void orc_restrict(int nf, int nc, const int *colptr, const int *rowidx, const double *val,
                  const double *res, int d, double *rc) {
    for (int c = 0; c < d; ++c)
        for (int j = 0; j < nc; ++j) {
            double s = 0.0;
            for (int p = colptr[j]; p < colptr[j + 1]; ++p) s += val[p] * res[(size_t)c * nf + rowidx[p]];
            rc[(size_t)c * nc + j] = s;
        }
}

/* multigrid_solver.cpp:1082   x = x + U*eps */
void orc_prolong_add(int nf, int nc, const int *colptr, const int *rowidx, const double *val,
                     const double *eps, int d, double *x) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)nf * (size_t)d);
    csc_mult(nf, nc, colptr, rowidx, val, eps, nc, tmp, nf, d);
    for (size_t i = 0; i < (size_t)nf * d; ++i) x[i] = x[i] + tmp[i];
    free(tmp);
}

/* multigrid_solver.cpp:1228-1277.  type 0: ||Ax-b||/||b||, 1: M^-1 norm, 2: M norm, 3: ||AX-B||_F.
 * Types 0-2 return the max over rhs columns. */
double orc_residual_check(int n, const int *colptr, const int *rowidx, const double *val,
                          const double *mass, const double *b, const double *x, int d, int type) {
    double *r = (double *)malloc(sizeof(double) * (size_t)n * (size_t)d);
    csc_mult(n, n, colptr, rowidx, val, x, n, r, n, d);
    for (size_t i = 0; i < (size_t)n * d; ++i) r[i] = r[i] - b[i];
    double out = 0.0;
    if (type == 3) {
        double s = 0.0;
        for (size_t i = 0; i < (size_t)n * d; ++i) s += r[i] * r[i];
        out = sqrt(s);
    } else {
        for (int c = 0; c < d; ++c) {
            const double *rc = r + (size_t)c * n, *bc = b + (size_t)c * n;
            double n1 = 0.0, n2 = 0.0, v;
            for (int i = 0; i < n; ++i) {
                double w = type == 0 ? 1.0 : (type == 1 ? 1.0 / mass[i] : mass[i]);
                n1 += (rc[i] * w) * rc[i];
                n2 += (bc[i] * w) * bc[i];
            }
            v = type == 0 ? sqrt(n1) / sqrt(n2) : sqrt(n1 / n2);
            if (c == 0 || v > out) out = v;
        }
    }
    free(r);
    return out;
}

/* ---------------------------------------------------------------- hierarchy object */

orc_hier *orc_create(int L) {
    orc_hier *h = (orc_hier *)calloc(1, sizeof(orc_hier));
    h->L = L;
    h->A = (orc_csc *)calloc((size_t)L + 1, sizeof(orc_csc));
    h->U = (orc_csc *)calloc((size_t)(L > 0 ? L : 1), sizeof(orc_csc));
    h->pre_iters = 2; h->post_iters = 2;     /* gravomg_bindings/src/gravomg/core.py:10 */
    return h;
}

void orc_destroy(orc_hier *h) {
    if (!h) return;
    for (int k = 0; k <= h->L; ++k) csc_free(&h->A[k]);
    for (int k = 0; k < h->L; ++k) csc_free(&h->U[k]);
    free(h->A); free(h->U); free(h->mass);
    ldlt_free(&h->coarse);
    free(h);
}

void orc_set_smoothing(orc_hier *h, int pre, int post) { h->pre_iters = pre; h->post_iters = post; }

void orc_set_prolongation(orc_hier *h, int k, int nf, int nc, const int *colptr, const int *rowidx,
                          const double *val) {
    csc_copy_in(&h->U[k], nf, nc, colptr, rowidx, val);
}

void orc_set_mass(orc_hier *h, int n, const double *mass) {
    free(h->mass);
    h->mass = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(h->mass, mass, sizeof(double) * (size_t)n);
}

/* multigrid_solver.cpp:1387-1392 and :1401.  Abar[1] = U0^T * LHS * U0 (evaluated left to right),
 * Abar[k] = U_{k-1}^T * Abar[k-1] * U_{k-1}; then factor Abar[L].  Returns ms spent in
 * {reduction, coarsest factor} through out2 (may be NULL). */
int orc_galerkin(orc_hier *h, int n, const int *colptr, const int *rowidx, const double *val,
                 double *out2) {
    double t0 = now_ms();
    csc_copy_in(&h->A[0], n, n, colptr, rowidx, val);
    for (int k = 1; k <= h->L; ++k) {
        orc_csc ut = {0}, uta = {0};
        csc_transpose(&h->U[k - 1], &ut);
        csc_matmul(&ut, &h->A[k - 1], &uta);
        csc_free(&h->A[k]);
        csc_matmul(&uta, &h->U[k - 1], &h->A[k]);
        csc_free(&ut); csc_free(&uta);
    }
    double t1 = now_ms();
    ldlt_factor(&h->A[h->L], &h->coarse);
    double t2 = now_ms();
    if (out2) { out2[0] = t1 - t0; out2[1] = t2 - t1; }
    return 0;
}

int orc_level_size(const orc_hier *h, int k) { return h->A[k].ncols; }
int orc_level_nnz(const orc_hier *h, int k) { return h->A[k].colptr ? h->A[k].colptr[h->A[k].ncols] : 0; }
void orc_get_level(const orc_hier *h, int k, int *colptr, int *rowidx, double *val) {
    const orc_csc *a = &h->A[k];
    memcpy(colptr, a->colptr, sizeof(int) * (size_t)(a->ncols + 1));
    memcpy(rowidx, a->rowidx, sizeof(int) * (size_t)a->colptr[a->ncols]);
    memcpy(val, a->val, sizeof(double) * (size_t)a->colptr[a->ncols]);
}

void orc_coarse_solve(orc_hier *h, const double *rc, double *e, int d) {
    int n = h->coarse.n;
    for (int c = 0; c < d; ++c) ldlt_solve(&h->coarse, rc + (size_t)c * n, e + (size_t)c * n);
}

/* multigrid_solver.cpp:1059-1088 */
static void vcycle_rec(orc_hier *h, int k, const double *b, double *x, int d) {
    const orc_csc *A = &h->A[k], *U = &h->U[k];
    int n = A->ncols, nc = U->ncols;
    orc_gauss_seidel(n, A->colptr, A->rowidx, A->val, b, x, d, h->pre_iters);          /* :1063 */
    double *res = (double *)malloc(sizeof(double) * (size_t)n * (size_t)d);
    orc_residual(n, A->colptr, A->rowidx, A->val, b, x, d, res);                        /* :1066 */
    double *rc = (double *)malloc(sizeof(double) * (size_t)nc * (size_t)d);
    orc_restrict(n, nc, U->colptr, U->rowidx, U->val, res, d, rc);                      /* :1069 */
    double *eps = (double *)calloc((size_t)nc * (size_t)d, sizeof(double));             /* :1072-1073 */
    if (k == h->L - 1) orc_coarse_solve(h, rc, eps, d);                                 /* :1074-1076 */
    else vcycle_rec(h, k + 1, rc, eps, d);                                              /* :1078 */
    orc_prolong_add(n, nc, U->colptr, U->rowidx, U->val, eps, d, x);                    /* :1082 */
    orc_gauss_seidel(n, A->colptr, A->rowidx, A->val, b, x, d, h->post_iters);          /* :1085 */
    free(res); free(rc); free(eps);
}

void orc_vcycle(orc_hier *h, const double *b, double *x, int d) { vcycle_rec(h, 0, b, x, d); }

/* multigrid_solver.cpp:1408-1419: do { V-cycle; residualCheck; log } while (residue > accuracy && it < maxIter).
 * conv (may be NULL) receives (elapsed_ms, residue) pairs, capacity max_iter.  Returns iterations. */
int orc_solve(orc_hier *h, const double *rhs, double *x, int d, double accuracy, int stop_type,
              int max_iter, double *conv, double *residue_out) {
    const orc_csc *A = &h->A[0];
    double t0 = now_ms(), residue;
    int it = 0;
    do {
        orc_vcycle(h, rhs, x, d);
        residue = orc_residual_check(A->ncols, A->colptr, A->rowidx, A->val, h->mass, rhs, x, d, stop_type);
        if (conv) { conv[2 * it] = now_ms() - t0; conv[2 * it + 1] = residue; }
        ++it;
    } while (residue > accuracy && it < max_iter);
    if (residue_out) *residue_out = residue;
    return it;
}
