"""CPU model of the DEFAULT engine's V-cycle, assembled from the oracle's operators (test infrastructure).

The engine's smoothers are parallel re-orderings of the reference's Gauss-Seidel (multigrid_solver.cpp:1194-1226), so the
oracle's natural-order V-cycle is a different iteration.  This model is the SAME iteration as the device's, built only from
the oracle's arithmetic plus the orderings the device reports:

  * level 0 (colour-major): multicolour SOR in the device's colour order -- per colour c,
    x[rows_c] += omega * (b - A x)[rows_c] / diag[rows_c]   with (b - A x) from ``oracle.residual`` (multigrid_solver.cpp:1066).
    omega = 1 is the reference's update ``x_i <- (b_i - sum_{j != i} a_ij x_j) / a_ii`` applied colour by colour;
  * blocked levels: x += T^-1 (b - A x), T = D + strict lower triangle of A restricted to the block diagonal in device
    order (Jacobi between blocks, Gauss-Seidel inside), residual from the oracle, triangular solve from scipy;
  * residual / restriction / prolongation / coarsest solve / Galerkin operators: the oracle's (gravomg_oracle.c).

A per-cycle comparison against this model checks the composition of the whole cycle (level order, zero initial coarse
guess, which vectors feed which operator) to rounding -- also for iterations that do not contract.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class VcycleModel:
    def __init__(self, eng, U, mass, lhs, oracle, omega, pre=2, post=2):
        self.oracle, self.omega, self.pre, self.post = oracle, float(omega), pre, post
        self.L = len(U)
        self.U = [sp.csc_matrix(u) for u in U]
        self.O = oracle.Hierarchy(U, mass, pre_iters=pre, post_iters=post)
        self.O.set_system(lhs)
        self.A = [sp.csc_matrix(self.O.level_operator(k)) for k in range(self.L + 1)]
        self.sm = []
        for k in range(self.L):
            blocks = eng.level_blocks(k)
            n2o, cb = eng.level_ordering(k)
            if blocks is None:
                rows = [n2o[cb[c]:cb[c + 1]] for c in range(len(cb) - 1)]
                self.sm.append(("colour", [r[r >= 0] for r in rows], self.A[k].diagonal()))
            else:
                bb, _ = blocks
                real = n2o >= 0
                blk = np.repeat(np.arange(len(bb) - 1), np.diff(bb))[real]
                order = n2o[real]
                Ap = self.A[k].tocsr()[order][:, order].tocoo()
                keep = (blk[Ap.row] == blk[Ap.col]) & (Ap.col <= Ap.row)
                data = Ap.data[keep]
                T = sp.csr_matrix((data, (Ap.row[keep], Ap.col[keep])), shape=Ap.shape)
                self.sm.append(("block", order, T))

    def smooth(self, k, b, x, iters, omega=None):
        kind, a, c = self.sm[k]
        x = np.array(x, dtype=np.float64, copy=True)
        om = self.omega if (k == 0 and omega is None) else (1.0 if omega is None else omega)
        for _ in range(iters):
            if kind == "colour":
                for rows in a:
                    r = self.oracle.residual(self.A[k], b, x)
                    x[rows] += om * r[rows] / (c[rows][:, None] if x.ndim == 2 else c[rows])
            else:
                r = self.oracle.residual(self.A[k], b, x)
                step = np.empty_like(x)
                step[a] = spla.spsolve_triangular(c, r[a], lower=True)
                x = x + step
        return x

    def vcycle(self, b, x, k=0):
        x = self.smooth(k, b, x, self.pre)                                   # :1063
        r = self.oracle.residual(self.A[k], b, x)                            # :1066
        rc = self.oracle.restrict(self.U[k], r)                              # :1069
        if k == self.L - 1:
            e = self.O.coarse_solve(rc)                                      # :1075
        else:
            e = self.vcycle(rc, np.zeros_like(rc), k + 1)                    # :1072-1078
        x = self.oracle.prolong_add(self.U[k], e, x)                         # :1082
        return self.smooth(k, b, x, self.post)                               # :1085
