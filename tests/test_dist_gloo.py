"""Multi-rank orchestration (gravo_mg_amd/dist.py) on CPU: world_size 2, 3 and 8 with the gloo backend and a numpy
backend for the local steps.  Checks that the row-partitioned V-cycle with an all-gather after every colour gives
the SAME iterates as the single-process multicolour V-cycle (the colours are global, so the result must not depend
on the number of ranks), and the same residual norms / iteration count.  CPU only."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyBackend:
    """Reference local steps on numpy arrays in DEVICE numbering (colour-major, classes padded to 64*world)."""

    def __init__(self, P, plan, rank, world, pre=2, post=2):
        import torch
        self.rank, self.world, self.pre_iters, self.post_iters = rank, world, pre, post
        self.new2old = plan["new2old"]; self.color_begin = plan["color_begin"]; self.n_pad = plan["n_pad"]
        self.d = P.rhs.shape[1]
        n2o = self.new2old
        real = n2o >= 0
        A = sp.csr_matrix(P.lhs)
        # device-numbered operator incl. padding rows (identity rows, zero rhs)
        rows = np.nonzero(real)[0]
        Pm = sp.csr_matrix((np.ones(rows.size), (rows, n2o[rows])), shape=(self.n_pad, A.shape[0]))
        self.A = (Pm @ A @ Pm.T + sp.diags((~real).astype(float))).tocsr()
        self.diag = self.A.diagonal()
        self.U0 = (Pm @ sp.csr_matrix(P.U[0])).tocsr()               # rows in device numbering, coarse cols natural
        self.mass = np.ones(self.n_pad); self.mass[rows] = P.mass[n2o[rows]]
        # replicated coarse part: exact recursion with natural-order GS on the Galerkin operators (deterministic)
        self.As = [A]
        for U in P.U:
            self.As.append(sp.csr_matrix(U.T @ self.As[-1] @ U))
        self.Us = P.U
        self.lu = spla.splu(sp.csc_matrix(self.As[-1]))
        self.Pm = Pm
        self.x = torch.zeros(self.n_pad * self.d, dtype=torch.float64)
        self.b = torch.zeros_like(self.x); self.r = torch.zeros_like(self.x)

    def stream_context(self):
        import contextlib
        return contextlib.nullcontext()

    def _v(self, t):
        return t.numpy().reshape(self.d, self.n_pad).T          # view: n_pad x d

    def load(self, b, x0):
        self._v(self.b)[:] = self.Pm @ b
        self._v(self.x)[:] = self.Pm @ x0

    all_rows = False

    def own(self, c):
        lo, hi = self.color_begin[c], self.color_begin[c + 1]
        if self.all_rows:
            return slice(lo, hi)
        piece = (hi - lo) // self.world
        return slice(lo + self.rank * piece, lo + (self.rank + 1) * piece)

    def _on_all_rows(self, fn, *a):
        self.all_rows = True
        try:
            return fn(*a)
        finally:
            self.all_rows = False

    def residual_all(self):
        self._on_all_rows(self.residual_own)

    def prolong_all(self):
        self._on_all_rows(self.prolong_own)

    def norm_all(self, type):
        return self._on_all_rows(self.norm_partial, type)

    def smooth_color(self, c):
        s = self.own(c)
        x, b = self._v(self.x), self._v(self.b)
        Ar = self.A[s]
        x[s] = (b[s] - (Ar @ x - self.diag[s, None] * x[s])) / self.diag[s, None]

    def residual_own(self):
        x, b, r = self._v(self.x), self._v(self.b), self._v(self.r)
        for c in range(len(self.color_begin) - 1):
            s = self.own(c)
            r[s] = b[s] - self.A[s] @ x

    def _coarse(self, k, b):
        from tests.test_oracle import _scipy_vcycle
        if k == len(self.Us):
            return np.column_stack([self.lu.solve(b[:, c]) for c in range(b.shape[1])])
        return _scipy_vcycle(self.As, self.Us, self.lu, b, np.zeros_like(b), k)

    def coarse_cycle(self):
        rc = self.U0.T @ self._v(self.r)
        self.e1 = self._coarse(1, rc)

    def prolong_own(self):
        x = self._v(self.x)
        for c in range(len(self.color_begin) - 1):
            s = self.own(c)
            x[s] += self.U0[s] @ self.e1

    def norm_partial(self, type):
        x, b = self._v(self.x), self._v(self.b)
        out = np.zeros(2 * self.d)
        for c in range(len(self.color_begin) - 1):
            s = self.own(c)
            r = self.A[s] @ x - b[s]
            w = np.ones(r.shape[0]) if type in (0, 3) else (self.mass[s] if type == 2 else 1.0 / self.mass[s])
            out[0::2] += (w[:, None] * r ** 2).sum(0)
            out[1::2] += (w[:, None] * b[s] ** 2).sum(0)
        return out

    def solution(self):
        return self.Pm.T @ self._v(self.x)


def _worker(rank, world, port, kind, out_dir, mode):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from gravo_mg_amd import cabi
    from gravo_mg_amd.dist import DistVCycle, HaloPlan
    from tests import problems
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = problems.torus_problem(48, 40, kind, 40) if kind != "smoothing" else problems.torus_problem(40, 36, "smoothing", 60)
        plan = cabi.host_plan_level(P.lhs, mode=0, row_align=64 * world)
        be = NumpyBackend(P, plan, rank, world)
        halo = None
        if mode == "halo":
            A = sp.csr_matrix(P.lhs)
            halo = HaloPlan(A.indptr, A.indices, plan["new2old"], plan["color_begin"], plan["n_pad"], world, rank, P.rhs.shape[1])
            assert 0 < halo.published_rows < P.lhs.shape[0]
        dv = DistVCycle(be, replicate=mode == "replicate", halo=halo)
        be.load(P.rhs, P.rhs)
        hist = []
        for _ in range(3):
            dv.vcycle()
            hist.append([dv.residual_norm(t) for t in range(4)])
        dv.gather_solution()
        x3 = be.solution()
        be.load(P.rhs, P.rhs)
        it, res, residues = dv.solve(1e-4, 2, 50)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), x3=x3, hist=np.array(hist), it=it, res=res, x=be.solution(), ncoll=dv.n_collectives,
                 ncolors=dv.n_colors)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _single(kind):
    from gravo_mg_amd import cabi
    from gravo_mg_amd.dist import DistVCycle
    from tests import problems
    P = problems.torus_problem(48, 40, kind, 40) if kind != "smoothing" else problems.torus_problem(40, 36, "smoothing", 60)
    plan = cabi.host_plan_level(P.lhs, mode=0, row_align=64)
    be = NumpyBackend(P, plan, 0, 1)
    dv = DistVCycle(be)
    be.load(P.rhs, P.rhs)
    hist = []
    for _ in range(3):
        dv.vcycle()
        hist.append([dv.residual_norm(t) for t in range(4)])
    x3 = be.solution()
    be.load(P.rhs, P.rhs)
    it, res, _ = dv.solve(1e-4, 2, 50)
    return P, x3, np.array(hist), it, res, be.solution()


@pytest.mark.parametrize("world,kind,mode", [(2, "poisson", "replicate"), (3, "smoothing", "replicate"), (2, "poisson", "partitioned"),
                                             (3, "smoothing", "partitioned"), (2, "poisson", "halo"), (3, "smoothing", "halo"),
                                             (8, "poisson", "halo"), (8, "smoothing", "partitioned")])      # the target's P = 8
def test_row_partitioned_vcycle_is_independent_of_world_size(world, kind, mode, tmp_path, cabi, oracle):
    import torch.multiprocessing as mp
    P, x3, hist, it, res, x = _single(kind)
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), mode), nprocs=world, join=True)
    outs = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    for o in outs:
        # every rank ends with the complete, identical iterate ...
        np.testing.assert_allclose(o["x3"], outs[0]["x3"], rtol=0, atol=0)
        # ... equal to the single-process multicolour V-cycle (ordering of padded classes differs, the algebra does not)
        assert np.linalg.norm(P.lhs @ (o["x3"] - x3)) <= 1e-11 * spla.norm(P.lhs) * np.linalg.norm(x3)
        np.testing.assert_allclose(o["hist"], hist, rtol=1e-6, atol=1e-13)
        assert int(o["it"]) == it and abs(float(o["res"]) - res) <= 1e-6 * res + 1e-12
        assert np.linalg.norm(o["x"] - x) <= 1e-6 * np.linalg.norm(x)
        # exchanges per cycle: (pre+post) sweeps * colours; row-partitioned residual / prolongation / norm add their own
        C, d = int(o["ncolors"]), P.rhs.shape[1]
        cycles = 3 + it
        want = {"replicate": cycles * (4 * C * d), "partitioned": cycles * (6 * C * d) + (3 * 4 + it),
                # halo: one small exchange per colour sweep (all columns at once), r all-gathered per colour segment, one
                # exchange after the prolongation, the norm all-reduce, and x completed once after the cycles / the solve
                "halo": cycles * (4 * C + C * d + 1) + (3 * 4 + it) + 2 * C * d}[mode]
        assert int(o["ncoll"]) == want
    # and it is the reference's answer: the oracle's residual check agrees on the distributed solution
    chk = oracle.residual_check(P.lhs, P.mass, P.rhs, outs[0]["x"], 2)
    assert chk <= 1e-4 and abs(chk - float(outs[0]["res"])) <= 1e-3 * chk + 1e-9
