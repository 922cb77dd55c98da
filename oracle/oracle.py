"""ctypes view of oracle/libgravomg_oracle.so (the CPU restatement in gravomg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Parity status: unpinned by the reference (it ships no tests and cannot be built here); the restatement is
cross-checked against scipy in tests/test_oracle.py and against committed fixtures in tests/golden/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgravomg_oracle.so")
_ip, _dp, _vp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gravomg_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        l = C.CDLL(_LIB)
        sig = {
            "orc_gauss_seidel": (None, [C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_int, C.c_int]),
            "orc_residual": (None, [C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_int, _dp]),
            "orc_restrict": (None, [C.c_int, C.c_int, _ip, _ip, _dp, _dp, C.c_int, _dp]),
            "orc_prolong_add": (None, [C.c_int, C.c_int, _ip, _ip, _dp, _dp, C.c_int, _dp]),
            "orc_residual_check": (C.c_double, [C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_int]),
            "orc_create": (_vp, [C.c_int]),
            "orc_destroy": (None, [_vp]),
            "orc_set_smoothing": (None, [_vp, C.c_int, C.c_int]),
            "orc_set_prolongation": (None, [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp]),
            "orc_set_mass": (None, [_vp, C.c_int, _dp]),
            "orc_galerkin": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp, _dp]),
            "orc_level_size": (C.c_int, [_vp, C.c_int]),
            "orc_level_nnz": (C.c_int, [_vp, C.c_int]),
            "orc_get_level": (None, [_vp, C.c_int, _ip, _ip, _dp]),
            "orc_coarse_solve": (None, [_vp, _dp, _dp, C.c_int]),
            "orc_vcycle": (None, [_vp, _dp, _dp, C.c_int]),
            "orc_solve": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def _csc(m):
    m = sp.csc_matrix(m).astype(np.float64)
    m.sum_duplicates()
    m.sort_indices()
    return sp.csc_matrix((m.data, m.indices.astype(np.int32), m.indptr.astype(np.int32)), shape=m.shape)


def _f(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    return np.asfortranarray(a)


def _pi(a):
    return a.ctypes.data_as(_ip)


def _pd(a):
    return a.ctypes.data_as(_dp)


def _like(out, ref):
    return out[:, 0].copy() if np.asarray(ref).ndim == 1 else out


def gauss_seidel(A, b, x, iters):
    A = _csc(A); B = _f(b); X = _f(x).copy(order="F")
    lib().orc_gauss_seidel(A.shape[0], _pi(A.indptr), _pi(A.indices), _pd(A.data), _pd(B), _pd(X), B.shape[1], int(iters))
    return _like(X, x)


def residual(A, b, x):
    A = _csc(A); B = _f(b); X = _f(x); R = np.empty_like(B, order="F")
    lib().orc_residual(A.shape[0], _pi(A.indptr), _pi(A.indices), _pd(A.data), _pd(B), _pd(X), B.shape[1], _pd(R))
    return _like(R, x)


def restrict(U, r):
    U = _csc(U); R = _f(r); out = np.empty((U.shape[1], R.shape[1]), order="F")
    lib().orc_restrict(U.shape[0], U.shape[1], _pi(U.indptr), _pi(U.indices), _pd(U.data), _pd(R), R.shape[1], _pd(out))
    return _like(out, r)


def prolong_add(U, e, x):
    U = _csc(U); E = _f(e); X = _f(x).copy(order="F")
    lib().orc_prolong_add(U.shape[0], U.shape[1], _pi(U.indptr), _pi(U.indices), _pd(U.data), _pd(E), E.shape[1], _pd(X))
    return _like(X, x)


def residual_check(A, mass, b, x, type=2):
    A = _csc(A); B = _f(b); X = _f(x)
    m = np.ascontiguousarray(mass, dtype=np.float64) if mass is not None else np.ones(A.shape[0])
    return float(lib().orc_residual_check(A.shape[0], _pi(A.indptr), _pi(A.indices), _pd(A.data), _pd(m), _pd(B), _pd(X), B.shape[1], int(type)))


class Hierarchy:
    """The reference solver state on the hot path: U, Abar, coarsest factor, M, pre/post sweeps."""

    def __init__(self, U, mass=None, pre_iters=2, post_iters=2):
        self.L = len(U)
        self._h = lib().orc_create(self.L)
        self._keep = []
        for k, u in enumerate(U):
            u = _csc(u)
            lib().orc_set_prolongation(self._h, k, u.shape[0], u.shape[1], _pi(u.indptr), _pi(u.indices), _pd(u.data))
        lib().orc_set_smoothing(self._h, int(pre_iters), int(post_iters))
        if mass is not None:
            m = np.ascontiguousarray(mass, dtype=np.float64)
            lib().orc_set_mass(self._h, m.shape[0], _pd(m))
        self.timing = {}

    def set_system(self, lhs):
        """Galerkin products + coarsest factorisation (multigrid_solver.cpp:1387-1401)."""
        a = _csc(lhs)
        t = np.zeros(2)
        lib().orc_galerkin(self._h, a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), _pd(t))
        self.timing["reduction"], self.timing["coarsest_solve"] = float(t[0]), float(t[1])

    def level_operator(self, k):
        n, nnz = lib().orc_level_size(self._h, k), lib().orc_level_nnz(self._h, k)
        cp = np.empty(n + 1, np.int32); ri = np.empty(nnz, np.int32); v = np.empty(nnz)
        lib().orc_get_level(self._h, k, _pi(cp), _pi(ri), _pd(v))
        return sp.csc_matrix((v, ri, cp), shape=(n, n))

    def coarse_solve(self, rc):
        R = _f(rc); E = np.empty_like(R, order="F")
        lib().orc_coarse_solve(self._h, _pd(R), _pd(E), R.shape[1])
        return _like(E, rc)

    def vcycle(self, b, x):
        B = _f(b); X = _f(x).copy(order="F")
        lib().orc_vcycle(self._h, _pd(B), _pd(X), B.shape[1])
        return _like(X, x)

    def solve(self, rhs, x0=None, tol=1e-4, stop_type=2, max_iter=100):
        B = _f(rhs); X = B.copy(order="F") if x0 is None else _f(x0).copy(order="F")
        conv = np.zeros(2 * max(int(max_iter), 1)); res = C.c_double()      # do-while: >= 1 cycle
        it = lib().orc_solve(self._h, _pd(B), _pd(X), B.shape[1], float(tol), int(stop_type), int(max_iter), _pd(conv), C.byref(res))
        self.timing["cycles"] = float(conv[2 * (it - 1)])
        self.timing["iterations"] = it
        self.timing["residue"] = res.value
        return _like(X, rhs), it, res.value, conv[: 2 * it].reshape(-1, 2)

    def __del__(self):
        try:
            if self._h:
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass
