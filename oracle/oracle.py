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
# the same source with -march=native (BASELINE.md 2.3 asks for the CPU baseline with and without it; the reference's own build
# has none, gravomg_bindings/setup.py:38,47).  Built on the machine that runs it (bench.py's cpu_baseline leg), never shipped.
_LIB_NATIVE = os.path.join(_HERE, "libgravomg_oracle_native.so")
_ip, _dp, _vp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_void_p
_libs = {}


def build(force: bool = False, native: bool = False) -> str:
    src = os.path.join(_HERE, "gravomg_oracle.c")
    out = _LIB_NATIVE if native else _LIB
    if force or native or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", os.path.basename(out)])
    return out


def lib(native: bool = False):
    _lib = _libs.get(native)
    if _lib is None:
        path = _LIB_NATIVE if native else _LIB
        if native or not os.path.exists(path):
            build(native=native)
        l = C.CDLL(path)
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
        _lib = _libs[native] = l
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

    def __init__(self, U, mass=None, pre_iters=2, post_iters=2, native=False):
        """native: the -march=native build of the same source (timing only; the default build is the checker)."""
        self.L = len(U)
        self._lib = lib(native)
        self._h = self._lib.orc_create(self.L)
        self._keep = []
        for k, u in enumerate(U):
            u = _csc(u)
            self._lib.orc_set_prolongation(self._h, k, u.shape[0], u.shape[1], _pi(u.indptr), _pi(u.indices), _pd(u.data))
        self._lib.orc_set_smoothing(self._h, int(pre_iters), int(post_iters))
        if mass is not None:
            m = np.ascontiguousarray(mass, dtype=np.float64)
            self._lib.orc_set_mass(self._h, m.shape[0], _pd(m))
        self.timing = {}

    def set_system(self, lhs):
        """Galerkin products + coarsest factorisation (multigrid_solver.cpp:1387-1401)."""
        a = _csc(lhs)
        t = np.zeros(2)
        self._lib.orc_galerkin(self._h, a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), _pd(t))
        self.timing["reduction"], self.timing["coarsest_solve"] = float(t[0]), float(t[1])

    def level_operator(self, k):
        n, nnz = self._lib.orc_level_size(self._h, k), self._lib.orc_level_nnz(self._h, k)
        cp = np.empty(n + 1, np.int32); ri = np.empty(nnz, np.int32); v = np.empty(nnz)
        self._lib.orc_get_level(self._h, k, _pi(cp), _pi(ri), _pd(v))
        return sp.csc_matrix((v, ri, cp), shape=(n, n))

    def coarse_solve(self, rc):
        R = _f(rc); E = np.empty_like(R, order="F")
        self._lib.orc_coarse_solve(self._h, _pd(R), _pd(E), R.shape[1])
        return _like(E, rc)

    def vcycle(self, b, x):
        B = _f(b); X = _f(x).copy(order="F")
        self._lib.orc_vcycle(self._h, _pd(B), _pd(X), B.shape[1])
        return _like(X, x)

    def solve(self, rhs, x0=None, tol=1e-4, stop_type=2, max_iter=100):
        B = _f(rhs); X = B.copy(order="F") if x0 is None else _f(x0).copy(order="F")
        conv = np.zeros(2 * max(int(max_iter), 1)); res = C.c_double()      # do-while: >= 1 cycle
        it = self._lib.orc_solve(self._h, _pd(B), _pd(X), B.shape[1], float(tol), int(stop_type), int(max_iter), _pd(conv), C.byref(res))
        self.timing["cycles"] = float(conv[2 * (it - 1)])
        self.timing["iterations"] = it
        self.timing["residue"] = res.value
        return _like(X, rhs), it, res.value, conv[: 2 * it].reshape(-1, 2)

    def __del__(self):
        try:
            if self._h:
                self._lib.orc_destroy(self._h)
                self._h = None
        except Exception:
            pass


def eigen_baseline(U, mass, lhs, rhs, cycles):
    """BASELINE.md 2.1 hook: the reference's own Eigen expressions on this path (oracle/eigen_baseline.cpp), timed on this host --
    if the host has the Eigen headers.  Returns None (with the reason in the second value) where it does not: this image ships
    no Eigen, so on the GPU boxes of this build the plain-C port above IS the CPU baseline."""
    path = os.path.join(_HERE, "libgravomg_eigen_baseline.so")
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", os.path.basename(path)], stderr=subprocess.DEVNULL)
        l = C.CDLL(path)
    except Exception as e:      # no C++ compiler, or Eigen present but the restatement does not compile against that version
        return None, f"could not be built: {type(e).__name__}"
    l.orc_eigen_available.restype = C.c_int
    if not l.orc_eigen_available():
        return None, "no <Eigen/Sparse> on this host's include path"
    l.orc_eigen_create.restype = _vp
    l.orc_eigen_create.argtypes = [C.c_int]
    l.orc_eigen_destroy.argtypes = [_vp]
    l.orc_eigen_set_prolongation.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp]
    l.orc_eigen_set_mass.argtypes = [_vp, C.c_int, _dp]
    l.orc_eigen_galerkin.argtypes = [_vp, C.c_int, _ip, _ip, _dp, _dp]
    l.orc_eigen_galerkin.restype = C.c_int
    l.orc_eigen_solve.argtypes = [_vp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp]
    l.orc_eigen_solve.restype = C.c_int
    h = l.orc_eigen_create(len(U))
    keep = []
    for k, u in enumerate(U):
        u = _csc(u); keep.append(u)
        l.orc_eigen_set_prolongation(h, k, u.shape[0], u.shape[1], _pi(u.indptr), _pi(u.indices), _pd(u.data))
    m = np.ascontiguousarray(mass, dtype=np.float64)
    l.orc_eigen_set_mass(h, m.shape[0], _pd(m))
    a = _csc(lhs)
    t = np.zeros(2)
    if l.orc_eigen_galerkin(h, a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), _pd(t)) != 0:
        l.orc_eigen_destroy(h)
        return None, "SimplicialLDLT failed on the coarsest operator"
    B = _f(rhs); X = B.copy(order="F")
    conv = np.zeros(2 * max(int(cycles), 1)); res = C.c_double()
    it = l.orc_eigen_solve(h, _pd(B), _pd(X), B.shape[1], 0.0, 2, int(cycles), _pd(conv), C.byref(res))
    l.orc_eigen_destroy(h)
    conv = conv[: 2 * it].reshape(-1, 2)
    return {"value": float(conv[-1, 0] / it), "unit": "ms per V-cycle (incl. residual check)", "cores": 1, "kind": "reference expressions (Eigen)",
            "cycles": int(it), "setup_ms": {"reduction": float(t[0]), "coarsest_solve": float(t[1])}, "residues": [float(v) for v in conv[:, 1]]}, None
