"""ctypes binding of the C-ABI in include/gravomg_hip.h (libgravomg_hip.so).

This is the thinnest possible Python view of the boundary: numpy / scipy arrays in, numpy arrays out,
every status code turned into an exception carrying gmg_last_error().  There is no CPU fallback: if the
shared library is missing the import of :func:`lib` fails loudly, and every device entry point raises
``GmgError`` (GMG_ERR_NO_DEVICE) on a box without a HIP device.

Sparse matrices cross the boundary in the reference's own storage, CSC with int32 sorted indices
(Eigen::SparseMatrix<double>; gravomg_bindings/src/cpp/core.cpp:4 and pybind11's sparse caster).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMG_LIB_PATH") or os.path.join(_HERE, "lib", "libgravomg_hip.so")      # (override: A/B runs of experimental builds)

GMG_OK, GMG_ERR_INVALID, GMG_ERR_NO_DEVICE, GMG_ERR_HIP, GMG_ERR_STATE, GMG_ERR_NUMERIC, GMG_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
DIVERGED = 1          # gmg_solve only, not an error: the iteration did not contract
SMOOTHER_MULTICOLOR_GS, SMOOTHER_JACOBI = 0, 1
COARSE_HOST_LDLT, COARSE_DEVICE_INVERSE, COARSE_AUTO = 0, 1, 2

_STATUS_NAMES = {
    -1: "GMG_ERR_INVALID", -2: "GMG_ERR_NO_DEVICE", -3: "GMG_ERR_HIP", -4: "GMG_ERR_STATE",
    -5: "GMG_ERR_NUMERIC", -6: "GMG_ERR_UNSUPPORTED",
}


class GmgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_STATUS_NAMES.get(code, code)}: {msg}")
        self.code = code


class GmgConfig(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("smoother", C.c_int), ("jacobi_omega", C.c_double), ("pre_iters", C.c_int),
        ("post_iters", C.c_int), ("coarse_mode", C.c_int), ("use_graph", C.c_int), ("sigma", C.c_int),
        ("row_align", C.c_int), ("block_rows", C.c_int), ("block_lanes", C.c_int), ("block_from_level", C.c_int),
        ("device_setup", C.c_int), ("device_rap", C.c_int), ("reorder_fine", C.c_int), ("inner_precision", C.c_int), ("block_csr", C.c_int), ("host_threads", C.c_int),
        ("verbose", C.c_int), ("gs_omega", C.c_double), ("restrict_sigma", C.c_int), ("block_ep", C.c_int), ("dist_shard_levels", C.c_int), ("block_fine", C.c_int), ("fine_col16", C.c_int), ("stream_gate", C.c_int), ("dist_exchange", C.c_int), ("prepare_structure", C.c_int), ("fuse_restrict_sweep", C.c_int), ("speculate_head", C.c_int), ("uniform_slices", C.c_int), ("color_ahead", C.c_int),
    ]


class GmgHierarchyOptions(C.Structure):
    _fields_ = [
        ("ratio", C.c_double), ("lower_bound", C.c_int), ("check_voronoi", C.c_int), ("nested", C.c_int),
        ("sampling", C.c_int), ("weighting", C.c_int), ("debug", C.c_int), ("full_clustering", C.c_int), ("use_device", C.c_int),
    ]


_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes).  Must list every symbol include/gravomg_hip.h declares
# (tests/test_host.py::test_library_exports_every_declared_symbol checks this table against the header).
SIGNATURES = {
    "gmg_config_default": (C.c_int, [C.POINTER(GmgConfig)]),
    "gmg_config_size": (C.c_int, []),
    "gmg_create": (C.c_int, [C.POINTER(GmgConfig), C.POINTER(_vp)]),
    "gmg_destroy": (None, [_vp]),
    "gmg_last_error": (C.c_char_p, [_vp]),
    "gmg_device_count": (C.c_int, []),
    "gmg_set_num_levels": (C.c_int, [_vp, C.c_int]),
    "gmg_set_prolongation": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp]),
    "gmg_set_mass": (C.c_int, [_vp, C.c_int, _dp]),
    "gmg_set_system": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp]),
    "gmg_num_levels": (C.c_int, [_vp]),
    "gmg_level_info": (C.c_int, [_vp, C.c_int, _ip, C.POINTER(C.c_int64), _ip, _ip]),
    "gmg_get_level_operator": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp]),
    "gmg_get_level_ordering": (C.c_int, [_vp, C.c_int, _ip, _ip]),
    "gmg_get_level_blocks": (C.c_int, [_vp, C.c_int, _ip, _ip, C.POINTER(C.c_ubyte)]),
    "gmg_get_timing": (C.c_int, [_vp, C.c_char_p, _dp]),
    "gmg_smooth": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, C.c_int]),
    "gmg_residual": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, _dp]),
    "gmg_spmv": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp]),
    "gmg_restrict": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp]),
    "gmg_prolong_add": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp]),
    "gmg_coarse_solve": (C.c_int, [_vp, _dp, C.c_int, _dp]),
    "gmg_residual_norm": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int, _dp]),
    "gmg_vcycle": (C.c_int, [_vp, _dp, _dp, C.c_int]),
    "gmg_smooth_residual": (C.c_int, [_vp, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp]),
    "gmg_solve": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _ip, _dp, _dp]),
    "gmg_solve_x0_rhs": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double, C.c_int, C.c_int, _ip, _dp, _dp]),
    "gmg_load_problem": (C.c_int, [_vp, _dp, _dp, C.c_int]),
    "gmg_run_cycles": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "gmg_fetch_solution": (C.c_int, [_vp, _dp]),
    "gmg_set_stream": (C.c_int, [_vp, _vp]),
    "gmg_dist_setup": (C.c_int, [_vp, C.c_int, C.c_int]),
    "gmg_dist_partition": (C.c_int, [_vp, C.c_int, C.c_int]),
    "gmg_dist_bind": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "gmg_dist_smooth_color": (C.c_int, [_vp, C.c_int]),
    "gmg_dist_residual_own": (C.c_int, [_vp]),
    "gmg_dist_coarse_cycle": (C.c_int, [_vp]),
    "gmg_dist_prolong_own": (C.c_int, [_vp]),
    "gmg_dist_norm_partial": (C.c_int, [_vp, C.c_int, _dp]),
    "gmg_dist_all_rows": (C.c_int, [_vp, C.c_int]),
    "gmg_dist_gather": (C.c_int, [_vp, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gmg_dist_scatter": (C.c_int, [_vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "gmg_p2p_blob_bytes": (C.c_int, []),
    "gmg_p2p_prepare": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "gmg_p2p_export": (C.c_int, [_vp, C.c_void_p]),
    "gmg_p2p_connect": (C.c_int, [_vp, C.c_void_p]),
    "gmg_p2p_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "gmg_p2p_connect_rccl": (C.c_int, [_vp, C.c_void_p]),
    "gmg_p2p_load": (C.c_int, [_vp, _dp, _dp]),
    "gmg_p2p_cycles": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "gmg_p2p_fetch": (C.c_int, [_vp, _dp]),
    "gmg_p2p_solve": (C.c_int, [_vp, _dp, _dp, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "gmg_p2p_stat": (C.c_int, [_vp, C.c_char_p, _dp]),
    "gmg_p2p_set_smoother": (C.c_int, [_vp, C.c_int]),
    "gmg_p2p_set_fences": (C.c_int, [_vp, C.c_int]),
    "gmg_hierarchy_options_default": (C.c_int, [C.POINTER(GmgHierarchyOptions)]),
    "gmg_hierarchy_build": (C.c_int, [_dp, C.c_int, _ip, C.c_int, C.POINTER(GmgHierarchyOptions), C.POINTER(_vp)]),
    "gmg_hierarchy_destroy": (None, [_vp]),
    "gmg_hierarchy_num_levels": (C.c_int, [_vp]),
    "gmg_hierarchy_level_shape": (C.c_int, [_vp, C.c_int, _ip, _ip, _ip]),
    "gmg_hierarchy_get_prolongation": (C.c_int, [_vp, C.c_int, _ip, _ip, _dp]),
    "gmg_hierarchy_get_timing": (C.c_int, [_vp, C.c_char_p, _dp]),
    "gmg_hierarchy_get_samples": (C.c_int, [_vp, C.c_int, _ip]),
    "gmg_hierarchy_get_nearest": (C.c_int, [_vp, C.c_int, _ip]),
    "gmg_hierarchy_get_points": (C.c_int, [_vp, C.c_int, _dp]),
    "gmg_host_threads": (C.c_int, []),
    "gmg_hierarchy_get_triangles": (C.c_int, [_vp, C.c_int, _ip, _ip]),
    "gmg_hierarchy_get_fine_order": (C.c_int, [_vp, _ip, _ip]),
    "gmg_set_fine_order": (C.c_int, [_vp, C.c_int, _ip]),
    "gmg_set_fine_graph": (C.c_int, [_vp, C.c_int, C.c_int, _ip]),
    "gmg_use_hierarchy": (C.c_int, [_vp, _vp]),
    "gmg_finalize_hierarchy": (C.c_int, [_vp]),
    "gmg_host_ldlt_solve": (C.c_int, [C.c_int, _ip, _ip, _dp, _dp, C.c_int, _dp, C.POINTER(C.c_int64)]),
}

# test / measurement hooks: include/gravomg_hip_internal.h (not part of the drop-in boundary)
INTERNAL_SIGNATURES = {
    "gmg_debug_sell_info": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "gmg_debug_sell_copy": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_int64), _ip, _dp, _ip, _dp]),
    "gmg_host_galerkin": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, _ip, _ip, _dp, _ip, _ip, _dp]),
    "gmg_host_plan_level": (C.c_int, [C.c_int, _ip, _ip, _dp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), _ip, _ip, _ip,
                                      C.POINTER(C.c_ubyte)]),
    "gmg_host_fine_block_rule": (C.c_int, [C.c_int, _ip, _ip, _dp, _ip, _ip]),
    "gmg_debug_set": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "gmg_bench_kernel": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _ip]),
    "gmg_profile_cycle": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int]),
    "gmg_algorithmic_bytes": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp]),
    "gmg_p2p_bench_kind": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp]),
    "gmg_p2p_debug_collective_roundtrip": (C.c_int, [_vp, _dp, C.POINTER(C.c_longlong)]),
    "gmg_host_ldlt_probe": (C.c_int, [C.c_int, _ip, _ip, _dp, _dp, C.c_int, C.c_char_p, C.c_int]),
}

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1 with the same SONAMEs as
    /opt/rocm.  Whichever copy is loaded first serves the whole process, and torch reports "No HIP GPUs are
    available" if the system copy got in first.  bench.py, dist.py and the tests use torch in the same process for
    streams and torch.distributed, so make torch's copy the process-wide one -- without importing torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def lib() -> C.CDLL:
    """Load libgravomg_hip.so (built in-tree by gravo_mg_amd/csrc/build.sh).  Fails loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with gravo_mg_amd/csrc/build.sh (or __graft_entry__.build()). "
                "gravo_mg_amd has no CPU fallback for the device path.")
        _preload_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(INTERNAL_SIGNATURES.items()):
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.gmg_config_size() != C.sizeof(GmgConfig):
            raise ImportError(f"{LIB_PATH} was built with a gmg_config of {l.gmg_config_size()} bytes, gravo_mg_amd/cabi.py mirrors one of {C.sizeof(GmgConfig)}: "
                              "rebuild the library (gravo_mg_amd/csrc/build.sh) or update the mirror")
        _lib = l
    return _lib


def device_count() -> int:
    return int(lib().gmg_device_count())


def _f64(a, shape2d: bool = True) -> np.ndarray:
    """Column-major (Fortran) float64 n x d view/copy of a vector or matrix."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    return np.asfortranarray(a)


def _csc_layout(m) -> sp.csc_matrix:
    """CSC with int32 indices and float64 values, without the (single-threaded) canonical-format passes of _csc."""
    m = m if sp.isspmatrix_csc(m) else sp.csc_matrix(m)
    if m.indptr.dtype != np.int32 or m.indices.dtype != np.int32:
        if m.nnz >= 2**31:
            raise ValueError("matrix too large for int32 indices")
        m = sp.csc_matrix((m.data, m.indices.astype(np.int32), m.indptr.astype(np.int32)), shape=m.shape)
    if m.data.dtype != np.float64:
        m = m.astype(np.float64)
    return m


def _csc(m) -> sp.csc_matrix:
    m = sp.csc_matrix(m)
    if not m.has_sorted_indices:
        m = m.copy()
        m.sort_indices()
    m.sum_duplicates()
    if m.indptr.dtype != np.int32 or m.indices.dtype != np.int32:
        if m.nnz >= 2**31:
            raise ValueError("matrix too large for int32 indices")
        m = sp.csc_matrix((m.data, m.indices.astype(np.int32), m.indptr.astype(np.int32)), shape=m.shape)
    if m.data.dtype != np.float64:
        m = m.astype(np.float64)
    return m


def _pi(a: np.ndarray):
    return a.ctypes.data_as(_ip)


def _pd(a: np.ndarray):
    return a.ctypes.data_as(_dp)


class Hierarchy:
    """Graph-Voronoi prolongation hierarchy built on the host (gmg_hierarchy_build).

    Mirrors what ``MGBS::MultigridSolver::buildHierarchy`` produces: ``U`` (list of scipy CSC matrices,
    n_k x n_{k+1}) and the reference's ``hierarchyTiming`` keys."""

    def __init__(self, pos, neigh, ratio=8.0, lower_bound=1000, check_voronoi=True, nested=False, sampling=0, weighting=0, debug=False, full_clustering=False, use_device=True):
        l = lib()
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        if pos.ndim != 2 or pos.shape[1] != 3 or neigh.ndim != 2 or neigh.shape[0] != pos.shape[0]:
            raise ValueError("pos must be n x 3 and neigh n x K")
        opt = GmgHierarchyOptions()
        l.gmg_hierarchy_options_default(C.byref(opt))
        opt.ratio, opt.lower_bound, opt.check_voronoi, opt.nested = float(ratio), int(lower_bound), int(bool(check_voronoi)), int(bool(nested))
        opt.sampling, opt.weighting, opt.debug, opt.full_clustering = int(sampling), int(weighting), int(bool(debug)), int(bool(full_clustering))
        opt.use_device = int(bool(use_device))
        self._h = _vp()
        rc = l.gmg_hierarchy_build(_pd(pos), pos.shape[0], _pi(neigh), neigh.shape[1], C.byref(opt), C.byref(self._h))
        if rc:
            raise GmgError(rc, "gmg_hierarchy_build failed (only Sampling.FASTDISK is supported)" if rc == GMG_ERR_UNSUPPORTED else "gmg_hierarchy_build failed")
        self.U: List[sp.csc_matrix] = []
        for k in range(l.gmg_hierarchy_num_levels(self._h)):
            nf, nc, nnz = C.c_int(), C.c_int(), C.c_int()
            l.gmg_hierarchy_level_shape(self._h, k, C.byref(nf), C.byref(nc), C.byref(nnz))
            colptr = np.empty(nc.value + 1, np.int32); rowidx = np.empty(nnz.value, np.int32); val = np.empty(nnz.value, np.float64)
            l.gmg_hierarchy_get_prolongation(self._h, k, _pi(colptr), _pi(rowidx), _pd(val))
            self.U.append(sp.csc_matrix((val, rowidx, colptr), shape=(nf.value, nc.value)))
        # what the reference keeps beside U (multigrid_solver.h:99-104): sample indices, cluster of every point, coarse positions
        self.samples, self.nearest, self.points = [], [], []
        for k, u in enumerate(self.U):
            s_ = np.empty(u.shape[1], np.int32); n_ = np.empty(u.shape[0], np.int32); p_ = np.empty((u.shape[1], 3), np.float64)
            if l.gmg_hierarchy_get_samples(self._h, k, _pi(s_)) or l.gmg_hierarchy_get_nearest(self._h, k, _pi(n_)) or l.gmg_hierarchy_get_points(self._h, k, _pd(p_)):
                raise GmgError(GMG_ERR_INVALID, "hierarchy getters failed")
            self.samples.append(s_); self.nearest.append(n_); self.points.append(p_)
        # the reference's debug dump of the triangle search (allTriangles): only kept with debug=True
        self.triangles = []
        for k in range(len(self.U)):
            cnt = C.c_int()
            l.gmg_hierarchy_get_triangles(self._h, k, None, C.byref(cnt))
            t = np.empty((cnt.value, 3), np.int32)
            if cnt.value:
                l.gmg_hierarchy_get_triangles(self._h, k, _pi(t), C.byref(cnt))
            self.triangles.append(t)
        # breadth-first order of the points (only made for inputs without locality; None otherwise)
        cnt = C.c_int()
        l.gmg_hierarchy_get_fine_order(self._h, None, C.byref(cnt))
        self.fine_order = None
        if cnt.value:
            self.fine_order = np.empty(cnt.value, np.int32)
            l.gmg_hierarchy_get_fine_order(self._h, _pi(self.fine_order), C.byref(cnt))

    def timing(self, key: str) -> float:
        out = C.c_double()
        rc = lib().gmg_hierarchy_get_timing(self._h, key.encode(), C.byref(out))
        if rc:
            raise KeyError(key)
        return out.value

    TIMING_KEYS = ("n_vertices", "hierarchy", "sampling", "cluster", "next_neighborhood", "next_positions",
                   "triangle_finding", "triangle_selection", "PDS", "levels")

    def timings(self) -> dict:
        return {k: self.timing(k) for k in self.TIMING_KEYS}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().gmg_hierarchy_destroy(self._h)
                self._h = None
        except Exception:
            pass


class Engine:
    """One gmg_handle: device-resident hierarchy + V-cycle on one HIP stream."""

    def __init__(self, smoother=SMOOTHER_MULTICOLOR_GS, pre_iters=2, post_iters=2, jacobi_omega=0.67,
                 coarse_mode=COARSE_AUTO, use_graph=False, sigma=0, row_align=64, block_rows=64, block_from_level=1, block_lanes=0,
                 device_setup=True, device_rap=True, reorder_fine=2, inner_precision=0, block_csr=True, device=0, verbose=False, gs_omega=None, block_ep=None, restrict_sigma=None, dist_shard_levels=None, block_fine=None, fine_col16=None, stream_gate=None, prepare_structure=None, dist_exchange=None, fuse_restrict_sweep=None, speculate_head=None, uniform_slices=None, color_ahead=None):
        l = lib()
        cfg = GmgConfig()
        l.gmg_config_default(C.byref(cfg))
        cfg.device, cfg.smoother, cfg.jacobi_omega = int(device), int(smoother), float(jacobi_omega)
        cfg.pre_iters, cfg.post_iters, cfg.coarse_mode = int(pre_iters), int(post_iters), int(coarse_mode)
        cfg.use_graph, cfg.sigma, cfg.row_align, cfg.verbose = int(bool(use_graph)), int(sigma), int(row_align), int(bool(verbose))
        cfg.block_rows, cfg.block_from_level, cfg.block_lanes = int(block_rows), int(block_from_level), int(block_lanes)
        cfg.device_setup, cfg.device_rap, cfg.reorder_fine = int(bool(device_setup)), int(bool(device_rap)), int(reorder_fine)
        cfg.inner_precision, cfg.block_csr = int(inner_precision), int(bool(block_csr))
        if gs_omega is not None:
            cfg.gs_omega = float(gs_omega)
        if block_ep is not None:
            cfg.block_ep = int(bool(block_ep))
        if restrict_sigma is not None:
            cfg.restrict_sigma = int(restrict_sigma)
        if dist_shard_levels is not None:
            cfg.dist_shard_levels = int(dist_shard_levels)
        if block_fine is not None:
            cfg.block_fine = int(bool(block_fine))
        if fine_col16 is not None:
            cfg.fine_col16 = int(bool(fine_col16))
        if stream_gate is not None:
            cfg.stream_gate = int(bool(stream_gate))
        if prepare_structure is not None:
            cfg.prepare_structure = int(bool(prepare_structure))
        if fuse_restrict_sweep is not None:
            cfg.fuse_restrict_sweep = int(bool(fuse_restrict_sweep))
        if speculate_head is not None:
            cfg.speculate_head = int(bool(speculate_head))
        if uniform_slices is not None:
            cfg.uniform_slices = int(bool(uniform_slices))
        if color_ahead is not None:
            cfg.color_ahead = int(bool(color_ahead))
        if dist_exchange is not None:
            cfg.dist_exchange = int(dist_exchange)
        self._h = _vp()
        rc = l.gmg_create(C.byref(cfg), C.byref(self._h))
        if rc:
            raise GmgError(rc, "gmg_create failed (invalid configuration)")
        self._n0 = None
        self._sizes: List[int] = []
        self.pre_iters, self.post_iters = int(pre_iters), int(post_iters)
        self.gs_omega = float(cfg.gs_omega)       # relaxation factor of the level-0 sweep (engine default unless given)

    @classmethod
    def borrow(cls, handle: int) -> "Engine":
        """View of a gmg_handle somebody else owns (e.g. the one inside gravomg.MultigridSolver, prepare_system()): never destroyed
        from here."""
        self = cls.__new__(cls)
        self._h = _vp(int(handle))
        self._borrowed = True
        self._n0 = None
        self._sizes = []
        self.pre_iters = self.post_iters = None
        self.gs_omega = None
        return self

    # -- plumbing
    def _chk(self, rc: int):
        if rc:
            raise GmgError(rc, lib().gmg_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                lib().gmg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hierarchy input
    def set_prolongations(self, U: Sequence, finalize: bool = True, fine_order=None, fine_graph=None):
        """fine_order: optional locality order of the level-0 points (gmg_set_fine_order), e.g. Hierarchy.fine_order.
        fine_graph: optional n x K neighbour table the hierarchy was built from (gmg_set_fine_graph): the engine then prepares the structure of
        the systems to come when the hierarchy is finalized."""
        l = lib()
        self._chk(l.gmg_set_num_levels(self._h, len(U)))
        self._sizes = []
        for k, u in enumerate(U):
            u = _csc(u)
            self._chk(l.gmg_set_prolongation(self._h, k, u.shape[0], u.shape[1], _pi(u.indptr), _pi(u.indices), _pd(u.data)))
            if k == 0:
                self._sizes.append(u.shape[0])
            self._sizes.append(u.shape[1])
        if fine_order is not None and len(U):
            fo = np.ascontiguousarray(fine_order, dtype=np.int32)
            self._chk(l.gmg_set_fine_order(self._h, fo.shape[0], _pi(fo)))
        if fine_graph is not None and len(U):
            fg = np.ascontiguousarray(fine_graph, dtype=np.int32)
            self._chk(l.gmg_set_fine_graph(self._h, fg.shape[0], fg.shape[1], _pi(fg)))
        if finalize and len(U):
            self._chk(l.gmg_finalize_hierarchy(self._h))

    def use_hierarchy(self, hier: Hierarchy):
        self._chk(lib().gmg_use_hierarchy(self._h, hier._h))
        self._sizes = [hier.U[0].shape[0]] + [u.shape[1] for u in hier.U] if hier.U else []

    def set_mass(self, mass_diag):
        m = np.ascontiguousarray(mass_diag, dtype=np.float64).ravel()
        self._chk(lib().gmg_set_mass(self._h, m.shape[0], _pd(m)))

    def set_system(self, lhs):
        a = _csc_layout(lhs)        # the engine checks (threaded) and canonicalises the storage itself
        if a.shape[0] != a.shape[1]:
            raise ValueError("lhs must be square")
        self._chk(lib().gmg_set_system(self._h, a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data)))
        self._n0 = a.shape[0]

    # -- introspection
    @property
    def num_levels(self) -> int:
        return int(lib().gmg_num_levels(self._h))

    def level_info(self, k: int) -> dict:
        n, nnz, nc, npad = C.c_int(), C.c_int64(), C.c_int(), C.c_int()
        self._chk(lib().gmg_level_info(self._h, k, C.byref(n), C.byref(nnz), C.byref(nc), C.byref(npad)))
        return {"n": n.value, "nnz": nnz.value, "n_colors": nc.value, "n_pad": npad.value}

    def level_operator(self, k: int) -> sp.csc_matrix:
        info = self.level_info(k)
        colptr = np.empty(info["n"] + 1, np.int32); rowidx = np.empty(info["nnz"], np.int32); val = np.empty(info["nnz"], np.float64)
        self._chk(lib().gmg_get_level_operator(self._h, k, _pi(colptr), _pi(rowidx), _pd(val)))
        return sp.csc_matrix((val, rowidx, colptr), shape=(info["n"], info["n"]))

    def level_ordering(self, k: int) -> Tuple[np.ndarray, np.ndarray]:
        info = self.level_info(k)
        new2old = np.empty(info["n_pad"], np.int32); cb = np.empty(info["n_colors"] + 1, np.int32)
        self._chk(lib().gmg_get_level_ordering(self._h, k, _pi(new2old), _pi(cb)))
        return new2old, cb

    def level_blocks(self, k: int):
        """(blk_begin[n_blocks+1], row_color[n_pad]) of a blocked level, or None for a colour-major level."""
        nb = C.c_int()
        self._chk(lib().gmg_get_level_blocks(self._h, k, C.byref(nb), None, None))
        if nb.value == 0:
            return None
        bb = np.empty(nb.value + 1, np.int32); rc = np.empty(self.level_info(k)["n_pad"], np.uint8)
        self._chk(lib().gmg_get_level_blocks(self._h, k, C.byref(nb), _pi(bb), rc.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return bb, rc

    def debug_sell(self, k: int, which: int) -> dict:
        """Device-resident SELL layout (0 A, 1 A_in, 2 A_out, 3 P, 4 R) copied back, for layout parity tests.  5 = the
        block-CSR of a big blocked level: slice_ptr holds the row pointers, row_of the per-row start of the in-block entries."""
        info = (C.c_int64 * 4)()
        self._chk(lib().gmg_debug_sell_info(self._h, int(k), int(which), info))
        ns, lpr, stored, has_row_of = (int(v) for v in info)
        sp_ = np.zeros(ns + 1, np.int64); col = np.zeros(stored, np.int32); val = np.zeros(stored)
        row_of = np.zeros(ns * (64 // lpr), np.int32) if has_row_of else None
        diag = np.zeros(self.level_info(k)["n_pad"]) if which == 0 else None
        self._chk(lib().gmg_debug_sell_copy(self._h, int(k), int(which), sp_.ctypes.data_as(C.POINTER(C.c_int64)), _pi(col), _pd(val),
                                            _pi(row_of) if has_row_of else None, _pd(diag) if diag is not None else None))
        return {"n_slices": ns, "lpr": lpr, "slice_ptr": sp_, "col": col, "val": val, "row_of": row_of, "diag": diag}

    def debug_set(self, key: str, value: float):
        """Set-up fault injection for the tests (gravomg_hip_internal.h), effective from the next set_system."""
        self._chk(lib().gmg_debug_set(self._h, key.encode(), float(value)))

    def timing(self, key: str) -> float:
        out = C.c_double()
        self._chk(lib().gmg_get_timing(self._h, key.encode(), C.byref(out)))
        return out.value

    # -- operators
    def _level_n(self, k: int) -> int:
        return self.level_info(k)["n"]

    @staticmethod
    def _shape_like(out: np.ndarray, ref) -> np.ndarray:
        return out[:, 0].copy() if np.asarray(ref).ndim == 1 else out

    def smooth(self, k, b, x, iters):
        B, X = _f64(b), _f64(x).copy(order="F")
        self._chk(lib().gmg_smooth(self._h, k, _pd(B), _pd(X), B.shape[1], int(iters)))
        return self._shape_like(X, x)

    def smooth_residual(self, k, b, x, iters, from_zero=False):
        """(x after `iters` sweeps, b - A x as the way down forms it) -- gmg_smooth_residual"""
        B = _f64(b)
        X = np.zeros_like(B, order="F") if x is None else _f64(x).copy(order="F")
        R = np.empty_like(B, order="F")
        self._chk(lib().gmg_smooth_residual(self._h, k, _pd(B), _pd(X), B.shape[1], int(iters), int(bool(from_zero)), _pd(R)))
        return self._shape_like(X, b), self._shape_like(R, b)

    def residual(self, k, b, x):
        B, X = _f64(b), _f64(x)
        R = np.empty_like(B, order="F")
        self._chk(lib().gmg_residual(self._h, k, _pd(B), _pd(X), B.shape[1], _pd(R)))
        return self._shape_like(R, x)

    def spmv(self, k, x):
        X = _f64(x)
        Y = np.empty_like(X, order="F")
        self._chk(lib().gmg_spmv(self._h, k, _pd(X), X.shape[1], _pd(Y)))
        return self._shape_like(Y, x)

    def restrict(self, k, r):
        R = _f64(r)
        RC = np.empty((self._level_n(k + 1), R.shape[1]), order="F")
        self._chk(lib().gmg_restrict(self._h, k, _pd(R), R.shape[1], _pd(RC)))
        return self._shape_like(RC, r)

    def prolong_add(self, k, e, x):
        E, X = _f64(e), _f64(x).copy(order="F")
        self._chk(lib().gmg_prolong_add(self._h, k, _pd(E), E.shape[1], _pd(X)))
        return self._shape_like(X, x)

    def coarse_solve(self, rc):
        R = _f64(rc)
        E = np.empty_like(R, order="F")
        self._chk(lib().gmg_coarse_solve(self._h, _pd(R), R.shape[1], _pd(E)))
        return self._shape_like(E, rc)

    def residual_norm(self, b, x, type=2) -> float:
        B, X = _f64(b), _f64(x)
        out = C.c_double()
        self._chk(lib().gmg_residual_norm(self._h, _pd(B), _pd(X), B.shape[1], int(type), C.byref(out)))
        return out.value

    # -- hot path
    def vcycle(self, b, x):
        B, X = _f64(b), _f64(x).copy(order="F")
        self._chk(lib().gmg_vcycle(self._h, _pd(B), _pd(X), B.shape[1]))
        return self._shape_like(X, x)

    def solve(self, rhs, x0=None, tol=1e-4, stop_type=2, max_iter=100, out=None):
        """Returns (x, iterations, residue, convergence[(ms, residue), ...]).  x0 defaults to rhs
        (gravomg_bindings/src/cpp/core.cpp:69).  The C-ABI takes column-major n x d blocks (Eigen::MatrixXd): a C-ordered (n, d) rhs with
        d > 1 is converted first -- pass np.asfortranarray(rhs) to keep that copy out of a timed call.  out: an (n, d) column-major float64
        array that receives x (with x0 = None; a caller that solves repeatedly spares the allocation and first touch of a fresh result:
        ~4 ms for 72 MB)."""
        B = _f64(rhs)
        iters, res = C.c_int(), C.c_double()
        conv = np.zeros(2 * max(int(max_iter), 1))
        if x0 is None:                       # gmg_solve_x0_rhs: x is output only (no copy of rhs made here, none uploaded)
            if out is not None:
                if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == B.shape and out.flags.f_contiguous):
                    raise ValueError("out must be a column-major float64 array of the shape of rhs")
                X = out
            else:
                X = np.empty(B.shape, order="F")
            rc = lib().gmg_solve_x0_rhs(self._h, _pd(B), _pd(X), B.shape[1], float(tol), int(stop_type), int(max_iter),
                                        C.byref(iters), C.byref(res), _pd(conv))
        else:
            X = _f64(x0).copy(order="F")
            rc = lib().gmg_solve(self._h, _pd(B), _pd(X), B.shape[1], float(tol), int(stop_type), int(max_iter),
                                 C.byref(iters), C.byref(res), _pd(conv))
        self.diverged = rc == DIVERGED       # not an error: the iteration did not contract, X holds the last iterate (include/gravomg_hip.h)
        if not self.diverged:
            self._chk(rc)
        return self._shape_like(X, rhs), iters.value, res.value, conv[: 2 * iters.value].reshape(-1, 2)

    def load_problem(self, b, x0):
        B, X = _f64(b), _f64(x0)
        self._chk(lib().gmg_load_problem(self._h, _pd(B), _pd(X), B.shape[1]))
        self._loaded_shape = (B.shape[0], B.shape[1])

    def run_cycles(self, n_cycles: int, stop_type: int = 2) -> np.ndarray:
        res = np.zeros(max(int(n_cycles), 1))
        self._chk(lib().gmg_run_cycles(self._h, int(n_cycles), int(stop_type), _pd(res)))
        return res[: int(n_cycles)]

    def fetch_solution(self) -> np.ndarray:
        X = np.empty(self._loaded_shape, order="F")
        self._chk(lib().gmg_fetch_solution(self._h, _pd(X)))
        return X

    # -- measurement
    def bench_kernel(self, kind: int, k: int, d: int, reps: int) -> Tuple[float, int]:
        ms, launches = C.c_double(), C.c_int()
        self._chk(lib().gmg_bench_kernel(self._h, int(kind), int(k), int(d), int(reps), C.byref(ms), C.byref(launches)))
        return ms.value, launches.value

    def profile_cycle(self, stop_type: int = 2, reps: int = 10) -> np.ndarray:
        """ms per leg of a V-cycle + residual check on the resident problem: levels 0 .. L-1, the coarsest solve, the check."""
        out = np.zeros(self.num_levels + 2)
        self._chk(lib().gmg_profile_cycle(self._h, int(stop_type), int(reps), _pd(out), out.size))
        return out

    def algorithmic_bytes(self, kind: int, k: int, d: int) -> float:
        out = C.c_double()
        self._chk(lib().gmg_algorithmic_bytes(self._h, int(kind), int(k), int(d), C.byref(out)))
        return out.value

    # -- multi-GPU steps (device pointers / stream handles are plain integers)
    def set_stream(self, stream_handle: int):
        self._chk(lib().gmg_set_stream(self._h, _vp(stream_handle) if stream_handle else None))

    def dist_partition(self, rank: int, world: int):
        """Before use_hierarchy / set_system: lay out and keep only rank `rank`'s rows of levels 0-1 (gmg_dist_partition)."""
        self._chk(lib().gmg_dist_partition(self._h, int(rank), int(world)))

    def dist_setup(self, rank: int, world: int):
        self._chk(lib().gmg_dist_setup(self._h, int(rank), int(world)))

    def dist_bind(self, x_ptr: int, b_ptr: int, r_ptr: int, d: int):
        self._chk(lib().gmg_dist_bind(self._h, _vp(x_ptr), _vp(b_ptr), _vp(r_ptr), int(d)))
        self._loaded_shape = (self.level_info(0)["n"], int(d))

    def dist_smooth_color(self, c: int):
        self._chk(lib().gmg_dist_smooth_color(self._h, int(c)))

    def dist_residual_own(self):
        self._chk(lib().gmg_dist_residual_own(self._h))

    def dist_coarse_cycle(self):
        self._chk(lib().gmg_dist_coarse_cycle(self._h))

    def dist_prolong_own(self):
        self._chk(lib().gmg_dist_prolong_own(self._h))

    def dist_norm_partial(self, type: int, d: int) -> np.ndarray:
        sums = np.zeros(2 * d)
        self._chk(lib().gmg_dist_norm_partial(self._h, int(type), _pd(sums)))
        return sums

    def dist_all_rows(self, on: bool):
        """dist_residual_own / dist_prolong_own / dist_norm_partial cover ALL rows of level 0 while on (gmg_dist_all_rows)."""
        self._chk(lib().gmg_dist_all_rows(self._h, int(bool(on))))

    def dist_residual_all(self):
        self.dist_all_rows(True)
        try:
            self.dist_residual_own()
        finally:
            self.dist_all_rows(False)

    def dist_prolong_all(self):
        self.dist_all_rows(True)
        try:
            self.dist_prolong_own()
        finally:
            self.dist_all_rows(False)

    def dist_norm_all(self, type: int, d: int) -> np.ndarray:
        self.dist_all_rows(True)
        try:
            return self.dist_norm_partial(type, d)
        finally:
            self.dist_all_rows(False)

    def dist_gather(self, src_ptr: int, idx_ptr: int, n: int, dst_ptr: int):
        """dst[i] = src[idx[i]] on the engine stream (device pointers)."""
        self._chk(lib().gmg_dist_gather(self._h, src_ptr, idx_ptr, int(n), dst_ptr))

    def dist_scatter(self, src_ptr: int, pos_ptr: int, idx_ptr: int, n: int, dst_ptr: int):
        """dst[idx[i]] = src[pos[i]] on the engine stream (device pointers)."""
        self._chk(lib().gmg_dist_scatter(self._h, src_ptr, pos_ptr, idx_ptr, int(n), dst_ptr))


class P2PCycle:
    """Engine-driven multi-GPU V-cycle (include/gravomg_hip.h "multi-GPU, engine-driven"): one instance per rank, each over an
    Engine created with row_align = 64 * world and the system set.  `connect` takes the blobs of all ranks in rank order
    (gathered by the caller, e.g. with torch.distributed.all_gather_object).  Ranks are separate processes."""

    def __init__(self, engine: "Engine", rank: int, world: int, d: int = 1):
        self.eng, self.rank, self.world, self.d = engine, int(rank), int(world), int(d)
        engine._chk(lib().gmg_p2p_prepare(engine._h, self.rank, self.world, self.d))
        self._n = engine.level_info(0)["n"]

    def export(self) -> bytes:
        buf = C.create_string_buffer(lib().gmg_p2p_blob_bytes())
        self.eng._chk(lib().gmg_p2p_export(self.eng._h, buf))
        return buf.raw

    def connect(self, blobs):
        blob = b"".join(blobs)
        assert len(blob) == self.world * lib().gmg_p2p_blob_bytes()
        self.eng._chk(lib().gmg_p2p_connect(self.eng._h, blob))

    def connect_rccl(self, unique_id: bytes):
        """gmg_config::dist_exchange = 1: join the RCCL communicator made from rank 0's id (rccl_unique_id()); collective."""
        assert len(unique_id) == 128
        self.eng._chk(lib().gmg_p2p_connect_rccl(self.eng._h, unique_id))

    def load(self, b, x0):
        B, X = _f64(b), _f64(x0)
        assert B.shape == (self._n, self.d) and X.shape == B.shape
        self.eng._chk(lib().gmg_p2p_load(self.eng._h, _pd(B), _pd(X)))

    def cycles(self, n: int, stop_type: int = 2) -> np.ndarray:
        res = np.zeros(max(int(n), 1))
        self.eng._chk(lib().gmg_p2p_cycles(self.eng._h, int(n), int(stop_type), _pd(res)))
        return res[: int(n)]

    def fetch(self) -> np.ndarray:
        X = np.empty((self._n, self.d), order="F")
        self.eng._chk(lib().gmg_p2p_fetch(self.eng._h, _pd(X)))
        return X

    def solve(self, b, x0, tol=1e-4, stop_type=2, max_iter=100):
        """gmg_p2p_solve: (x, iterations, residue); collective."""
        B = _f64(b)
        X = np.array(_f64(x0), order="F", copy=True)
        assert B.shape == (self._n, self.d) and X.shape == B.shape
        it, res = C.c_int(), C.c_double()
        rc = lib().gmg_p2p_solve(self.eng._h, _pd(B), _pd(X), float(tol), int(stop_type), int(max_iter), C.byref(it), C.byref(res))
        self.diverged = rc == DIVERGED       # not an error: the iteration did not contract on any rank (same residues everywhere), X = last iterate
        if not self.diverged:
            self.eng._chk(rc)
        return X, it.value, res.value

    def bench_exchange(self, reps: int = 100) -> float:
        return self.bench_kind("color0", reps)

    def bench_kind(self, kind: str, reps: int = 100) -> float:
        """ms per exchange of the named kind ("color<k>", "halo_all", "rows0", "x1_halo", "rows1", "r0_halo"); collective;
        overwrites halo entries (load() afterwards)."""
        out = C.c_double()
        self.eng._chk(lib().gmg_p2p_bench_kind(self.eng._h, kind.encode(), int(reps), C.byref(out)))
        return out.value

    def set_smoother(self, hybrid):
        """0 / False: exact multicolour GS, an exchange launch per colour (default); 1 / True: hybrid GS (GS inside a rank, Jacobi across ranks), one
        exchange per sweep; 2: exact, the exchange of a colour folded into that colour's sweep launch (mailbox backend: no exchange launch)."""
        self.eng._chk(lib().gmg_p2p_set_smoother(self.eng._h, int(hybrid)))

    def set_fences(self, fenced: bool):
        """True (default): system-scope release / acquire fences around the sequence words of the mailbox exchanges; False: the gfx942 / gfx950
        form without the cache write-back (gmg_p2p_set_fences) -- check the first cycles against a trusted run when you take it."""
        self.eng._chk(lib().gmg_p2p_set_fences(self.eng._h, int(bool(fenced))))

    def stat(self, key: str) -> float:
        out = C.c_double()
        self.eng._chk(lib().gmg_p2p_stat(self.eng._h, key.encode(), C.byref(out)))
        return out.value

    def collective_roundtrip(self):
        """(max |sent - gathered|, values compared) of one rows-of-x exchange through the collective backend (internal test hook)."""
        diff, cnt = C.c_double(), C.c_longlong()
        self.eng._chk(lib().gmg_p2p_debug_collective_roundtrip(self.eng._h, C.byref(diff), C.byref(cnt)))
        return diff.value, cnt.value


def rccl_unique_id() -> bytes:
    """128-byte id of a new RCCL communicator (rank 0 makes it, every rank gets it: P2PCycle.connect_rccl)."""
    buf = C.create_string_buffer(128)
    rc = lib().gmg_p2p_rccl_unique_id(buf)
    if rc:
        raise GmgError(rc, "gmg_p2p_rccl_unique_id (librccl not loadable?)")
    return buf.raw


def host_galerkin(A, U) -> sp.csc_matrix:
    """Ac = U^T A U on the host (the product engine's RAP, exposed for parity tests)."""
    a, u = _csc(A), _csc(U)
    nc = u.shape[1]
    colptr = np.zeros(nc + 1, np.int32)
    rc = lib().gmg_host_galerkin(a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), nc, _pi(u.indptr), _pi(u.indices), _pd(u.data),
                                 _pi(colptr), None, None)
    if rc:
        raise GmgError(rc, "gmg_host_galerkin")
    nnz = int(colptr[nc])
    rowidx = np.empty(nnz, np.int32); val = np.empty(nnz, np.float64)
    lib().gmg_host_galerkin(a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), nc, _pi(u.indptr), _pi(u.indices), _pd(u.data),
                            _pi(colptr), _pi(rowidx), _pd(val))
    return sp.csc_matrix((val, rowidx, colptr), shape=(nc, nc))


def default_host_threads() -> int:
    """Host threads a handle uses by default: the CPUs this process may use (affinity, cgroup quota) divided by LOCAL_WORLD_SIZE."""
    return int(lib().gmg_host_threads())


def host_plan_level(A, mode: int = 0, block_rows: int = 256, sigma: int = 1024, row_align: int = 64) -> dict:
    """Device layout of one level computed on the host (no GPU needed): ordering, colours / blocks, SELL
    padding statistics.  mode 0 = colour-major (exact multicolour GS; colour classes padded to `row_align` rows),
    1 = block ordering (block-hybrid GS, `block_rows` rows per block)."""
    a = _csc(A)
    n = a.shape[0]
    if mode != 1:
        block_rows = row_align
    info = (C.c_int64 * 6)()
    # sizes first (null outputs), then the arrays: the padded length and the block count depend on the graph
    rc = lib().gmg_host_plan_level(n, _pi(a.indptr), _pi(a.indices), _pd(a.data), int(mode), int(block_rows), int(sigma), info, None, None, None, None)
    if rc:
        raise GmgError(rc, "gmg_host_plan_level")
    new2old = np.empty(int(info[0]), np.int32); color_begin = np.zeros(int(info[1]) + 1, np.int32)
    blk_begin = np.zeros(int(info[2]) + 1, np.int32); row_color = np.zeros(int(info[0]), np.uint8)
    rc = lib().gmg_host_plan_level(n, _pi(a.indptr), _pi(a.indices), _pd(a.data), int(mode), int(block_rows), int(sigma), info,
                                   _pi(new2old), _pi(color_begin), _pi(blk_begin), row_color.ctypes.data_as(C.POINTER(C.c_ubyte)))
    if rc:
        raise GmgError(rc, "gmg_host_plan_level")
    n_pad, n_colors, n_blocks = int(info[0]), int(info[1]), int(info[2])
    out = {"n_pad": n_pad, "n_colors": n_colors, "n_blocks": n_blocks, "sell_stored": int(info[3]), "offdiag_nnz": int(info[4]),
           "new2old": new2old[:n_pad].copy()}
    if mode == 1:
        out["blk_begin"] = blk_begin[: n_blocks + 1].copy()
        out["row_color"] = row_color[:n_pad].copy()
    else:
        out["color_begin"] = color_begin[: n_colors + 1].copy()
    return out


def host_plan_ahead(n, indptr, indices, row_align: int = 64, sigma: int = 1024) -> dict:
    """gmg_host_plan_level mode 4: the colour-major ordering of a level from the colouring that a cold gmg_set_system starts AHEAD of its inspection,
    on the arrays exactly as given (int32, no canonicalisation here: that is the point).  Raises GmgError(INVALID) for arrays that would take a
    reader out of bounds.  `colored_ahead`: False when 64 colours were not enough and the general loop coloured instead."""
    indptr = np.ascontiguousarray(indptr, np.int32); indices = np.ascontiguousarray(indices, np.int32)
    vals = np.ones(max(len(indices), 1))
    info = (C.c_int64 * 6)()
    rc = lib().gmg_host_plan_level(int(n), _pi(indptr), _pi(indices), _pd(vals), 4, int(row_align), int(sigma), info, None, None, None, None)
    if rc:
        raise GmgError(rc, "gmg_host_plan_level")
    new2old = np.empty(int(info[0]), np.int32); color_begin = np.zeros(int(info[1]) + 1, np.int32)
    rc = lib().gmg_host_plan_level(int(n), _pi(indptr), _pi(indices), _pd(vals), 4, int(row_align), int(sigma), info, _pi(new2old), _pi(color_begin), None, None)
    if rc:
        raise GmgError(rc, "gmg_host_plan_level")
    return {"n_pad": int(info[0]), "n_colors": int(info[1]), "colored_ahead": bool(info[5]), "new2old": new2old, "color_begin": color_begin}


def host_fine_block_rule(A):
    """(blocked, reason) of the rule behind gmg_config::block_fine for the system matrix A (host only): reason 0 chosen, 1 rows too short, 2 signs."""
    a = _csc(A)
    blocked, reason = C.c_int(0), C.c_int(0)
    rc = lib().gmg_host_fine_block_rule(a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), C.byref(blocked), C.byref(reason))
    if rc:
        raise GmgError(rc, "gmg_host_fine_block_rule")
    return bool(blocked.value), int(reason.value)


def host_ldlt_solve(A, b):
    """x = A^-1 b with the product's coarsest-level solver (host sparse LDL^T); returns (x, nnz(L))."""
    a = _csc(A)
    B = _f64(b)
    X = np.empty_like(B, order="F")
    nnz = C.c_int64()
    rc = lib().gmg_host_ldlt_solve(a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), _pd(B), B.shape[1], _pd(X), C.byref(nnz))
    if rc:
        raise GmgError(rc, "gmg_host_ldlt_solve (zero pivot?)")
    return (X[:, 0].copy() if np.asarray(b).ndim == 1 else X), nnz.value


def host_ldlt_probe(A, b, reps: int = 3) -> str:
    """Report of the coarsest-level solver's probe (gravomg_hip_internal.h): timings on 1 .. 8 threads, bitwise agreement of the team
    solves with the one-thread solve, supernodal against simplicial factorisation."""
    a = _csc(A)
    B = _f64(b)
    buf = C.create_string_buffer(1 << 16)
    rc = lib().gmg_host_ldlt_probe(a.shape[0], _pi(a.indptr), _pi(a.indices), _pd(a.data), _pd(B), int(reps), buf, len(buf))
    if rc:
        raise GmgError(rc, "gmg_host_ldlt_probe")
    return buf.value.decode()
