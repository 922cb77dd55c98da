"""One engine handle driven through many systems: different sizes, patterns, hierarchies, numbers of right-hand sides,
interleaved with repeats (ordering cache hits), failures (singular / malformed input) and re-use after a failure.
Guards the per-handle device memory pool, the ordering cache and the error paths."""
import numpy as np
import pytest
import scipy.sparse as sp

from tests import problems

pytestmark = pytest.mark.gpu


def _free_device_bytes():
    import torch
    free, _ = torch.cuda.mem_get_info()
    return free


def test_one_handle_many_systems(cabi, oracle):
    eng = cabi.Engine()
    cases = [problems.torus_problem(48, 40, "poisson", 60), problems.torus_problem(64, 60, "smoothing", 60),
             problems.pointcloud_problem(3000), problems.torus_problem(96, 80, "poisson", 30), problems.sphere_problem(6000),
             problems.torus_problem(48, 40, "poisson", 60, order="random")]
    rng = np.random.default_rng(0)
    free_after_warmup = None
    for rnd in range(4):
        for P in (cases if rnd % 2 == 0 else cases[::-1]):
            eng.set_prolongations(P.U); eng.set_mass(P.mass)
            for rep in range(2):                                   # second call: same pattern -> cached orderings
                scale = 1.0 + 0.5 * rep
                eng.set_system(P.lhs * scale)
                assert eng.timing("setup_ordering_cached") == float(rep)
                d = int(rng.integers(1, 6))
                B = np.repeat(P.rhs[:, :1], d, axis=1) * rng.uniform(0.5, 2.0, size=(1, d))
                x, it, res, _ = eng.solve(B, tol=1e-6, max_iter=100)
                assert res <= 1e-6 and x.shape == B.shape
                assert abs(oracle.residual_check(P.lhs * scale, P.mass, B, x, 2) - res) <= 1e-3 * res + 1e-7
            # a malformed system in between must fail loudly and leave the handle usable
            bad = sp.csc_matrix(P.lhs.shape)
            with pytest.raises(cabi.GmgError):
                eng.set_system(bad + sp.identity(P.n, format="csc") * 0.0)
            with pytest.raises(cabi.GmgError):
                eng.solve(P.rhs)                                    # no system after the failure
            eng.set_system(P.lhs)
            x, it, res, _ = eng.solve(P.rhs, tol=1e-6)
            assert res <= 1e-6
        if rnd == 1:
            free_after_warmup = _free_device_bytes()
    # the pool parks blocks for reuse but must not grow without bound: two more rounds cost (almost) no extra device memory
    assert free_after_warmup - _free_device_bytes() < 64 << 20


def test_host_planner_set_system_repeated_on_a_small_mesh(cabi):
    """Regression (round 3): with device_setup=False level 0 is ORDERED from the caller's arrays but LAID OUT from the engine's host
    copy of the LHS, which is made by a background task -- on a small mesh the ordering used to finish first and the layout task read a
    half-made copy (a segmentation fault in about one of three runs of scripts/parity_sweep.py).  Many set-ups in a row, each checked."""
    from gravo_mg_amd import meshgen
    V, F = meshgen.torus_mesh(180, 150)
    S, mass = meshgen.cotan_laplacian(V, F)
    lhs, rhs = meshgen.poisson_system(S, mass)
    H = cabi.Hierarchy(V, meshgen.neighbors_from_stiffness(S), lower_bound=1000)
    ref = None
    for rep in range(40):
        eng = cabi.Engine(device_setup=False)
        eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        x, it, res, _ = eng.solve(rhs, tol=1e-4, max_iter=100)
        assert res <= 1e-4
        if ref is None:
            ref = (x, it)
        else:
            assert it == ref[1] and np.array_equal(x, ref[0])
        del eng


def test_handles_in_threads_do_not_deadlock_on_a_gated_stream():
    """Regression (round 4): a polled handle parks its stream behind a word its own host thread writes (hipStreamWaitValue64) while it
    enqueues the way up; hipFree / hipHostFree / hipStreamDestroy in ANOTHER thread (another handle destroyed or re-set) wait for the whole
    device under the runtime's lock -- the gate's thread then never gets to its next launch: two threads in gmg_destroy, one in gmg_solve,
    for ever (within 1-25 rounds of scripts/soak_factor.py).  Gates now hold a process-wide shared lock that those calls take exclusively
    (engine_state.hip.hpp::gate_mutex).  Three handles in three threads, values-only and cold set-ups, solves, destruction, 60 rounds;
    every solve repeats the single-threaded run bit for bit.  Run in a subprocess so that a relapse is a time-out, not a hung suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GMG_SOAK_WATCHDOG="60")
    out = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(root, "scripts", "soak_factor.py"), "60"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and "soak ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
