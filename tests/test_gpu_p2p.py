"""Engine-driven multi-GPU cycle (gmg_p2p_*: device-initiated peer-to-peer exchanges, no collective call per colour) on ONE
MI355X: 2 / 3 / 4 / 8 ranks as separate PROCESSES that map each other's mailboxes through hipIpc handles -- the mechanism real
multi-GPU runs use, here with all ranks on the same device (the blobs travel over a gloo process group).  In every case the solution must equal the single-engine one bit for bit -- the partition must not
change the iterates (colours are global) -- and the residual history to rounding (the norm sums are formed per rank and then
added in rank order, identically on every rank)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import problems  # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(cabi, P, world):
    # block_lanes=1: the small levels of these problems take the layout of big ones (one lane per row, entry-parallel block sweep),
    # the layout the level-1 partition is built for
    eng = cabi.Engine(row_align=64 * world, block_lanes=1)
    eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    return eng


def _reference(cabi, P, world, cycles):
    ref = _engine(cabi, P, world)                    # same padded layout, one handle doing everything
    ref.load_problem(P.rhs, P.rhs)
    hist = ref.run_cycles(cycles, 2)
    return hist, ref.fetch_solution()


def _problem(kind):
    from tests import problems as pr
    if kind == "poisson":
        return pr.torus_problem(96, 80, "poisson", 30)
    if kind == "poisson-big":
        return pr.torus_problem(300, 280, "poisson", 100)       # level 1: ~10 k rows, ~165 blocks
    if kind == "cloud":
        return pr.pointcloud_problem(9000, 8, 120)              # kNN operator: level 0 itself runs the block sweep (gmg_config::block_fine)
    return pr.torus_problem(64, 60, "smoothing", 60)


def _worker(rank, world, port, q, kind, shard, budget, partition=False, exchange=0):
    try:
        sys.path.insert(0, ROOT)
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ["LOCAL_WORLD_SIZE"] = str(world)          # what torch.distributed.run sets: a rank's host threads are its share of the CPUs
        if world >= 8:                                       # on some boxes eight processes TIME-SLICE the one device: a peer's push can be seconds away
            os.environ["GMG_P2P_TIMEOUT_S"] = "60"            # (~0.1 s per exchange there, microseconds where they run side by side)
        import torch.distributed as dist
        from gravo_mg_amd import cabi
        from tests.test_gpu_p2p import _problem
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        P = _problem(kind)
        assert cabi.default_host_threads() == max(1, budget // world), (cabi.default_host_threads(), budget, world)
        eng = cabi.Engine(row_align=64 * world, dist_shard_levels=shard, block_lanes=1, dist_exchange=exchange)
        if partition:                                        # lay out and keep this rank's rows of levels 0 (and 1) only
            eng.dist_partition(rank, world)
        eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
        if partition:
            with pytest.raises(cabi.GmgError, match="partitioned"):      # no whole operator on this handle: the single-process entry points say so
                eng.residual(0, P.rhs, P.rhs)
        blocked0 = eng.level_blocks(0) is not None           # level 0 on the block sweep: partitioned by runs of whole blocks, one exchange per sweep
        assert blocked0 == (kind == "cloud")
        rk = cabi.P2PCycle(eng, rank, world, P.rhs.shape[1])
        assert rk.stat("level1_partitioned") == (1.0 if shard == 2 else 0.0)
        blobs = [None] * world
        dist.all_gather_object(blobs, rk.export())
        rk.connect(blobs=blobs)
        dist.barrier()
        rk.load(P.rhs, P.rhs)
        hist = rk.cycles(4, 2)
        x = rk.fetch()
        xs, its, ress = rk.solve(P.rhs, P.rhs, tol=1e-4, stop_type=2, max_iter=50)      # the collective solve loop completes on this budget
        assert ress <= 1e-4 and its >= 1 and eng.timing("diverged") == 0.0
        if exchange == 0:
            # the opt-in publication without the cache write-back (gmg_p2p_set_fences(0); default: release / acquire fences): the same bits
            assert rk.stat("fenced") == 1.0
            rk.set_fences(False)
            rk.load(P.rhs, P.rhs)
            hist_nf = rk.cycles(4, 2)
            x_nf = rk.fetch()
            rk.set_fences(True)
            dist.barrier()
            assert np.array_equal(hist_nf, hist) and np.array_equal(x_nf, x)
        # hybrid Gauss-Seidel (one exchange per sweep): another convergent iteration -- same solution to the tolerance; more cycles the smaller a
        # rank's piece is (this 7 680-vertex problem: 4 exact, 7 at four ranks, 11 at eight).  No assertion between collectives: a rank that
        # leaves early shows up on the others as a time-out.  Eight ranks run two hybrid cycles only (see the time-out note above)
        rk.set_smoother(True)
        if world >= 8:
            rk.load(P.rhs, P.rhs)
            hh = rk.cycles(2, 2)
            ok_h = bool(hh[1] < hh[0] < 1.0)
            why_h = tuple(hh)
        else:
            xh, ith, resh = rk.solve(P.rhs, P.rhs, tol=1e-4, stop_type=2, max_iter=100)
            ok_h = bool(resh <= 1e-4 and its <= ith <= 3 * its + 3 and np.abs(xh - xs).max() <= 1e-2 * np.abs(xs).max())
            why_h = (resh, ith, its, float(np.abs(xh - xs).max() / np.abs(xs).max()))
        rk.set_smoother(False)
        dist.barrier()
        assert ok_h, why_h
        # the colour exchanges folded into the colour launches (gmgk::gs_color_push): the same iterates bit for bit, 4 C fewer launches per cycle
        if exchange == 0 and not blocked0:
            n0 = rk.stat("exchange_launches")
            rk.load(P.rhs, P.rhs)
            hist_u = rk.cycles(2, 2)
            n1 = rk.stat("exchange_launches")
            rk.set_smoother(2)
            rk.load(P.rhs, P.rhs)
            hist_f = rk.cycles(4, 2)
            x_f = rk.fetch()
            n2 = rk.stat("exchange_launches")
            rk.set_smoother(0)
            dist.barrier()
            assert np.array_equal(hist_f, hist) and np.array_equal(x_f, x), (hist_f, hist)
            assert np.array_equal(hist_u, hist[:2])
            assert (n2 - n1) / 4 < (n1 - n0) / 2, (n0, n1, n2)
        kinds = [] if world >= 8 else (["color0", "halo_all", "rows0"] + (["x1_halo", "rows1", "r0_halo"] if shard == 2 else []))
        us = {k: 1e3 * rk.bench_kind(k, 20) for k in kinds}
        assert all(v > 0 for v in us.values()), us
        dist.barrier()
        q.put((rank, hist, x, None))
        dist.destroy_process_group()
    except Exception as e:              # noqa: BLE001
        import traceback
        q.put((rank, None, None, traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world,kind,shard,partition", [(2, "poisson", 2, False), (3, "poisson", 2, False), (3, "poisson", 1, False), (4, "smoothing-d3", 2, False),
                                                        (4, "poisson-big", 2, False), (2, "poisson-big", 1, False), (8, "poisson", 2, False), (8, "smoothing-d3", 1, False),
                                                        (2, "poisson", 2, True), (3, "poisson", 1, True), (4, "smoothing-d3", 2, True), (4, "poisson-big", 2, True),
                                                        (2, "poisson-big", 1, True), (8, "poisson", 2, True),
                                                        (2, "cloud", 2, False), (3, "cloud", 1, False), (4, "cloud", 2, False)])      # 8: the target's rank count; cloud: blocked level 0
def test_processes_through_ipc_handles(cabi, world, kind, shard, partition, exchange=0):
    """shard = levels partitioned over the ranks: 2 = level 0 by rows per colour and level 1 by blocks (default), 1 = level 0 only.
    partition: the SET-UP is partitioned too (gmg_dist_partition) -- every rank lays out and keeps only its rows of levels 0 (and 1);
    the other ranks' rows are zero-width slices of the same global numbering, so the iterates stay those of one GPU, bit for bit."""
    import torch.multiprocessing as mp
    P = _problem(kind)
    want_hist, want_x = _reference(cabi, P, world, 4)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    budget = cabi.default_host_threads()             # this process has no LOCAL_WORLD_SIZE: the CPUs it may use
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind, shard, budget, partition, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=420) for _ in range(world)]
    for p in procs:
        p.join(60)
    errs = [f"rank {rank}: {err}" for rank, hist, x, err in got if err is not None]
    assert not errs, "\n".join(sorted(errs, key=lambda e: "timed out" in e))      # (time-outs last: they are the echo of another rank's failure)
    for rank, hist, x, err in got:
        np.testing.assert_allclose(hist, want_hist, rtol=1e-12)
        assert np.array_equal(x, want_x), rank


@pytest.mark.parametrize("world,kind,shard,partition", [(2, "poisson", 2, False), (3, "poisson", 1, False), (3, "smoothing-d3", 2, True), (4, "poisson-big", 2, True), (2, "cloud", 2, False)])
def test_collective_exchange_sequence_gives_the_same_iterates(cabi, world, kind, shard, partition):
    """gmg_config::dist_exchange: every exchange of the cycle as pack -> all-gather -> unpack on the engine's stream (the north star's RCCL
    all-gather of the halo, csrc/engine_dist.hip.hpp::coll_exchange) instead of one mailbox launch.  Here with the all-gather emulated through
    hipIpc mappings (mode 2: the ranks share the one device, which RCCL refuses); tests/test_gpu_multi_device.py runs mode 1 -- ncclAllGather --
    whenever two devices are visible.  Same plan, same kernels around it: the single-engine solution bit for bit, level 1 partitioned or
    replicated, whole or partitioned set-up, d = 1 and 3, the hybrid smoother, every exchange kind timed."""
    test_processes_through_ipc_handles(cabi, world, kind, shard, partition, exchange=2)


def _same_device_worker(rank, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ.pop("GMG_P2P_SHARED_DEVICE", None)       # what a real job runs with
        import torch.distributed as dist
        from gravo_mg_amd import cabi
        from tests.test_gpu_p2p import _problem
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
        P = _problem("poisson")
        eng = cabi.Engine(row_align=128, block_lanes=1)
        eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
        rk = cabi.P2PCycle(eng, rank, 2, 1)
        blobs = [None, None]
        dist.all_gather_object(blobs, rk.export())
        try:
            rk.connect(blobs=blobs)
            q.put((rank, "connected"))
        except cabi.GmgError as e:
            q.put((rank, str(e)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:              # noqa: BLE001
        import traceback
        q.put((rank, "worker failed: " + traceback.format_exc() + repr(e)))


def test_two_ranks_on_one_device_are_refused_at_connect_time(cabi):
    """A mis-mapped HIP_VISIBLE_DEVICES (every rank on device 0) must be an error when the ranks connect, not seconds of device-side spinning per
    exchange: the connection records carry the device's UUID (gmg_p2p_connect; GMG_P2P_SHARED_DEVICE=1 -- what this suite sets -- allows it)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_same_device_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    for r in range(2):
        assert "same device" in got[r] and "GMG_P2P_SHARED_DEVICE" in got[r], got


_RCCL_ONE_RANK = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from gravo_mg_amd import cabi
from tests.test_gpu_p2p import _problem, _reference
out = {}
for kind in ("poisson", "smoothing-d3"):
    P = _problem(kind)
    want_hist, want_x = _reference(cabi, P, 1, 4)
    eng = cabi.Engine(row_align=64, block_lanes=1, dist_exchange=1)
    eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    rk = cabi.P2PCycle(eng, 0, 1, P.rhs.shape[1])
    mode = rk.stat("collective_mode")
    rk.connect_rccl(cabi.rccl_unique_id())            # dlopen(librccl) -> ncclGetUniqueId -> ncclCommInitRank(nranks = 1)
    rk.load(P.rhs, P.rhs)
    hist = rk.cycles(4, 2)                            # every exchange of the cycle: pack -> ncclAllGather -> unpack on the engine's stream
    n_coll = rk.stat("collective_exchanges")
    x = rk.fetch()
    diff, cnt = rk.collective_roundtrip()
    xs, its, ress = rk.solve(P.rhs, P.rhs, tol=1e-4, stop_type=2, max_iter=50)
    out[kind] = {"mode": mode, "hist_equal": bool(np.allclose(hist, want_hist, rtol=1e-12, atol=0)), "x_equal": bool(np.array_equal(x, want_x)), "collectives": n_coll,
                 "roundtrip_diff": diff, "roundtrip_doubles": cnt, "solve_residue": ress, "solve_iters": its,
                 "colors": eng.level_info(0)["n_colors"]}
    del rk; eng.close()
out["librccl_mapped"] = any("librccl" in l for l in open("/proc/self/maps"))
print("RESULT " + json.dumps(out))
"""


def test_rccl_all_gather_backend_on_one_rank(cabi):
    """First contact with the real library on the one-GPU box: gmg_config::dist_exchange = 1 with world = 1 -- RCCL takes a one-rank communicator --
    runs dlopen(librccl) -> ncclGetUniqueId -> ncclCommInitRank -> every exchange kind of the cycle as pack -> ncclAllGather -> unpack on the
    engine's stream, the residual sums through the gathered buffer, and gives the plain engine's iterates bit for bit (d = 1 and 3).  In a
    child process with a time-out: a communicator that cannot initialise on some box must fail this test, not hang the suite."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    res = json.loads([l for l in run.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res.pop("librccl_mapped") is True
    for kind, r in res.items():
        assert r["mode"] == 1.0 and r["hist_equal"] and r["x_equal"], (kind, r)
        # per cycle: (pre + post) x colours + the rows of r0 + the halo of all colours + the norm sums
        assert r["collectives"] >= 4 * (4 * r["colors"] + 3), (kind, r)
        assert r["roundtrip_diff"] == 0.0 and r["roundtrip_doubles"] > 1000, (kind, r)
        assert r["solve_residue"] <= 1e-4 and r["solve_iters"] >= 1, (kind, r)


def _absent_peer_worker(rank, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        import time
        import torch.distributed as dist
        from gravo_mg_amd import cabi
        from tests.test_gpu_p2p import _problem
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
        P = _problem("poisson")
        eng = cabi.Engine(row_align=128, block_lanes=1)
        eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
        rk = cabi.P2PCycle(eng, rank, 2, 1)
        blobs = [None, None]
        dist.all_gather_object(blobs, rk.export())
        rk.connect(blobs=blobs)
        dist.barrier()
        rk.load(P.rhs, P.rhs)
        if rank == 0:
            t = time.perf_counter()
            try:
                rk.cycles(2, 2)
                q.put((rank, "no error", time.perf_counter() - t))
            except cabi.GmgError as e:
                q.put((rank, str(e), time.perf_counter() - t))
        else:
            time.sleep(9.0)            # connected, mailboxes mapped -- but never takes part in a cycle
            q.put((rank, "absent", 0.0))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:              # noqa: BLE001
        import traceback
        q.put((rank, "worker failed: " + traceback.format_exc() + repr(e), 0.0))


def test_a_missing_peer_is_an_error_not_a_hang(cabi):
    """A rank that never runs its share of the cycle: the device-side wait of the first exchange gives up after ~4 s, every launch
    queued behind it returns at once, and gmg_p2p_cycles reports GMG_ERR_STATE."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_absent_peer_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (msg, dt)) for r, msg, dt in [q.get(timeout=120) for _ in range(2)])
    for p in procs:
        p.join(60)
    assert got[1][0] == "absent", got
    msg, dt = got[0]
    assert "timed out" in msg, got
    assert 3.0 <= dt <= 8.5, dt          # one 4 s wait, not one per queued exchange


def _dropin_inputs(kind="mesh"):
    from gravo_mg_amd import meshgen
    if kind == "cloud":                                       # kNN graph Laplacian: level 0 on the block sweep (gmg_config::block_fine), d = 1
        P = meshgen.torus_points(20000, noise=0.002)
        S, mass = meshgen.knn_graph_laplacian(P, 8)
        lhs, rhs = meshgen.poisson_system(S, mass)
        return P, meshgen.neighbors_from_stiffness(S), mass, lhs, rhs
    V, F = meshgen.torus_mesh(150, 140)
    S, mass = meshgen.cotan_laplacian(V, F)
    lhs, rhs = meshgen.smoothing_system(S, mass, V)           # the demos' call: n x 3 right-hand side
    return V, meshgen.neighbors_from_stiffness(S), mass, lhs, rhs


def _dropin_worker(rank, world, port, q, kind="mesh"):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "gravo_mg_amd", "dropin"))
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        import scipy.sparse as sp
        import torch.distributed as dist
        import gravomg
        from tests.test_gpu_p2p import _dropin_inputs
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        V, neigh, mass, lhs, rhs = _dropin_inputs(kind)
        solver = gravomg.MultigridSolver(V, neigh, sp.diags(mass).tocsc(), lower_bound=100, tolerance=1e-6)

        def all_gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        solver.enable_distributed(rank, world, all_gather, device=0)           # every rank on the one GPU of the test box
        x = solver.solve(lhs, rhs)
        x2 = solver.solve(lhs, 2.0 * rhs)                                      # same layout: no second connect
        dist.barrier()
        q.put((rank, x, x2, dict(solver.distributed_info), None))
        dist.destroy_process_group()
    except Exception as e:              # noqa: BLE001
        import traceback
        q.put((rank, None, None, None, traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world,kind", [(2, "mesh"), (3, "mesh"), (2, "cloud")])
def test_multigridsolver_solve_as_a_collective_over_ranks(cabi, world, kind):
    """gravomg.MultigridSolver.enable_distributed(): solve() of the drop-in class runs the engine-driven multi-GPU cycle on the engine
    it already owns (`prepare_system` hands out its C-ABI handle) -- here `world` processes on one GPU.  Same number of V-cycles and,
    bit for bit, the single-process solution on every rank (global colours: the partition does not change the iterates).  cloud: a kNN operator,
    whose level 0 runs the block sweep on one GPU and on N (partitioned by runs of whole blocks; its set-up falls back to whole operators)."""
    import scipy.sparse as sp
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "gravo_mg_amd", "dropin"))
    import gravomg
    V, neigh, mass, lhs, rhs = _dropin_inputs(kind)
    single = gravomg.MultigridSolver(V, neigh, sp.diags(mass).tocsc(), lower_bound=100, tolerance=1e-6)
    want = single.solve(lhs, rhs)
    want_iters = int(single.solver_timing["iterations"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dropin_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, x, x2, info, err in got:
        assert err is None, err
        assert info["iterations"] == want_iters and info["residue"] <= 1e-6 and info["world"] == world
        assert np.array_equal(x, want), (rank, np.abs(x - want).max())
        np.testing.assert_allclose(x2, 2.0 * want, rtol=1e-9, atol=1e-12 * np.abs(want).max())


def _bench_workload_file(tmpdir):
    """The 3 M-vertex bench workload, built once by the parent and loaded by every rank (the hierarchy is rebuilt per process: 0.3 s)."""
    import scipy.sparse as sp
    sys.path.insert(0, ROOT)
    from gravo_mg_amd import meshgen
    V, F = meshgen.torus_mesh(1732, 1732)
    S, mass = meshgen.cotan_laplacian(V, F)
    neigh = meshgen.neighbors_from_stiffness(S)
    lhs, rhs = meshgen.poisson_system(S, mass, tau=1e-6, seed=42, d=1)
    lhs = sp.csc_matrix(lhs)
    path = os.path.join(str(tmpdir), "bench3m.npz")
    np.savez(path, V=V, neigh=neigh, mass=mass, indptr=lhs.indptr, indices=lhs.indices, data=lhs.data, rhs=rhs)
    return path


def _load_bench_workload(path):
    import scipy.sparse as sp
    z = np.load(path)
    n = z["V"].shape[0]
    return z["V"], z["neigh"], z["mass"], sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=(n, n)), z["rhs"]


def _memory_worker(rank, world, port, q, path):
    try:
        sys.path.insert(0, ROOT)
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ["LOCAL_WORLD_SIZE"] = str(world)
        os.environ["GMG_P2P_TIMEOUT_S"] = "120"               # processes sharing one device may be time-sliced
        import time
        import torch.distributed as dist
        from gravo_mg_amd import cabi
        from tests.test_gpu_p2p import _load_bench_workload
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        V, neigh, mass, lhs, rhs = _load_bench_workload(path)
        H = cabi.Hierarchy(V, neigh, lower_bound=1000)
        eng = cabi.Engine(row_align=64 * world, block_fine=0)
        eng.dist_partition(rank, world)
        eng.use_hierarchy(H); eng.set_mass(mass)
        t = time.perf_counter(); eng.set_system(lhs); set_ms = 1e3 * (time.perf_counter() - t)
        rk = cabi.P2PCycle(eng, rank, world, 1)
        blobs = [None] * world
        dist.all_gather_object(blobs, rk.export())
        rk.connect(blobs=blobs)
        dist.barrier()
        rk.load(rhs, rhs)
        hist = rk.cycles(2, 2)
        stats = {"device_bytes": rk.stat("device_bytes"), "device_bytes_peak": eng.timing("device_bytes_peak"), "set_system_ms": set_ms,
                 "plan_ms": eng.timing("dist_plan_ms"), "level1_partitioned": rk.stat("level1_partitioned")}
        dist.barrier()
        q.put((rank, hist, stats, None))
        dist.destroy_process_group()
    except Exception as e:              # noqa: BLE001
        import traceback
        q.put((rank, None, None, traceback.format_exc() + repr(e)))


def test_partitioned_set_up_holds_a_rank_s_share_of_the_3m_system(cabi, tmp_path):
    """round-4 verdict item 1a: with a partitioned set-up (gmg_dist_partition) a rank of a P-rank job holds its rows of levels 0-1 plus the
    replicated small levels and the (whole) vectors: device_bytes <= (1 / P + 0.15) x what one GPU holds for the 3 M-vertex bench system, for
    P = 2, 4, 8 (here: P processes on one device), and the first cycles reproduce the single-GPU residues."""
    import json
    import torch.multiprocessing as mp
    path = _bench_workload_file(tmp_path)
    V, neigh, mass, lhs, rhs = _load_bench_workload(path)
    H = cabi.Hierarchy(V, neigh, lower_bound=1000)
    one = cabi.Engine(block_fine=0)
    one.use_hierarchy(H); one.set_mass(mass); one.set_system(lhs)
    one.load_problem(rhs, rhs)
    want = one.run_cycles(2, 2)
    single = one.timing("device_bytes_now")
    one.close(); del one, H
    report = {"single_gpu_device_bytes": single}
    for world in (2, 4, 8):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_memory_worker, args=(r, world, port, q, path)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=900) for _ in range(world)]
        for p in procs:
            p.join(60)
        errs = [f"rank {rank}: {err}" for rank, hist, stats, err in got if err is not None]
        assert not errs, "\n".join(errs)
        worst = max(stats["device_bytes"] for _, _, stats, _ in got)
        report[f"P{world}"] = {"device_bytes_max": worst, "ratio": worst / single, "bound": 1.0 / world + 0.15,
                               "device_bytes_peak_max": max(st["device_bytes_peak"] for _, _, st, _ in got),
                               "set_system_ms_max": max(st["set_system_ms"] for _, _, st, _ in got), "plan_ms_max": max(st["plan_ms"] for _, _, st, _ in got)}
        for rank, hist, stats, err in got:
            np.testing.assert_allclose(hist, want, rtol=1e-12)
            assert stats["level1_partitioned"] == 1.0
            assert stats["device_bytes"] <= (1.0 / world + 0.15) * single, (world, rank, stats, single)
    print("partitioned set-up:", json.dumps(report))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "partitioned_setup_memory.json"), "w") as f:
            json.dump(report, f, indent=1)
