"""Engine-driven multi-GPU cycle (gmg_p2p_*: device-initiated peer-to-peer exchanges, no collective call per colour) on ONE
MI355X: 2 / 3 / 4 ranks as separate PROCESSES that map each other's mailboxes through hipIpc handles -- the mechanism real
multi-GPU runs use, here with all ranks on the same device (the blobs travel over a gloo process group).  In every case the solution must equal the single-engine one bit for bit -- the partition must not
change the iterates (colours are global) -- and the residual history to rounding (the norm sums are formed per rank and then
added in rank order, identically on every rank)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import problems

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(cabi, P, world):
    eng = cabi.Engine(row_align=64 * world)
    eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    return eng


def _reference(cabi, P, world, cycles):
    ref = _engine(cabi, P, world)                    # same padded layout, one handle doing everything
    ref.load_problem(P.rhs, P.rhs)
    hist = ref.run_cycles(cycles, 2)
    return hist, ref.fetch_solution()


def _worker(rank, world, port, q, kind):
    try:
        sys.path.insert(0, ROOT)
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        import torch.distributed as dist
        from gravo_mg_amd import cabi
        from tests import problems as pr
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        P = pr.torus_problem(96, 80, "poisson", 30) if kind == "poisson" else pr.torus_problem(64, 60, "smoothing", 60)
        eng = cabi.Engine(row_align=64 * world)
        eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
        rk = cabi.P2PCycle(eng, rank, world, P.rhs.shape[1])
        blobs = [None] * world
        dist.all_gather_object(blobs, rk.export())
        rk.connect(blobs=blobs)
        dist.barrier()
        rk.load(P.rhs, P.rhs)
        hist = rk.cycles(4, 2)
        x = rk.fetch()
        dist.barrier()
        q.put((rank, hist, x, None))
        dist.destroy_process_group()
    except Exception as e:              # noqa: BLE001
        import traceback
        q.put((rank, None, None, traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world,kind", [(2, "poisson"), (3, "poisson"), (4, "smoothing-d3")])
def test_processes_through_ipc_handles(cabi, world, kind):
    import torch.multiprocessing as mp
    P = problems.torus_problem(96, 80, "poisson", 30) if kind == "poisson" else problems.torus_problem(64, 60, "smoothing", 60)
    want_hist, want_x = _reference(cabi, P, world, 4)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, hist, x, err in got:
        assert err is None, err
        np.testing.assert_allclose(hist, want_hist, rtol=1e-12)
        assert np.array_equal(x, want_x), rank
