"""N processes on one GPU (hipIpc mailboxes): residual histories of the exact per-colour exchange and of hybrid Gauss-Seidel.
  python scripts/p2p_hybrid_probe.py <world> <kind: poisson|poisson-big> <shard>"""
import os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, kind, shard):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"; os.environ["LOCAL_WORLD_SIZE"] = str(world); os.environ["GMG_P2P_TIMEOUT_S"] = "20"; os.environ["GMG_P2P_SHARED_DEVICE"] = "1"
    import numpy as np
    import torch.distributed as dist
    from gravo_mg_amd import cabi
    from tests.test_gpu_p2p import _problem
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    P = _problem(kind)
    eng = cabi.Engine(row_align=64 * world, dist_shard_levels=shard, block_lanes=1)
    eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    rk = cabi.P2PCycle(eng, rank, world, P.rhs.shape[1])
    blobs = [None] * world
    dist.all_gather_object(blobs, rk.export()); rk.connect(blobs); dist.barrier()
    if os.environ.get("PROBE_TEST_SEQUENCE"):
        import time
        rk.load(P.rhs, P.rhs); rk.cycles(4, 2); rk.fetch()
        t = time.time(); xs, its, ress = rk.solve(P.rhs, P.rhs, tol=1e-4, stop_type=2, max_iter=50); t1 = time.time() - t
        rk.set_smoother(True)
        t = time.time()
        try:
            xh, ith, resh = rk.solve(P.rhs, P.rhs, tol=1e-4, stop_type=2, max_iter=100)
        except Exception as e:
            ith, resh = -1, repr(e)[:60]
        print(rank, "exact", its, ress, "%.2fs" % t1, "hybrid", ith, resh, "%.2fs" % (time.time() - t), flush=True)
        rk.set_smoother(False); dist.barrier(); dist.destroy_process_group(); return
    for hybrid in (False, True):
        rk.set_smoother(hybrid); rk.load(P.rhs, P.rhs)
        hist = []
        for i in range(30):
            try:
                hist.append(float(rk.cycles(1, 2)[0]))
            except Exception as e:
                hist.append(repr(e)[:80]); break
            if hist[-1] <= 1e-4 or not np.isfinite(hist[-1]): break
        if rank == 0: print("hybrid" if hybrid else "exact ", world, kind, shard, ["%.2e" % v if isinstance(v, float) else v for v in hist], flush=True)
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world, kind, shard = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, kind, shard), nprocs=world, join=True)
