"""Soak of the engine-driven multi-GPU cycle: W processes on one GPU (hipIpc mailboxes, device-initiated exchanges), R repetitions of load -> cycles -> fetch and of the
collective solve, exact and hybrid smoother alternating, every repetition compared bit for bit with the first (the exchange protocol -- sequence numbers, parity-double-buffered
regions, the 2 d-double all-reduce -- must not depend on timing).  python scripts/soak_p2p.py [world=3] [reps=200] [kind=poisson-big] [shard=2]"""
import os, sys, socket, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def worker(rank, world, port, q, kind, shard, reps):
    try:
        os.environ["GMG_P2P_SHARED_DEVICE"] = "1"; os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"; os.environ["LOCAL_WORLD_SIZE"] = str(world); os.environ["GMG_P2P_TIMEOUT_S"] = "30"
        import torch.distributed as dist
        from gravo_mg_amd import cabi
        from tests.test_gpu_p2p import _problem
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        P = _problem(kind)
        eng = cabi.Engine(row_align=64 * world, dist_shard_levels=shard, block_lanes=1, block_fine=0)
        eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
        rk = cabi.P2PCycle(eng, rank, world, P.rhs.shape[1])
        blobs = [None] * world
        dist.all_gather_object(blobs, rk.export()); rk.connect(blobs=blobs); dist.barrier()
        ref = {}
        t0 = time.time()
        for rep in range(reps):
            hybrid = rep % 4 == 3
            rk.set_smoother(1 if hybrid else (2 if rep % 4 == 2 else 0))      # (2: the colour exchanges folded into the colour launches -- the exact smoother's bits)
            if rep % 2 == 0:
                rk.load(P.rhs, P.rhs); hist = rk.cycles(5, 2); x = rk.fetch()
                sig = (hist.tobytes(), x.tobytes())
            else:
                xs, its, ress = rk.solve(P.rhs, P.rhs, tol=1e-5, stop_type=2, max_iter=60)
                sig = (int(its), float(ress), xs.tobytes())
            key = (hybrid, rep % 2)
            if key in ref: assert ref[key] == sig, (rank, rep, key)
            else: ref[key] = sig
            if rank == 0 and rep % 50 == 0: print("rep", rep, "%.1f s" % (time.time() - t0), flush=True)
        dist.barrier()
        q.put((rank, None)); dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, traceback.format_exc() + repr(e)))

if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 3; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    kind = sys.argv[3] if len(sys.argv) > 3 else "poisson-big"; shard = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, kind, shard, reps)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs: p.join(30)
    errs = [f"rank {r}: {e}" for r, e in got if e]
    print("\n".join(errs) if errs else f"soak ok: {world} ranks x {reps} repetitions ({kind}, shard {shard})")
    sys.exit(1 if errs else 0)
