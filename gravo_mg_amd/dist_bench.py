"""bench.py --gpus N (N > 1): the same 3 M-vertex Poisson V-cycle, finest level row-partitioned over N ranks.
Launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W`;
also works with N = 1 (WORLD_SIZE=1) to exercise the distributed code path on one GPU.

Exchange engines (`--exchange`):
  p2p (default)  the cycle is driven inside the library (gmg_p2p_*, csrc/engine_dist.hip.hpp): every exchange is a
                 device-initiated store into the peers' mailboxes, no collective call and no Python per colour.  Level 0 is
                 partitioned by rows per colour and (--shard-levels 2, default) level 1 by blocks.  Taken only if
                 EVERY rank could set it up and its first cycles reproduce the residues of the plain single-GPU engine
                 (the iterates do not depend on the number of ranks); otherwise all ranks fall back together to
  halo           the RCCL orchestration of gravo_mg_amd/dist.py: pack -> all_gather_into_tensor -> unpack per colour
  allgather      the same with whole colour segments.
Timing: W untimed cycles, barrier + synchronize, K V-cycles each followed by the residual check, synchronize + barrier,
MAX over ranks.  Rank 0 prints ONE JSON line."""
from __future__ import annotations

import json
import os
import sys
import time


def main(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    from gravo_mg_amd import cabi
    from gravo_mg_amd.dist import DistVCycle, EngineBackend, HaloPlan

    import bench as single

    # RCCL prints a version banner on the native stdout when the communicator is created; the contract is ONE JSON
    # line on stdout, so everything until the final print goes to stderr.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the device-side time-out of an exchange (default 4 s): the first exchanges of a job meet ranks that are still loading code objects or paging the
    # image in -- a rank that is merely late must not send all ranks to the fallback
    os.environ.setdefault("GMG_P2P_TIMEOUT_S", "20")
    # GMG_DIST_BACKEND=gloo lets several ranks share one GPU (RCCL refuses that): a functional end-to-end check of the
    # N > 1 path on a 1-GPU box, not a measurement
    backend = os.environ.get("GMG_DIST_BACKEND", "nccl")
    if backend == "gloo":
        os.environ.setdefault("GMG_P2P_SHARED_DEVICE", "1")      # (read by the library at its first use, below: gmg_p2p_connect refuses ranks on one device otherwise)
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus or args.gpus <= 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    coarse_mode = {"auto": cabi.COARSE_AUTO, "host": cabi.COARSE_HOST_LDLT, "device": cabi.COARSE_DEVICE_INVERSE}[args.coarse]

    grow = world ** 0.5 if args.scaling == "weak" else 1.0                         # weak: n1 x n2 vertices per rank, same aspect ratio
    n1, n2 = int(round(args.n1 * grow)), int(round(args.n2 * grow))
    workload = f"torus{n1}x{n2}-poisson-tau1e-6-d1-{args.order}"
    H, mass, lhs, rhs = single.build_workload(n1, n2, args.order)                # deterministic: every rank builds the same

    setup_ms = {}

    def new_engine(partition=False, tag=None, dist_exchange=0):
        """partition: the set-up is partitioned too (gmg_dist_partition) -- this rank lays out and keeps its rows of levels 0-1 only.
        dist_exchange: 0 mailboxes, 1 pack -> ncclAllGather -> unpack enqueued by the engine, 2 the same with the all-gather emulated over hipIpc."""
        kw = {} if args.block_lanes is None else {"block_lanes": args.block_lanes}
        e = cabi.Engine(device=local, row_align=64 * world, block_fine=0, use_graph=False, coarse_mode=coarse_mode, dist_shard_levels=args.shard_levels,
                        dist_exchange=dist_exchange, **kw)
        if partition:
            e.dist_partition(rank, world)
        t = time.perf_counter(); e.use_hierarchy(H); t_h = 1e3 * (time.perf_counter() - t)
        e.set_mass(mass)
        t = time.perf_counter(); e.set_system(lhs); t_s = 1e3 * (time.perf_counter() - t)
        if tag:
            setup_ms[tag] = {"use_hierarchy_ms": t_h, "set_system_ms": t_s, "device_bytes_after_setup": e.timing("device_bytes"), "device_bytes_peak": e.timing("device_bytes_peak")}
        return e

    # what ONE GPU computes (plain engine holding the whole operator, same padded layout): the partitioned cycle must reproduce these residues
    n_warm = max(args.warmup, 2)
    ref = new_engine(tag="whole_operator")
    levels = [ref.level_info(k) for k in range(ref.num_levels + 1)]
    ref.load_problem(rhs, rhs)
    ref_res = ref.run_cycles(n_warm, 2)
    ref.load_problem(rhs, rhs)
    ref_hist = []
    while True:                               # the single-GPU solve to 1e-4: what the partitioned solve below must reproduce
        ref_hist += [float(v) for v in ref.run_cycles(1, 2)]
        if not (ref_hist[-1] > 1e-4 and len(ref_hist) < 100):
            break
    setup_ms["whole_operator"]["device_bytes_with_vectors"] = ref.timing("device_bytes_now")
    partitioned = world > 1 and args.exchange == "p2p" and not os.environ.get("GMG_BENCH_WHOLE_SETUP")
    if partitioned:
        ref.close(); del ref
        eng = new_engine(partition=True, tag="partitioned")
    else:
        eng = ref

    # ---- engine-driven cycle: mailboxes (hipIpc) first; should a rank be unable to set them up, the same cycle with every exchange as
    # pack -> ncclAllGather -> unpack on the engine's stream (gmg_config::dist_exchange = 1; no hipIpc) -- the Python orchestration below is the last resort
    p2p, note, exchange_us, engine_exchange = None, None, None, None
    cpu_group = dist.new_group(backend="gloo") if world > 1 else None

    def agreed(local_ok):
        """MIN over the ranks: every step below is collective, so a rank that failed locally still takes part in the next
        agreement (contributing 0) instead of leaving the others waiting in a collective it never enters."""
        flags = [None] * world
        dist.all_gather_object(flags, int(local_ok), group=cpu_group)
        return min(flags) == 1

    def engine_cycle(mode, engine, fenced=True):
        """One candidate: plan + buffers (local), connect (collective), first cycles = the single-GPU residues (collective on the devices).
        fenced (mailboxes only): gmg_p2p_set_fences -- False is the gfx950 publication without the cache write-back, taken only if those first
        cycles reproduce the single-GPU residues.  Returns (P2PCycle or None, why not)."""
        ok, why = 1, None
        cand, blob = None, None
        try:
            if mode == 0 and os.environ.get("GMG_P2P_SELFTEST_FAIL") and rank == world - 1:      # test hook: one rank cannot set the mailboxes up
                raise RuntimeError("forced by GMG_P2P_SELFTEST_FAIL")
            cand = cabi.P2PCycle(engine, rank, world, 1)
            if mode == 0:
                cand.set_fences(fenced)
            blob = cand.export() if mode != 1 else b""
        except Exception as e:          # noqa: BLE001
            ok, why = 0, f"set-up failed on rank {rank}: {e!r}"
        blobs = [None] * world
        dist.all_gather_object(blobs, blob, group=cpu_group)
        if ok and any(b is None for b in blobs):
            ok, why = 0, f"rank {[i for i, b in enumerate(blobs) if b is None]} could not set it up"
        ids = [None] * world
        if mode == 1:
            my_id = None
            if rank == 0 and ok:
                try:
                    my_id = cabi.rccl_unique_id()
                except Exception as e:      # noqa: BLE001
                    ok, why = 0, f"ncclGetUniqueId failed: {e!r}"
            dist.all_gather_object(ids, my_id, group=cpu_group)
            if ok and ids[0] is None:
                ok, why = 0, "rank 0 could not make the RCCL id"
        if ok:
            try:
                if mode == 1:
                    cand.connect_rccl(ids[0])
                else:
                    cand.connect(blobs)
            except Exception as e:      # noqa: BLE001
                ok, why = 0, f"connect failed on rank {rank}: {e!r}"
        if not agreed(ok):
            return None, why or "another rank could not set it up"
        # (collective on the devices: a rank that fails here shows up on the others as a device-side time-out -- an error, not a hang)
        try:
            cand.load(rhs, rhs)
            got = cand.cycles(n_warm, 2)
            if not np.allclose(got, ref_res, rtol=1e-9):
                ok, why = 0, f"residues {list(got)} differ from the single-GPU engine's {list(ref_res)}"
        except Exception as e:          # noqa: BLE001
            ok, why = 0, f"cycles failed on rank {rank}: {e!r}"
        flag = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            return None, why or "another rank's first cycles failed"
        return cand, None

    exchange_fences = None
    if args.exchange == "p2p" and world > 1:
        # mailboxes, first WITHOUT the release / acquire fences around the sequence words (csrc/kernels.hip.hpp::publish_order: the gfx942 / gfx950
        # form, ~3.5 us less per exchange launch) -- an opt-in of this caller, which checks the first cycles against the single-GPU residues;
        # should they differ (or the set-up fail), the library's default: the fenced form
        want_fence_free = not os.environ.get("GMG_BENCH_P2P_FENCED")
        p2p, why0 = engine_cycle(0, eng, fenced=not want_fence_free)
        exchange_fences = "none (write-through stores drained ahead of the sequence word; first cycles checked against the single-GPU residues)" if want_fence_free else "release/acquire (system scope)"
        if p2p is None and want_fence_free:
            single.log(f"[bench] rank {rank}: mailboxes without fences not taken ({why0}); trying the fenced form")
            p2p, why0b = engine_cycle(0, eng, fenced=True)
            exchange_fences = f"release/acquire (system scope); the fence-free form was not taken: {why0}"
            why0 = why0b or why0
        if p2p is not None:
            engine_exchange, note = "p2p", "peer-to-peer mailboxes (hipIpc), one exchange kernel per colour"
        else:
            single.log(f"[bench] rank {rank}: no peer-to-peer mailboxes ({why0}); trying the engine's collective exchange")
            mode = 1 if backend == "nccl" else 2
            okc, eng_c = 1, None
            try:
                if os.environ.get("GMG_BENCH_NO_ENGINE_COLLECTIVE"):          # test hook: straight to the last resort
                    raise RuntimeError("skipped by GMG_BENCH_NO_ENGINE_COLLECTIVE")
                eng_c = new_engine(partition=partitioned, dist_exchange=mode)
            except Exception as e:      # noqa: BLE001
                okc, why0 = 0, f"{why0}; collective engine: {e!r}"
            if agreed(okc):
                p2p, why1 = engine_cycle(mode, eng_c)
            else:
                p2p, why1 = None, "a rank could not build the collective engine"
            if p2p is not None:
                eng.close(); eng = eng_c
                engine_exchange = "engine-collective"
                note = (f"mailboxes unavailable ({why0}); every exchange as pack -> " + ("ncclAllGather" if mode == 1 else "all-gather emulated over hipIpc")
                        + " -> unpack on the engine's stream (gmg_config::dist_exchange)")
            else:
                note = f"peer-to-peer set-up failed ({why0}); engine collective failed ({why1})"
                single.log(f"[bench] rank {rank}: falling back to the RCCL halo exchange orchestrated from Python ({note})")
                if eng_c is not None:
                    eng_c.close()
                eng = new_engine()           # the candidates may have left their handles mid-exchange (and hold a rank's share only): the fallback starts from a fresh one
        if p2p is not None:
            sharded1 = p2p.stat("level1_partitioned") == 1.0
            C, pre, post = levels[0]["n_colors"], 2, 2
            # every exchange of a cycle, timed alone (push + wait + pull in one launch, 200 back to back), and how often a cycle runs it
            per_cycle = {"color0": None, "halo_all": 1}
            per_cycle.update({"r0_halo": 1, "x1_halo": pre + post + 1, "rows1": 1} if sharded1 else {"rows0": 1})
            exchange_us = {k: 1e3 * p2p.bench_kind(k, 200) for k in per_cycle}
            exchange_us["per_cycle"] = {**{k: v for k, v in per_cycle.items() if v}, "color<k> (each of %d colours)" % C: (pre + post)}
            exchange_us["sum_per_cycle"] = (sum(exchange_us[k] * n for k, n in per_cycle.items() if n)
                                            + (pre + post) * sum(1e3 * p2p.bench_kind(f"color{c}", 100) for c in range(C)))

    # ---- RCCL orchestration (requested, or the fallback)
    dv, be, halo = None, None, None
    if p2p is None:
        be = EngineBackend(eng, 1, rank, world, torch.device("cuda", local))
        if world > 1 and args.exchange in ("halo", "p2p"):
            t = time.perf_counter()
            new2old, cb = eng.level_ordering(0)
            A = lhs.tocsr()
            halo = HaloPlan(A.indptr, A.indices, new2old, cb, be.n_pad, world, rank, 1, device=be.device)
            single.log(f"[bench] rank {rank}: halo plan {halo.published_rows} published rows of {lhs.shape[0]} ({time.perf_counter() - t:.1f}s)")
        dv = DistVCycle(be, halo=halo)

    def run(k):
        if p2p is not None:
            return [float(v) for v in p2p.cycles(k, 2)]
        out = []
        for _ in range(k):
            dv.vcycle()
            out.append(dv.residual_norm(2))
        return out

    def load():
        (p2p.load if p2p is not None else be.load)(rhs, rhs)

    # ---- mailboxes: which form of the level-0 colour exchanges is faster HERE -- one exchange launch behind every colour launch, or the exchange
    # folded into the colour launch (gmgk::gs_color_push, gmg_p2p_set_smoother(2): the same iterates bit for bit, (pre + post) x colours launches
    # fewer per cycle)?  A short probe of both, decided on the MAX over ranks (the same number on every rank); the headline runs the faster one.
    exchange_form = None
    if p2p is not None and world > 1 and engine_exchange == "p2p" and not os.environ.get("GMG_BENCH_NO_FOLD_PROBE"):
        def probe(mode, k=6):
            p2p.set_smoother(mode)
            load(); run(2)
            torch.cuda.synchronize(); dist.barrier()
            tq = time.perf_counter(); r = run(k); torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([time.perf_counter() - tq], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return 1e3 * float(tt.item()) / k, r
        t_plain, r_plain = probe(0)
        t_fold, r_fold = probe(2)
        same = bool(np.array_equal(np.asarray(r_plain), np.asarray(r_fold)))
        flag = torch.tensor([int(same)], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_fold = bool(int(flag.item()) == 1 and t_fold < 0.97 * t_plain)
        p2p.set_smoother(2 if use_fold else 0)
        exchange_form = {"chosen": "folded into the colour launches" if use_fold else "one exchange launch per colour", "probe_ms_per_cycle": {"separate": t_plain, "folded": t_fold},
                         "same_residues": same}
    load()
    first = run(n_warm)
    reproduced = bool(np.allclose(first, ref_res, rtol=1e-9))
    load()
    run(args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    residues = run(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    t1 = time.perf_counter()
    tmax = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ms_per_step = 1e3 * float(tmax.item()) / args.steps
    colls_per_cycle = (dv.n_collectives / max(n_warm + args.warmup + args.steps, 1)) if dv is not None else 0

    variants = {}
    # ---- variant (never `value`): the exchange of a colour FOLDED into that colour's sweep launch (gmgk::gs_color_push, gmg_p2p_set_smoother(2)):
    # publishing waves store their rows straight into the peers' mailboxes, the last one publishes the sequence number and pulls -- no exchange
    # launch for the (pre + post) x C colour exchanges of a cycle; the same iterates.  The default is timed again right after it (same state of
    # the box), so that the two numbers compare.
    if p2p is not None and world > 1 and engine_exchange == "p2p":
        def timed(k):
            load(); run(args.warmup)
            n_before = p2p.stat("exchange_launches")
            torch.cuda.synchronize(); dist.barrier()
            tq = time.perf_counter(); r = run(k); torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([time.perf_counter() - tq], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return 1e3 * float(tt.item()) / k, (p2p.stat("exchange_launches") - n_before) / k, r
        p2p.set_smoother(2)
        load()
        folded_first = run(n_warm)
        # (ranks that share a device are scheduled against each other: a run of cycles is sometimes 2-3 x slower than the next one for no reason
        # in the code -- both forms are sampled alternately and the minimum of each is what compares)
        samples_f, samples_u, launches_f, launches_u = [], [], 0, 0
        for _ in range(4):
            p2p.set_smoother(2)
            ms, launches_f, _ = timed(args.steps); samples_f.append(ms)
            p2p.set_smoother(0)
            ms, launches_u, _ = timed(args.steps); samples_u.append(ms)
        p2p.set_smoother(2 if (exchange_form and exchange_form["chosen"].startswith("folded")) else 0)
        variants["halo_push_fold"] = {"ms_per_step": min(samples_f), "exchange_launches_per_cycle": launches_f, "samples_ms": samples_f,
                                      "default_again": {"ms_per_step": min(samples_u), "exchange_launches_per_cycle": launches_u, "samples_ms": samples_u},
                                      "residues_match_single_gpu": bool(np.allclose(folded_first, ref_res, rtol=1e-9)),
                                      "what": "level-0 colour exchanges folded into the colour sweep launches (boundary waves push from registers, the last one "
                                              "publishes the sequence number and pulls); the other exchanges of the cycle keep their launch"}

    # ---- variant (never `value`): hybrid Gauss-Seidel on level 0 -- GS inside a rank, Jacobi across ranks, ONE exchange per
    # sweep instead of one per colour: fewer, equally small messages; the iterates depend on the rank count, so the cycle count is recorded
    if p2p is not None and world > 1:
        p2p.set_smoother(True)
        load()
        hh = []
        while True:
            hh += run(1)
            if not (hh[-1] > 1e-4 and len(hh) < 100):
                break
        load(); run(args.warmup)
        torch.cuda.synchronize(); dist.barrier()
        th0 = time.perf_counter(); run(args.steps); torch.cuda.synchronize(); dist.barrier()
        th = torch.tensor([time.perf_counter() - th0], dtype=torch.float64, device="cuda")
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        p2p.set_smoother(2 if (exchange_form and exchange_form["chosen"].startswith("folded")) else 0)
        variants["hybrid_gs"] = {"ms_per_step": 1e3 * float(th.item()) / args.steps, "iterations_to_1e-4": len(hh), "residues": [float(v) for v in hh],
                                 "exchanges_per_cycle_level0": 2 + 2 + 1,
                                 "what": "level 0: Gauss-Seidel inside a rank, Jacobi across ranks, one halo exchange per sweep (gmg_p2p_set_smoother); the default exchanges after every colour"}

    # ---- variant (never `value`): the north star's collective -- an all-gather of the packed x halo after every colour sweep -- as the ENGINE runs
    # it (gmg_config::dist_exchange): pack -> all-gather -> unpack enqueued on the engine's stream, the same partition plan and kernels around it,
    # no Python between the colours.  With one device per rank the all-gather is ncclAllGather (librccl, loaded by the library); ranks that share
    # a device (GMG_DIST_BACKEND=gloo: RCCL refuses that) run the same sequence with the all-gather emulated through hipIpc mappings -- what
    # this measures there is the cost of three launches per exchange instead of one, not a link.
    if p2p is not None and world > 1 and engine_exchange == "p2p" and not os.environ.get("GMG_BENCH_NO_HALO_VARIANT"):
        mode = 1 if backend == "nccl" else 2
        okv, cyc2, why = 1, None, None
        try:
            eng_c = new_engine(partition=True, dist_exchange=mode)
            cyc2 = cabi.P2PCycle(eng_c, rank, world, 1)
        except Exception as e:          # noqa: BLE001
            okv, why = 0, repr(e)
        if agreed(okv):
            try:
                if mode == 1:
                    ids = [None] * world
                    dist.all_gather_object(ids, cabi.rccl_unique_id() if rank == 0 else None, group=cpu_group)
                    cyc2.connect_rccl(ids[0])
                else:
                    blobs2 = [None] * world
                    dist.all_gather_object(blobs2, cyc2.export(), group=cpu_group)
                    cyc2.connect(blobs2)
            except Exception as e:      # noqa: BLE001
                okv, why = 0, repr(e)
        else:
            okv = 0
        if agreed(okv):
            k = max(1, min(args.steps, 10))
            cyc2.load(rhs, rhs)
            warm = cyc2.cycles(n_warm, 2)
            torch.cuda.synchronize(); dist.barrier()
            th0 = time.perf_counter()
            cyc2.cycles(k, 2)
            torch.cuda.synchronize(); dist.barrier()
            th = torch.tensor([time.perf_counter() - th0], dtype=torch.float64, device="cuda")
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            variants["rccl_halo_allgather"] = {"ms_per_step": 1e3 * float(th.item()) / k, "steps": k,
                                               "transport": "ncclAllGather (librccl)" if mode == 1 else "all-gather emulated through hipIpc mappings (the ranks share a device)",
                                               "launches_per_exchange": 3, "exchanges_per_cycle": 25 if cyc2.stat("level1_partitioned") == 1.0 else 19,
                                               "residues_match_single_gpu": bool(np.allclose(warm, ref_res, rtol=1e-9)),
                                               "ratio_to_mailbox_path": 1e3 * float(th.item()) / k / ms_per_step,
                                               "what": "every exchange of the engine-driven cycle as pack -> all-gather -> unpack on the engine's stream (gmg_config::dist_exchange); "
                                                       "same partition (levels 0-1), same iterates"}
            del cyc2
            eng_c.close(); del eng_c
        else:
            variants["rccl_halo_allgather"] = {"ms_per_step": None, "reason": why or "another rank could not set the collective exchange up"}

    load()
    t = time.perf_counter()
    hist = []
    while True:                               # do { V-cycle; residualCheck } while (res > tol && it < maxIter)   (:1408-1419)
        hist += run(1)
        if not (hist[-1] > 1e-4 and len(hist) < 100):
            break
    iters, res = len(hist), hist[-1]
    torch.cuda.synchronize()
    solve_ms = 1e3 * (time.perf_counter() - t)

    # the reference algorithm on this host's cores (rank 0, one core, a bounded sample of the same workload: like the N = 1 line)
    cpu = None
    if rank == 0 and args.cpu_cycles > 0:          # (every N > 1 line carries it; a weak-scaling workload is N times larger: fewer cycles)
        cpu = single.cpu_baseline(H, mass, lhs, rhs, min(args.cpu_cycles, 6 if args.scaling == "strong" else 3))
    # roofline of the dominant kernel (the fine-level colour sweep, same kernel as on one GPU; measured on the whole level)
    roofline = None
    if rank == 0:
        probe = eng if p2p is None else new_engine()      # (kernel timing needs the whole level 0 and its vectors: not on the live p2p handle)
        sweep_ms, launches = probe.bench_kernel(0, 0, 1, args.kernel_reps)
        sweep_bytes = probe.algorithmic_bytes(0, 0, 1)
        achieved = sweep_bytes / (sweep_ms * 1e-3) / 1e9
        traffic, traffic_source = single.load_pmc_traffic(workload)
        roofline = {"bound": "hbm", "kernel": "gmgk::gs_color<1,1> (whole level 0 on one GPU; a rank launches 1/N of it per colour)",
                    "achieved": achieved, "peak": single.HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / single.HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_source, "launch_ms": sweep_ms / launches, "launches_per_sweep": launches}
        n0 = lhs.shape[0]
        if p2p is not None:
            if p2p.stat("level1_partitioned") == 1.0:
                lower = (f"level 1 split {world}-way by 64-row blocks, each owned by the rank holding most of its fine rows ({int(p2p.stat('level1_own_rows'))} of {levels[1]['n_pad']} rows on rank 0): "
                         f"one x1 halo exchange per block sweep ({int(p2p.stat('x1_halo_rows_published'))} rows published by rank 0), r0 halo before the restriction "
                         f"({int(p2p.stat('r0_halo_rows_published'))} rows), r1 completed on every rank once per cycle; levels >= 2 replicated")
            else:
                lower = "levels >= 1 replicated, r0 pushed to all peers once per cycle"
            if engine_exchange == "p2p":
                how = (f"per colour sweep ONE exchange kernel: each rank stores the halo entries its peers read into their mailboxes over xGMI "
                       f"({int(p2p.stat('halo_rows_published'))} rows published by rank 0) and waits for theirs")
                if exchange_form and exchange_form["chosen"].startswith("folded"):
                    how += " -- the colour halos leave INSIDE the colour launches here (gs_color_push: boundary waves store from registers, the last one publishes and pulls; chosen by a probe of both forms)"
                tail = "no collective call in the cycle"
            else:
                how = (f"per colour sweep one pack -> all-gather -> unpack sequence on the engine's stream ({int(p2p.stat('halo_rows_published'))} rows published by rank 0)")
                tail = "every exchange of the cycle is such a sequence, none is issued from Python"
            partition = f"level 0 split {world}-way by rows (sweeps, residual, prolongation, norm); {how}; {lower}; {tail}"
        elif halo is not None:
            partition = (f"level 0 split {world}-way by rows (sweeps, residual, prolongation, norm), coarse levels replicated; per colour sweep one RCCL "
                         f"all-gather of the packed halo entries of x ({halo.published_rows} rows in all), r all-gathered once per cycle")
        else:
            partition = f"level 0 colour sweeps split {world}-way by rows, everything else replicated; one RCCL all-gather of x per colour sweep"
        out = {
            "metric": "V-cycle wall time (ms per V-cycle incl. residual check) + solve-to-1e-4 iterations, 3M-vertex Poisson"
                      + (" per GPU (weak scaling)" if args.scaling == "weak" else ""),
            "value": ms_per_step, "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "n_vertices": n0, "levels": [l["n"] for l in levels], "colors": [l["n_colors"] for l in levels],
                       "smoother": f"2+2 sweeps; level 0: multicolour Gauss-Seidel over-relaxed by {eng.gs_omega:g} (row-partitioned); levels >= 1: "
                                   "block-hybrid Gauss-Seidel (replicated)",
                       "coarse_solve": args.coarse, "hipgraph": False, "partition": partition, "tolerance": 1e-4, "stopping_criteria": 2},
            "iterations_to_1e-4": iters, "residue": res, "residues_to_1e-4": [float(v) for v in hist], "solve_ms": solve_ms,
            "exchange": engine_exchange if p2p is not None else ("none (one rank)" if world == 1 else args.exchange if args.exchange != "p2p" else "halo (fallback)"),
            "exchange_note": note, "exchange_us": exchange_us, "single_gpu_residues_reproduced": reproduced,
            "exchange_fences": exchange_fences if engine_exchange == "p2p" else None,
            "level0_exchange_form": exchange_form,
            "single_gpu_solve_reproduced": bool(len(hist) == len(ref_hist) and np.allclose(hist, ref_hist, rtol=1e-9)),
            "collectives_per_cycle": colls_per_cycle, "collective_backend": backend,
            "mvertex_cycles_per_s": n0 / ms_per_step / 1e3,
            "timed_residues_tail": [float(r) for r in residues[-3:]],
            "iterations_by_smoother": {"exact_per_colour_exchange": iters, "hybrid_gs": variants.get("hybrid_gs", {}).get("iterations_to_1e-4")},
            "variants": variants,
            "host_threads_per_rank": cabi.default_host_threads(),
            # a rank's device memory: its rows of levels 0-1 + the replicated small levels + whole vectors (partitioned set-up, gmg_dist_partition)
            "device_bytes_per_rank": (p2p.stat("device_bytes") if p2p is not None else None),
            "device_bytes_per_rank_peak_during_setup": (p2p.stat("device_bytes_peak") if p2p is not None else None),
            "setup": {**setup_ms, "set_up_partitioned": bool(partitioned and p2p is not None),
                      "note": "whole_operator: the plain single-GPU set-up every rank ran first for the reference residues (and what device_bytes compares with); "
                              "partitioned: this rank's set-up of the partitioned engine -- whole A_0 / U_0 up and through the Galerkin chain (the replicated coarse "
                              "levels need every row), layouts of its own rows only, natural copies released"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    dist.barrier()
    dist.destroy_process_group()
