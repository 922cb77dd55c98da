"""Real multi-GPU smoke (needs >= 2 visible devices; skipped on the 1-GPU test boxes): `bench.py --gpus 2` launched the way
the driver launches it, with the real RCCL backend -- once with the engine-driven peer-to-peer exchange, once with the RCCL halo
all-gathers.  Both must reproduce the single-GPU iteration count and residue, and the peer-to-peer line must say that it did
not fall back.  First contact with RCCL / xGMI then happens in the test suite, not in the scaling run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("exchange", ["p2p", "halo", "engine-collective"])
def test_bench_two_gpus_with_rccl(cabi, exchange):
    """engine-collective: a rank is made to fail the mailbox set-up, all ranks move to the engine-driven cycle whose exchanges are
    pack -> ncclAllGather -> unpack on the engine's stream (librccl loaded by the library, gmg_p2p_connect_rccl)."""
    if cabi.device_count() < 2:
        pytest.skip("needs two HIP devices")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GMG_DIST_BACKEND", None)
    if exchange == "engine-collective":
        env["GMG_P2P_SELFTEST_FAIL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--n1", "600", "--n2", "600", "--exchange",
           "p2p" if exchange == "engine-collective" else exchange]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["collective_backend"] == "nccl"
    if exchange == "engine-collective":
        assert line["exchange"] == "engine-collective" and "ncclAllGather" in line["exchange_note"] and line["collectives_per_cycle"] == 0
        assert line["residue"] <= 1e-4 and line["single_gpu_residues_reproduced"] is True
    else:
        _check_line(line, exchange)


def _check_line(line, exchange):
    assert line["residue"] <= 1e-4 and 3 <= line["iterations_to_1e-4"] <= 8
    assert line["single_gpu_residues_reproduced"] is True
    if exchange == "p2p":
        assert line["exchange"] == "p2p", line.get("exchange_note")
        us = line["exchange_us"]
        assert line["collectives_per_cycle"] == 0 and us["color0"] > 0 and us["halo_all"] > 0 and us["sum_per_cycle"] > 0
        assert ("x1_halo" in us) == ("level 1 split" in line["config"]["partition"])
    else:
        assert line["collectives_per_cycle"] > 0


@pytest.mark.parametrize("exchange,shard", [("p2p", 2), ("p2p", 1), ("halo", 2)])
def test_bench_two_ranks_on_one_gpu(cabi, exchange, shard):
    """The same launch with two ranks SHARING the one GPU of the test box (gloo for the bootstrap and the fallback's collectives: RCCL
    refuses two ranks on one device): the whole `bench.py --gpus 2` path -- partition, peer-to-peer set-up through IPC handles,
    validation against the single-GPU residues, per-exchange timings, the JSON line -- runs in the suite."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GMG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--n1", "600", "--n2", "600", "--exchange", exchange,
           "--shard-levels", str(shard), "--kernel-reps", "5", "--block-lanes", "1"]      # (one lane per row: the 60 k-row level 1 takes the big levels' layout)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["collective_backend"] == "gloo" and line["scaling"] == "strong"
    _check_line(line, exchange)
    if exchange == "p2p":
        assert ("level 1 split" in line["config"]["partition"]) == (shard == 2)


def test_bench_two_ranks_on_one_gpu_at_the_benchmark_size(cabi):
    """BASELINE config 4 as BASELINE words it -- the 3 M-vertex Poisson problem, row-partitioned -- through the driver's launch line with two
    ranks on the one GPU of the test box: the same 4 cycles and residues as one GPU, level 1 partitioned, every exchange timed, and the
    round-4 fields of the N > 1 line (hybrid Gauss-Seidel variant with its own cycle count, host threads and device bytes per rank)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GMG_DIST_BACKEND="gloo", GMG_BENCH_NO_HALO_VARIANT="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--kernel-reps", "5"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["n_vertices"] == 2999824 and line["scaling"] == "strong"
    _check_line(line, "p2p")
    assert line["iterations_to_1e-4"] == 4 and "level 1 split" in line["config"]["partition"]
    hy = line["variants"]["hybrid_gs"]
    assert hy["ms_per_step"] > 0 and 4 <= hy["iterations_to_1e-4"] <= 6 and hy["residues"][-1] <= 1e-4
    assert line["iterations_by_smoother"] == {"exact_per_colour_exchange": 4, "hybrid_gs": hy["iterations_to_1e-4"]}
    assert line["host_threads_per_rank"] >= 1 and line["device_bytes_per_rank"] > 5e8


@pytest.mark.parametrize("last_resort", [False, True])
def test_a_rank_that_cannot_set_up_peer_to_peer_takes_every_rank_to_the_fallback(cabi, last_resort):
    """bench.py --gpus N decides together: if one rank fails to set the mailboxes up, all ranks run the engine-driven cycle with every exchange as
    pack -> all-gather -> unpack on the engine's stream (ncclAllGather; emulated over hipIpc for ranks that share a device, as here); if that
    cannot be set up either, the orchestration from Python (RCCL / here gloo).  The line says which."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GMG_DIST_BACKEND="gloo", GMG_P2P_SELFTEST_FAIL="1")
    if last_resort:
        env["GMG_BENCH_NO_ENGINE_COLLECTIVE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29545",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--n1", "400", "--n2", "400", "--kernel-reps", "5"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert "forced" in line["exchange_note"] or "could not set" in line["exchange_note"]
    if last_resort:
        assert line["exchange"] == "halo (fallback)" and line["collectives_per_cycle"] > 0
    else:
        assert line["exchange"] == "engine-collective" and line["collectives_per_cycle"] == 0 and "pack" in line["config"]["partition"]
    assert line["single_gpu_residues_reproduced"] is True
    assert line["residue"] <= 1e-4 and 3 <= line["iterations_to_1e-4"] <= 8
