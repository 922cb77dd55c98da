"""GPU side of the multi-GPU path on ONE MI355X.
(a) EngineBackend with world = 1 reproduces the plain engine's residual history.
(b) world = 2 / 4 emulated on one device: one engine handle per "rank" (each launching only its own row ranges
    through gmg_dist_*), run in lockstep with the all-gathers replaced by device-to-device copies of the pieces.
    The iterates must equal the single-engine ones: the partition must not change the result."""
import numpy as np
import pytest

from tests import problems

pytestmark = pytest.mark.gpu


def _engine(cabi, P, world):
    eng = cabi.Engine(row_align=64 * world, use_graph=False)
    eng.set_prolongations(P.U); eng.set_mass(P.mass); eng.set_system(P.lhs)
    return eng


@pytest.mark.parametrize("kind", ["poisson", "smoothing"])
def test_world1_backend_matches_plain_engine(cabi, kind):
    import torch
    from gravo_mg_amd.dist import DistVCycle, EngineBackend
    P = problems.torus_problem(96, 80, "poisson", 30) if kind == "poisson" else problems.torus_problem(64, 60, "smoothing", 60)
    ref = cabi.Engine(use_graph=False)
    ref.set_prolongations(P.U); ref.set_mass(P.mass); ref.set_system(P.lhs)
    ref.load_problem(P.rhs, P.rhs)
    want = ref.run_cycles(4, 2)
    eng = _engine(cabi, P, 1)
    be = EngineBackend(eng, P.rhs.shape[1], 0, 1)
    dv = DistVCycle(be)
    be.load(P.rhs, P.rhs)
    got = []
    for _ in range(4):
        dv.vcycle()
        got.append(dv.residual_norm(2))
    np.testing.assert_allclose(got, want, rtol=1e-9)
    np.testing.assert_allclose(be.fetch(), ref.fetch_solution(), rtol=1e-9, atol=1e-12 * np.abs(P.rhs).max())
    it, res, _ = dv.solve(1e-4, 2, 50)
    assert res <= 1e-4


@pytest.mark.parametrize("world", [2, 4])
def test_emulated_ranks_on_one_device(cabi, world):
    import torch
    from gravo_mg_amd.dist import EngineBackend
    P = problems.torus_problem(96, 80, "poisson", 30)
    d = P.rhs.shape[1]
    # single-rank reference with the SAME padded layout (row_align = 64*world), one handle doing everything
    ref_eng = _engine(cabi, P, world)
    ref = EngineBackend(ref_eng, d, 0, 1)          # world=1 on a 64*world-aligned layout is valid
    bes = [EngineBackend(_engine(cabi, P, world), d, r, world) for r in range(world)]
    cb, n_pad, C = bes[0].color_begin, bes[0].n_pad, len(bes[0].color_begin) - 1
    for b in bes + [ref]:
        b.load(P.rhs, P.rhs)
    torch.cuda.synchronize()

    def exchange(name, c):
        torch.cuda.synchronize()
        lo, hi = cb[c], cb[c + 1]
        piece = (hi - lo) // world
        for col in range(d):
            for src in range(world):
                s = slice(col * n_pad + lo + src * piece, col * n_pad + lo + (src + 1) * piece)
                for dst in range(world):
                    if dst != src:
                        getattr(bes[dst], name)[s].copy_(getattr(bes[src], name)[s])
        torch.cuda.synchronize()

    def cycle_ranks():
        for _ in range(2):
            for c in range(C):
                for b in bes: b.smooth_color(c)
                exchange("x", c)
        for b in bes: b.residual_own()
        for c in range(C): exchange("r", c)
        for b in bes: b.coarse_cycle()
        for b in bes: b.prolong_own()
        for c in range(C): exchange("x", c)
        for _ in range(2):
            for c in range(C):
                for b in bes: b.smooth_color(c)
                exchange("x", c)

    def cycle_ref():
        for _ in range(2):
            for c in range(C): ref.smooth_color(c)
        ref.residual_own(); ref.coarse_cycle(); ref.prolong_own()
        for _ in range(2):
            for c in range(C): ref.smooth_color(c)

    for _ in range(3):
        cycle_ranks(); cycle_ref()
        torch.cuda.synchronize()
        for b in bes:
            assert torch.equal(b.x, ref.x)          # bitwise: same kernels, same row-local arithmetic
        sums = sum(b.norm_partial(2) for b in bes)
        want = ref.norm_partial(2)
        np.testing.assert_allclose(sums, want, rtol=1e-12)


@pytest.mark.parametrize("world,kind", [(2, "poisson"), (4, "poisson"), (3, "smoothing")])
def test_emulated_halo_exchange_on_one_device(cabi, world, kind):
    """Halo mode (dist.HaloPlan + gmg_dist_gather / gmg_dist_scatter): after a colour sweep a rank publishes only the
    entries other ranks read.  Ranks emulated on one device, the all-gather of the packed buffers replaced by a
    concatenation.  Own rows must equal the single-engine iterate bitwise; after completing x, everything does."""
    import scipy.sparse as sp
    import torch
    from gravo_mg_amd.dist import EngineBackend, HaloPlan, row_owner
    P = problems.torus_problem(96, 80, "poisson", 30) if kind == "poisson" else problems.torus_problem(64, 60, "smoothing", 60)
    d = P.rhs.shape[1]
    ref = EngineBackend(_engine(cabi, P, world), d, 0, 1)
    bes = [EngineBackend(_engine(cabi, P, world), d, r, world) for r in range(world)]
    cb, n_pad, C = bes[0].color_begin, bes[0].n_pad, len(bes[0].color_begin) - 1
    new2old, _ = bes[0].eng.level_ordering(0)
    A = sp.csr_matrix(P.lhs)
    dev = bes[0].x.device
    plans = [HaloPlan(A.indptr, A.indices, new2old, cb, n_pad, world, r, d, device=dev) for r in range(world)]
    assert 0 < plans[0].published_rows < 0.5 * P.lhs.shape[0]
    owner, _ = row_owner(cb, n_pad, world)
    own = [torch.as_tensor(np.concatenate([np.nonzero(owner == r)[0] + k * n_pad for k in range(d)]), device=dev) for r in range(world)]
    for b in bes + [ref]:
        b.load(P.rhs, P.rhs)
    torch.cuda.synchronize()

    def halo(name, key):
        sends = []
        for r, b in enumerate(bes):
            si, sp_, _, _ = plans[r].tensors(key)
            send = torch.zeros(plans[r].maxlen[key] * d, dtype=torch.float64, device=dev)
            with b.stream_context():
                b.halo_gather(getattr(b, name), si, sp_, send)
            sends.append(send)
        torch.cuda.synchronize()
        recv = torch.cat(sends)
        for r, b in enumerate(bes):
            _, _, rp, ri = plans[r].tensors(key)
            with b.stream_context():
                b.halo_scatter(recv, rp, ri, getattr(b, name))
        torch.cuda.synchronize()

    def full(name):
        torch.cuda.synchronize()
        for src in range(world):
            for dst in range(world):
                if dst != src:
                    getattr(bes[dst], name)[own[src]] = getattr(bes[src], name)[own[src]]
        torch.cuda.synchronize()

    def cycle_ranks():
        for _ in range(2):
            for c in range(C):
                for b in bes: b.smooth_color(c)
                halo("x", c)
        for b in bes: b.residual_own()
        full("r")
        for b in bes: b.coarse_cycle()
        for b in bes: b.prolong_own()
        halo("x", "all")
        for _ in range(2):
            for c in range(C):
                for b in bes: b.smooth_color(c)
                halo("x", c)

    def cycle_ref():
        for _ in range(2):
            for c in range(C): ref.smooth_color(c)
        ref.residual_own(); ref.coarse_cycle(); ref.prolong_own()
        for _ in range(2):
            for c in range(C): ref.smooth_color(c)

    for _ in range(3):
        cycle_ranks(); cycle_ref()
        torch.cuda.synchronize()
        for r, b in enumerate(bes):
            assert torch.equal(b.x[own[r]], ref.x[own[r]])
        sums = sum(b.norm_partial(2) for b in bes)
        np.testing.assert_allclose(sums, ref.norm_partial(2), rtol=1e-12)
    full("x")
    for b in bes:
        assert torch.equal(b.x, ref.x)
