"""Multi-GPU V-cycle: one process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI).

Partition (BASELINE.json north_star): 1-D row split of the finest level.  The device numbering
of level 0 is colour-major; every colour class is padded to 64*P rows and cut into P equal contiguous pieces,
rank p owns piece p of every colour.  Levels >= 1 are small (<= n_0/6) and are REPLICATED: every rank runs them
redundantly, which needs no communication at all.

Exchange steps per V-cycle (each an in-place all-gather of one colour's segment of x; a rank's input is its own
piece of the output buffer): one after every colour of every Gauss-Seidel sweep (the next colour reads it) --
16 per cycle on a 4-colour mesh with 2+2 sweeps.  After such an exchange every rank holds the complete x, so the
cheap steps around the sweeps (residual, prolongation-add, residual norm) are computed REDUNDANTLY on all rows
instead of being exchanged (`replicate=True`, the default; `replicate=False` keeps them row-partitioned with an
all-gather of r / x and an all-reduce of the norm sums: 24 collectives + 1 per cycle).  Because the colours are
GLOBAL, the distributed sweep is the same multicolour Gauss-Seidel as on one GPU: results are independent of P.

`halo=HaloPlan(...)` (what bench.py uses for N > 1) replaces the whole-segment all-gathers by a HALO exchange
(SURVEY.md 8e "v2"): after a colour sweep a rank publishes only those of its entries of x that rows of OTHER ranks
read -- O(boundary) values instead of n/(C*P) -- packed into a fixed-size buffer, all-gathered, and unpacked by the
receivers (pack + collective + unpack, all d columns at once).  x is then complete on a rank only on its own rows and
their halo, so residual, prolongation and norm are row-partitioned: one all-gather of r per colour segment before the
replicated coarse part (24 MB per cycle at 3 M vertices instead of 16 x 6 MB), one halo exchange of x after the
prolongation, one all-reduce of the norm sums.  The iterates are still those of the global multicolour sweep
(independent of P); the norm sums are added across ranks, so residues agree to rounding.  `gather_solution()`
completes x on every rank (solve() calls it once at the end).

The local work is done by a *backend*: `EngineBackend` launches this rank's share on its GPU through the C-ABI
(gmg_dist_* in include/gravomg_hip.h).  tests/test_dist_gloo.py plugs in a numpy backend to check the
orchestration (partition arithmetic, order of exchanges) with the gloo backend on CPU.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def row_owner(color_begin, n_pad: int, world: int) -> np.ndarray:
    """Owner rank of every device row of level 0: piece p of every (64*world-aligned) colour class."""
    cb = np.asarray(color_begin, np.int64)
    rows = np.arange(n_pad, dtype=np.int64)
    col = np.searchsorted(cb, rows, side="right") - 1
    piece = np.maximum((cb[1:] - cb[:-1]) // world, 1)
    return ((rows - cb[col]) // piece[col]).astype(np.int32), col.astype(np.int32)


class HaloPlan:
    """Index lists of the halo exchange, the same on every rank (built from the replicated matrix pattern).

    indptr/indices: pattern of the level-0 operator (natural numbering, CSR or CSC -- it is symmetric);
    new2old/color_begin/n_pad: the device ordering of level 0 (Engine.level_ordering(0) / host_plan_level).
    For key c (a colour) or "all": rank q publishes pub[q][key] (device rows it owns that another rank's rows read);
    the send buffer holds d * maxlen[key] values (column k at k * maxlen), the receive buffer world such pieces."""

    def __init__(self, indptr, indices, new2old, color_begin, n_pad: int, world: int, rank: int, d: int, device=None):
        new2old = np.asarray(new2old, np.int64)
        n = int(len(indptr) - 1)
        real = new2old >= 0
        old2new = np.empty(n, np.int64)
        old2new[new2old[real]] = np.nonzero(real)[0]
        owner, color = row_owner(color_begin, n_pad, world)
        ri = np.repeat(old2new, np.diff(np.asarray(indptr, np.int64)))      # device row of every entry
        ci = old2new[np.asarray(indices, np.int64)]
        read_elsewhere = np.zeros(n_pad, bool)
        read_elsewhere[ci[owner[ri] != owner[ci]]] = True
        need = np.nonzero(read_elsewhere)[0]                                  # sorted device rows
        self.world, self.rank, self.d, self.n_pad = world, rank, d, n_pad
        self.n_colors = len(color_begin) - 1
        self.keys = list(range(self.n_colors)) + ["all"]
        self.maxlen, self.n_send, self.n_recv = {}, {}, {}
        self._t = {}
        dev = device
        for key in self.keys:
            sel = need if key == "all" else need[color[need] == key]
            pub = [sel[owner[sel] == q] for q in range(world)]
            m = max(1, max(len(p_) for p_ in pub))
            cols = np.arange(d, dtype=np.int64)
            mine = pub[rank]
            send_idx = (mine[None, :] + cols[:, None] * n_pad).ravel()
            send_pos = (np.arange(len(mine), dtype=np.int64)[None, :] + cols[:, None] * m).ravel()
            rp, rx = [], []
            for q in range(world):
                if q == rank or len(pub[q]) == 0:
                    continue
                rp.append((q * d * m + np.arange(len(pub[q]), dtype=np.int64)[None, :] + cols[:, None] * m).ravel())
                rx.append((pub[q][None, :] + cols[:, None] * n_pad).ravel())
            recv_pos = np.concatenate(rp) if rp else np.zeros(0, np.int64)
            recv_idx = np.concatenate(rx) if rx else np.zeros(0, np.int64)
            self.maxlen[key], self.n_send[key], self.n_recv[key] = m, len(send_idx), len(recv_idx)
            self._t[key] = tuple(torch.as_tensor(a, device=dev) for a in (send_idx, send_pos, recv_pos, recv_idx))
        self.published_rows = int(len(need))

    def tensors(self, key):
        return self._t[key]


class DistVCycle:
    """V-cycle + residual check + solve loop over `world` ranks.  `backend` provides the local steps and the
    level-0 vectors x, b, r as flat torch tensors of length d * n_pad (column-major)."""

    def __init__(self, backend, group=None, replicate: bool = True, halo: Optional[HaloPlan] = None):
        self.be = backend
        self.halo = halo
        self.replicate = bool(replicate) and halo is None        # a halo exchange leaves x incomplete: row-partitioned steps
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.color_begin = [int(v) for v in backend.color_begin]
        self.n_pad = int(backend.n_pad)
        self.d = int(backend.d)
        self.n_colors = len(self.color_begin) - 1
        for c in range(self.n_colors):
            assert (self.color_begin[c + 1] - self.color_begin[c]) % (64 * self.world) == 0, "colour classes must be 64*world aligned"
        self.n_collectives = 0
        self._send = {}

    # -- exchange ------------------------------------------------------------------------------------------
    def _allgather_color(self, buf: torch.Tensor, c: int):
        if self.world == 1:
            return
        lo, hi = self.color_begin[c], self.color_begin[c + 1]
        piece = (hi - lo) // self.world
        if piece == 0:
            return
        for col in range(self.d):
            seg = buf[col * self.n_pad + lo: col * self.n_pad + hi]
            mine = seg[self.rank * piece: (self.rank + 1) * piece]
            # the send buffer is a copy of this rank's piece (a few MB, device to device): no reliance on a backend's
            # handling of an input that aliases the output
            send = self._send.get(piece)
            if send is None or send.device != mine.device or send.dtype != mine.dtype:
                send = self._send[piece] = torch.empty_like(mine)
            send.copy_(mine)
            try:
                dist.all_gather_into_tensor(seg, send, group=self.group)
            except (RuntimeError, NotImplementedError):          # backends without the flat variant (old gloo)
                dist.all_gather(list(seg.chunk(self.world)), send, group=self.group)
            self.n_collectives += 1

    def _allgather_all(self, buf: torch.Tensor):
        for c in range(self.n_colors):
            self._allgather_color(buf, c)

    def _halo_exchange(self, buf: torch.Tensor, key):
        """Publish this rank's entries of `buf` that other ranks read (colour `key`, or every colour for "all")."""
        if self.world == 1:
            return
        hp = self.halo
        send_idx, send_pos, recv_pos, recv_idx = hp.tensors(key)
        m = hp.maxlen[key] * self.d
        bufs = self._send.get(("halo", key))
        if bufs is None or bufs[0].device != buf.device:
            bufs = self._send[("halo", key)] = (torch.zeros(m, dtype=buf.dtype, device=buf.device),
                                                torch.zeros(m * self.world, dtype=buf.dtype, device=buf.device))
        send, recv = bufs
        gather = getattr(self.be, "halo_gather", None)
        if gather is not None:
            gather(buf, send_idx, send_pos, send)
        elif send_idx.numel():
            send.index_copy_(0, send_pos, buf.index_select(0, send_idx))
        try:
            dist.all_gather_into_tensor(recv, send, group=self.group)
        except (RuntimeError, NotImplementedError):
            dist.all_gather(list(recv.chunk(self.world)), send, group=self.group)
        self.n_collectives += 1
        scatter = getattr(self.be, "halo_scatter", None)
        if scatter is not None:
            scatter(recv, recv_pos, recv_idx, buf)
        elif recv_idx.numel():
            buf.index_copy_(0, recv_idx, recv.index_select(0, recv_pos))

    def gather_solution(self):
        """Complete x on every rank (needed after halo-mode cycles before the solution is read)."""
        if self.halo is not None and self.world > 1:
            with self.be.stream_context():
                self._allgather_all(self.be.x)

    # -- the cycle (gravomg/src/multigrid_solver.cpp:1059-1088) ----------------------------------------------
    def smooth(self, iters: int):
        for _ in range(iters):
            for c in range(self.n_colors):
                self.be.smooth_color(c)
                if self.halo is not None:
                    self._halo_exchange(self.be.x, c)
                else:
                    self._allgather_color(self.be.x, c)

    def vcycle(self):
        with self.be.stream_context():              # kernels and collectives ordered on the backend's stream
            self.smooth(self.be.pre_iters)              # :1063
            if self.replicate:
                self.be.residual_all()                  # :1066, all rows on every rank (x is complete)
            else:
                self.be.residual_own()
                self._allgather_all(self.be.r)
            self.be.coarse_cycle()                      # :1069-1079 (replicated)
            if self.replicate:
                self.be.prolong_all()                   # :1082
            elif self.halo is not None:
                self.be.prolong_own()
                self._halo_exchange(self.be.x, "all")
            else:
                self.be.prolong_own()
                self._allgather_all(self.be.x)
            self.smooth(self.be.post_iters)             # :1085

    # -- residualCheck (:1228-1277) ---------------------------------------------------------------------------
    def residual_norm(self, type: int = 2) -> float:
        sums = torch.as_tensor(np.asarray(self.be.norm_all(type) if self.replicate else self.be.norm_partial(type), dtype=np.float64))
        if self.world > 1 and not self.replicate:
            with self.be.stream_context():
                t = sums.to(self.be.x.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                sums = t.cpu()
            self.n_collectives += 1
        s = sums.numpy()
        if type == 3:
            return math.sqrt(float(s[0::2].sum()))
        vals = [(math.sqrt(s[2 * c]) / math.sqrt(s[2 * c + 1])) if type == 0 else math.sqrt(s[2 * c] / s[2 * c + 1]) for c in range(self.d)]
        return max(vals)

    # -- solve loop (:1408-1419) --------------------------------------------------------------------------------
    def solve(self, tol: float = 1e-4, stop_type: int = 2, max_iter: int = 100):
        it, residues = 0, []
        while True:
            self.vcycle()
            res = self.residual_norm(stop_type)
            residues.append(res)
            it += 1
            if not (res > tol and it < max_iter):
                break
        self.gather_solution()
        return it, res, residues


class EngineBackend:
    """This rank's share of level 0 + the replicated coarse levels on one MI355X, through the C-ABI.
    All launches and the RCCL collectives are ordered on one dedicated torch stream."""

    def __init__(self, engine, d: int, rank: int, world: int, device: Optional[torch.device] = None):
        self.eng = engine
        self.d = int(d)
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        info = engine.level_info(0)
        self.n_pad = info["n_pad"]
        _, cb = engine.level_ordering(0)
        self.color_begin = [int(v) for v in cb]
        self.pre_iters, self.post_iters = engine.pre_iters, engine.post_iters
        self.stream = torch.cuda.Stream(device=self.device)
        self.x = torch.zeros(self.n_pad * self.d, dtype=torch.float64, device=self.device)
        self.b = torch.zeros_like(self.x)
        self.r = torch.zeros_like(self.x)
        torch.cuda.synchronize(self.device)
        engine.set_stream(self.stream.cuda_stream)
        engine.dist_setup(rank, world)
        engine.dist_bind(self.x.data_ptr(), self.b.data_ptr(), self.r.data_ptr(), self.d)

    def stream_context(self):
        return torch.cuda.stream(self.stream)

    def load(self, b_host, x0_host):
        """Host rhs / initial guess (natural numbering, n x d) -> the bound device vectors (every rank loads all rows)."""
        self.eng.load_problem(b_host, x0_host)

    def fetch(self):
        return self.eng.fetch_solution()

    def smooth_color(self, c):
        self.eng.dist_smooth_color(c)

    def residual_own(self):
        self.eng.dist_residual_own()

    def coarse_cycle(self):
        self.eng.dist_coarse_cycle()

    def prolong_own(self):
        self.eng.dist_prolong_own()

    def norm_partial(self, type):
        return self.eng.dist_norm_partial(type, self.d)

    def residual_all(self):
        self.eng.dist_residual_all()

    def prolong_all(self):
        self.eng.dist_prolong_all()

    def norm_all(self, type):
        return self.eng.dist_norm_all(type, self.d)

    # pack / unpack of a halo exchange: one launch each on the engine stream (gmg_dist_gather / gmg_dist_scatter)
    def halo_gather(self, buf, send_idx, send_pos, send):
        if self.d == 1:                         # column 0 only: positions are 0..n-1
            self.eng.dist_gather(buf.data_ptr(), send_idx.data_ptr(), send_idx.numel(), send.data_ptr())
        else:
            self.eng.dist_scatter(buf.data_ptr(), send_idx.data_ptr(), send_pos.data_ptr(), send_idx.numel(), send.data_ptr())

    def halo_scatter(self, recv, recv_pos, recv_idx, buf):
        self.eng.dist_scatter(recv.data_ptr(), recv_pos.data_ptr(), recv_idx.data_ptr(), recv_idx.numel(), buf.data_ptr())
