// engine_dist.hip.hpp -- part of libgravomg_hip.so's single translation unit (included by engine.hip after engine_cycle).
// Multi-GPU V-cycle driven by the ENGINE (one process per GPU): level 0 row-partitioned per colour, levels >= 1 replicated,
// and every exchange a device-initiated store into the peer's mailbox (kernels.hip.hpp::p2p_exchange) -- no collective-library
// call and no Python between the launches of a cycle.  gravo_mg_amd/dist.py keeps the RCCL orchestration of the same cycle
// (torch.distributed all-gathers), which remains the fallback and the reference for the tests.
//
// Partition (BASELINE.json north_star): the device numbering of level 0 is colour-major, every colour class is
// padded to 64 * world rows and cut into `world` equal contiguous pieces; rank p owns piece p of every colour.  What moves:
//   * after every colour of every sweep: the entries of x this rank just updated that rows of ANOTHER rank read (the halo of
//     that colour; a few thousand values at 3 M vertices) -- per peer, only what that peer reads;
//   * once per cycle: this rank's rows of the residual, to every peer (the replicated coarse part restricts the whole r);
//   * after the prolongation: the halo of all colours at once;
//   * per residual check: 2 d partial sums, added in rank order on every rank (same bits everywhere).
// The iterates are those of the single-GPU multicolour sweep: bitwise independent of the number of ranks.
#pragma once

struct P2PPeer {
    int rank = -1;
    void *mbox_base = nullptr, *flag_base = nullptr;      // the peer's mailbox / counters mapped here (hipIpcOpenMemHandle)
};

struct gmg_p2p_blob {                                     // what a rank publishes to the others (plain bytes, exchanged out of band)
    hipIpcMemHandle_t mbox, flags;
    int rank, world, d, n_pad, n_colors, reserved;
    long long mbox_doubles;
};

struct DistP2P {
    bool planned = false, connected = false;
    int rank = 0, world = 1, d = 0, nk = 0;               // nk = exchange kinds: colours 0..C-1, C = halo of all colours, C+1 = residual rows, C+2 = norm sums
    // plan: rows (device numbering of level 0) rank s publishes to rank t for halo kind k, ascending
    std::vector<std::vector<int>> halo;                   // [(s * world + t) * (C + 1) + k]
    std::vector<int> own_lo, own_cnt;                     // [colour]: this partition's piece of a colour (same count on every rank)
    // mailbox layout of EVERY rank (all ranks compute the same table): offset (doubles) of region (src, kind, parity) in dst's mailbox
    std::vector<long long> box_off;                       // [((dst * world + src) * nk + kind) * 2 + parity]
    std::vector<long long> box_total;                     // [dst]
    double* mbox = nullptr;                               // local mailbox (fine-grained device memory)
    unsigned long long* flags = nullptr;                  // local arrival counters, one per source rank
    std::vector<P2PPeer> peers;                           // world - 1 entries, ascending rank
    int* d_idx = nullptr;                                 // all index lists, concatenated
    gmgk::P2POp* d_ops = nullptr;                         // [(kind * 2 + parity) * n_peers + j]
    int* d_err = nullptr;
    double* d_sums = nullptr;                             // 2 d partial sums of this rank / the reduced sums
    std::vector<unsigned long long> kind_count;           // exchanges done per kind (parity = count & 1)
    unsigned long long seq = 0;                           // exchanges done in all (the arrival counters carry it)
    std::map<std::string, double> stats;
};

namespace {

inline int p2p_owner(const LevelOrdering& o, int world, int row, int* colour_out) {
    int c = (int)(std::upper_bound(o.color_begin.begin(), o.color_begin.end(), row) - o.color_begin.begin()) - 1;
    const int piece = (o.color_begin[c + 1] - o.color_begin[c]) / world;
    if (colour_out) *colour_out = c;
    return piece > 0 ? (row - o.color_begin[c]) / piece : 0;
}

void p2p_release(gmg_handle h) {
    DistP2P* p = h->p2p;
    if (!p) return;
    for (auto& peer : p->peers) {
        if (peer.mbox_base) (void)hipIpcCloseMemHandle(peer.mbox_base);
        if (peer.flag_base) (void)hipIpcCloseMemHandle(peer.flag_base);
    }
    if (p->mbox) (void)hipFree(p->mbox);
    if (p->flags) (void)hipFree(p->flags);
    if (p->d_idx) (void)hipFree(p->d_idx);
    if (p->d_ops) (void)hipFree(p->d_ops);
    if (p->d_err) (void)hipFree(p->d_err);
    if (p->d_sums) (void)hipFree(p->d_sums);
    delete p;
    h->p2p = nullptr;
}

// One exchange of kind `kind` on vector `vec` (level-0 layout) -- or nothing with a single rank.
int p2p_exchange(gmg_handle h, int kind, double* vec) {
    DistP2P* p = h->p2p;
    const int np = (int)p->peers.size();
    if (np == 0) return GMG_OK;
    const int parity = (int)(p->kind_count[kind]++ & 1);
    ++p->seq;
    hipLaunchKernelGGL(gmgk::p2p_exchange, dim3(2 * np), dim3(256), 0, h->stream, p->d_ops + (size_t)(kind * 2 + parity) * np, np, vec, h->lv[0].n_pad, p->d,
                       p->seq, p->d_err);
    return GMG_OK;
}

}  // namespace

void p2p_release_handle(gmg_handle h) { p2p_release(h); }

extern "C" {

int gmg_p2p_blob_bytes(void) { return (int)sizeof(gmg_p2p_blob); }

// Plan + local allocations.  The system must be set on a handle created with row_align = 64 * world (like gmg_dist_setup).
int gmg_p2p_prepare(gmg_handle h, int rank, int world, int d) try {
    NEED_DEVICE();
    int rc = check_level(h, 0, false);
    if (rc) return rc;
    if (world < 1 || rank < 0 || rank >= world || d < 1 || d > 4) return fail(h, GMG_ERR_INVALID, "bad rank / world size / column count (d <= 4)");
    if ((rc = gmg_dist_setup(h, rank, world))) return rc;
    p2p_release(h);
    if ((rc = ensure_vectors(h, d))) return rc;
    unbind_level0(h);
    DistP2P* p = h->p2p = new DistP2P();
    p->rank = rank; p->world = world; p->d = d;
    Level& l = h->lv[0];
    const LevelOrdering& o = l.ord;
    const int C = o.n_colors;
    p->nk = C + 3;
    p->kind_count.assign(p->nk, 0);
    p->own_lo.resize(C); p->own_cnt.resize(C);
    for (int c = 0; c < C; ++c) { p->own_cnt[c] = (o.color_begin[c + 1] - o.color_begin[c]) / world; p->own_lo[c] = o.color_begin[c] + rank * p->own_cnt[c]; }
    // ---- who reads what: for every row r (owner t) and every entry (r, c) with owner(c) = s != t, s publishes c to t
    p->halo.assign((size_t)world * world * (C + 1), std::vector<int>());
    if (world > 1) {
        if ((rc = ensure_host_A(h, 0, false))) return rc;
        const Compressed& A = l.A;
        std::vector<int> owner(l.n_pad), colour(l.n_pad);
        parallel_ranges(l.n_pad, h->cfg.host_threads, [&](int lo, int hi, int) { for (int r = lo; r < hi; ++r) owner[r] = p2p_owner(o, world, r, &colour[r]); });
        const int T = std::max(1, std::min(h->cfg.host_threads, 16));
        std::vector<std::vector<std::vector<int>>> part(T, std::vector<std::vector<int>>((size_t)world * world));
        parallel_ranges(l.n, T, [&](int lo, int hi, int t) {
            auto& mine = part[std::min(t, T - 1)];
            for (int i = lo; i < hi; ++i) {
                const int r = o.old2new[i], tr = owner[r];
                for (int q = A.ptr[i]; q < A.ptr[i + 1]; ++q) {
                    const int c = o.old2new[A.idx[q]], s = owner[c];
                    if (s != tr) mine[(size_t)s * world + tr].push_back(c);
                }
            }
        }, 1);
        for (int s = 0; s < world; ++s)
            for (int t = 0; t < world; ++t) {
                if (s == t) continue;
                std::vector<int> all;
                for (int w = 0; w < T; ++w) { auto& v = part[w][(size_t)s * world + t]; all.insert(all.end(), v.begin(), v.end()); }
                std::sort(all.begin(), all.end());
                all.erase(std::unique(all.begin(), all.end()), all.end());
                for (int c : all) p->halo[((size_t)s * world + t) * (C + 1) + colour[c]].push_back(c);
                p->halo[((size_t)s * world + t) * (C + 1) + C] = all;
            }
    }
    // ---- mailbox layout of every rank: region (src, kind, parity) in dst's mailbox
    const long long own_rows = (long long)l.n_pad / world;
    p->box_off.assign((size_t)world * world * p->nk * 2, 0);
    p->box_total.assign(world, 0);
    for (int dst = 0; dst < world; ++dst) {
        long long off = 0;
        for (int src = 0; src < world; ++src)
            for (int k = 0; k < p->nk; ++k)
                for (int par = 0; par < 2; ++par) {
                    long long cnt;
                    if (k <= C) cnt = src == dst ? 0 : (long long)p->halo[((size_t)src * world + dst) * (C + 1) + k].size() * d;
                    else if (k == C + 1) cnt = src == dst ? 0 : own_rows * d;
                    else cnt = 2LL * d;                                   // norm sums: a slot for every source, the own one included
                    p->box_off[(((size_t)dst * world + src) * p->nk + k) * 2 + par] = off;
                    off += (cnt + 7) / 8 * 8;                             // 64-byte aligned regions
                }
        p->box_total[dst] = std::max<long long>(off, 8);
    }
    HIPCHK(hipExtMallocWithFlags((void**)&p->mbox, sizeof(double) * (size_t)p->box_total[rank], hipDeviceMallocFinegrained));
    HIPCHK(hipExtMallocWithFlags((void**)&p->flags, sizeof(unsigned long long) * 64 * (size_t)world, hipDeviceMallocFinegrained));
    HIPCHK(hipMemsetAsync(p->mbox, 0, sizeof(double) * (size_t)p->box_total[rank], h->stream));
    HIPCHK(hipMemsetAsync(p->flags, 0, sizeof(unsigned long long) * 64 * (size_t)world, h->stream));
    HIPCHK(hipMalloc((void**)&p->d_err, sizeof(int)));
    HIPCHK(hipMemsetAsync(p->d_err, 0, sizeof(int), h->stream));
    HIPCHK(hipMalloc((void**)&p->d_sums, sizeof(double) * 4 * d));
    HIPCHK(hipStreamSynchronize(h->stream));
    p->planned = true;
    p->stats["halo_rows_published"] = 0;
    for (int t = 0; t < world; ++t) if (t != rank) p->stats["halo_rows_published"] += (double)p->halo[((size_t)rank * world + t) * (C + 1) + C].size();
    return GMG_OK;
} GMG_CATCH_H

int gmg_p2p_export(gmg_handle h, void* blob_out) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->planned || !blob_out) return fail(h, GMG_ERR_STATE, "call gmg_p2p_prepare first");
    gmg_p2p_blob b;
    std::memset(&b, 0, sizeof(b));
    HIPCHK(hipIpcGetMemHandle(&b.mbox, p->mbox));
    HIPCHK(hipIpcGetMemHandle(&b.flags, p->flags));
    b.rank = p->rank; b.world = p->world; b.d = p->d; b.n_pad = h->lv[0].n_pad; b.n_colors = h->lv[0].ord.n_colors; b.mbox_doubles = p->box_total[p->rank];
    std::memcpy(blob_out, &b, sizeof(b));
    return GMG_OK;
} GMG_CATCH_H

// blobs: `world` blobs in rank order (every rank's gmg_p2p_export output, gathered by the caller -- e.g. torch.distributed
// all_gather_object, MPI, a file).  (Ranks must be separate PROCESSES: a process's streams share a few hardware queues, and an
// exchange kernel waiting in front of the kernel it waits for would never be served.)
int gmg_p2p_connect(gmg_handle h, const void* blobs) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->planned || !blobs) return fail(h, GMG_ERR_STATE, "call gmg_p2p_prepare first");
    const int world = p->world, rank = p->rank, C = h->lv[0].ord.n_colors, nk = p->nk, d = p->d;
    const gmg_p2p_blob* bl = (const gmg_p2p_blob*)blobs;
    p->peers.clear();
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        P2PPeer peer;
        peer.rank = q;
        if (bl[q].rank != q || bl[q].world != world || bl[q].d != d || bl[q].n_pad != h->lv[0].n_pad || bl[q].n_colors != C || bl[q].mbox_doubles != p->box_total[q])
            return fail(h, GMG_ERR_INVALID, "peer " + std::to_string(q) + " published a different partition plan (different system / ordering?)");
        HIPCHK(hipIpcOpenMemHandle(&peer.mbox_base, bl[q].mbox, hipIpcMemLazyEnablePeerAccess));
        HIPCHK(hipIpcOpenMemHandle(&peer.flag_base, bl[q].flags, hipIpcMemLazyEnablePeerAccess));
        p->peers.push_back(peer);
    }
    const int np = (int)p->peers.size();
    if (np == 0) { p->connected = true; return GMG_OK; }
    // ---- index lists on the device + one op table per (kind, parity)
    std::vector<int> idx;
    std::vector<size_t> send_at((size_t)np * (C + 1)), recv_at((size_t)np * (C + 1));
    for (int j = 0; j < np; ++j)
        for (int k = 0; k <= C; ++k) {
            const auto& sl = p->halo[((size_t)rank * world + p->peers[j].rank) * (C + 1) + k];
            const auto& rl = p->halo[((size_t)p->peers[j].rank * world + rank) * (C + 1) + k];
            send_at[(size_t)j * (C + 1) + k] = idx.size(); idx.insert(idx.end(), sl.begin(), sl.end());
            recv_at[(size_t)j * (C + 1) + k] = idx.size(); idx.insert(idx.end(), rl.begin(), rl.end());
        }
    if (p->d_idx) { (void)hipFree(p->d_idx); p->d_idx = nullptr; }
    HIPCHK(hipMalloc((void**)&p->d_idx, sizeof(int) * std::max<size_t>(idx.size(), 1)));
    if (!idx.empty()) HIPCHK(hipMemcpy(p->d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    std::vector<gmgk::P2POp> ops((size_t)nk * 2 * np);
    const int own_rows = h->lv[0].n_pad / world;
    for (int k = 0; k < nk; ++k)
        for (int par = 0; par < 2; ++par)
            for (int j = 0; j < np; ++j) {
                const int q = p->peers[j].rank;
                gmgk::P2POp& op = ops[(size_t)(k * 2 + par) * np + j];
                double* qbox = (double*)p->peers[j].mbox_base;
                unsigned long long* qflags = (unsigned long long*)p->peers[j].flag_base;
                op.remote_box = qbox + p->box_off[(((size_t)q * world + rank) * nk + k) * 2 + par];
                op.remote_flag = qflags + 64 * (size_t)rank;                 // one cache line per source rank
                op.local_box = p->mbox + p->box_off[(((size_t)rank * world + q) * nk + k) * 2 + par];
                op.local_flag = p->flags + 64 * (size_t)q;
                op.send_idx = op.recv_idx = nullptr; op.n_send = op.n_recv = 0; op.send_lo = op.recv_lo = 0;
                if (k <= C) {
                    op.n_send = (int)p->halo[((size_t)rank * world + q) * (C + 1) + k].size();
                    op.n_recv = (int)p->halo[((size_t)q * world + rank) * (C + 1) + k].size();
                    op.send_idx = p->d_idx + send_at[(size_t)j * (C + 1) + k];
                    op.recv_idx = p->d_idx + recv_at[(size_t)j * (C + 1) + k];
                } else if (k == C + 1) {
                    // residual rows: the pieces of all colours of a rank are NOT contiguous -> one index-free op per colour would be
                    // C launches; instead the rows travel in device order of the sender's pieces, described by a generated list
                    op.n_send = own_rows; op.n_recv = own_rows;
                }
            }
    // residual rows need explicit lists too (piece p of every colour): append them
    {
        std::vector<int> rows((size_t)world * own_rows);
        for (int s = 0; s < world; ++s) {
            size_t at = (size_t)s * own_rows;
            for (int c = 0; c < C; ++c) {
                const int cnt = (h->lv[0].ord.color_begin[c + 1] - h->lv[0].ord.color_begin[c]) / world, lo = h->lv[0].ord.color_begin[c] + s * cnt;
                for (int i = 0; i < cnt; ++i) rows[at++] = lo + i;
            }
        }
        int* d_rows = nullptr;
        const size_t base = idx.size();
        HIPCHK(hipMalloc((void**)&d_rows, sizeof(int) * (base + rows.size())));
        if (base) HIPCHK(hipMemcpy(d_rows, p->d_idx, sizeof(int) * base, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(d_rows + base, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice));
        // re-point the lists at the new array
        for (auto& op : ops) {
            if (op.send_idx) op.send_idx = d_rows + (op.send_idx - p->d_idx);
            if (op.recv_idx) op.recv_idx = d_rows + (op.recv_idx - p->d_idx);
        }
        (void)hipFree(p->d_idx);
        p->d_idx = d_rows;
        for (int par = 0; par < 2; ++par)
            for (int j = 0; j < np; ++j) {
                gmgk::P2POp& op = ops[(size_t)((C + 1) * 2 + par) * np + j];
                op.send_idx = p->d_idx + base + (size_t)rank * own_rows;
                op.recv_idx = p->d_idx + base + (size_t)p->peers[j].rank * own_rows;
            }
    }
    if (p->d_ops) { (void)hipFree(p->d_ops); p->d_ops = nullptr; }
    HIPCHK(hipMalloc((void**)&p->d_ops, sizeof(gmgk::P2POp) * ops.size()));
    HIPCHK(hipMemcpy(p->d_ops, ops.data(), sizeof(gmgk::P2POp) * ops.size(), hipMemcpyHostToDevice));
    p->connected = true;
    return GMG_OK;
} GMG_CATCH_H

// Every rank loads the whole right-hand side and initial guess (host, natural numbering), like gmg_load_problem.
int gmg_p2p_load(gmg_handle h, const double* b, const double* x0) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected) return fail(h, GMG_ERR_STATE, "call gmg_p2p_prepare / gmg_p2p_connect first");
    unbind_level0(h);
    int rc = gmg_load_problem(h, b, x0, p->d);
    if (rc) return rc;
    // the distributed steps address the level-0 vectors through the `bound` state of the gmg_dist_* entry points
    Level& l = h->lv[0];
    h->own_x0 = l.x; h->own_b0 = l.b; h->own_r0 = l.r;
    h->bound = true;
    return GMG_OK;
} GMG_CATCH_H

namespace {

int p2p_smooth(gmg_handle h, int iters) {
    Level& l = h->lv[0];
    int rc;
    for (int it = 0; it < iters; ++it)
        for (int c = 0; c < l.ord.n_colors; ++c) {
            if ((rc = gmg_dist_smooth_color(h, c))) return rc;
            if ((rc = p2p_exchange(h, c, l.x))) return rc;
        }
    return GMG_OK;
}

int p2p_vcycle(gmg_handle h) {
    Level& l = h->lv[0];
    const int C = l.ord.n_colors;
    int rc;
    if ((rc = p2p_smooth(h, h->cfg.pre_iters))) return rc;                // :1063
    if ((rc = gmg_dist_residual_own(h))) return rc;                       // :1066, own rows
    if ((rc = p2p_exchange(h, C + 1, l.r))) return rc;                    //        everybody's rows -> complete r on every rank
    if ((rc = gmg_dist_coarse_cycle(h))) return rc;                       // :1069-1079, replicated
    if ((rc = gmg_dist_prolong_own(h))) return rc;                        // :1082, own rows
    if ((rc = p2p_exchange(h, C, l.x))) return rc;                        //        halo of all colours
    return p2p_smooth(h, h->cfg.post_iters);                              // :1085
}

}  // namespace

// n V-cycles, each followed by the residual check (stop_type >= 0): every rank calls this with the same arguments.
int gmg_p2p_cycles(gmg_handle h, int n_cycles, int stop_type, double* residues) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected || !h->bound) return fail(h, GMG_ERR_STATE, "no distributed problem loaded (gmg_p2p_load)");
    int rc;
    if (stop_type >= 0 && (rc = check_norm_type(h, stop_type))) return rc;
    const int d = p->d, np = (int)p->peers.size(), C = h->lv[0].ord.n_colors;
    for (int i = 0; i < n_cycles; ++i) {
        if ((rc = p2p_vcycle(h))) return rc;
        if (stop_type < 0) continue;
        double sums[8];
        if ((rc = dist_norm_launch(h, stop_type))) return rc;                // this rank's rows -> h->d_norm
        const double* d_result = h->d_norm;
        if (np > 0) {
            const int kind = C + 2, parity = (int)(p->kind_count[kind]++ & 1);
            ++p->seq;
            hipLaunchKernelGGL(gmgk::p2p_allreduce_small, dim3(1), dim3(64), 0, h->stream, p->d_ops + (size_t)(kind * 2 + parity) * np, np, p->rank,
                               (const double*)h->d_norm, 2 * d, p->d_sums, p->seq, p->d_err);
            d_result = p->d_sums;
        }
        HIPCHK(hipMemcpyAsync(h->h_norm, d_result, sizeof(double) * 2 * d, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        std::memcpy(sums, h->h_norm, sizeof(double) * 2 * d);
        if (residues) residues[i] = norm_from_sums(sums, d, stop_type);
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    int herr = 0;
    HIPCHK(hipMemcpy(&herr, p->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) return fail(h, GMG_ERR_STATE, "a peer-to-peer exchange timed out (a rank is missing or ran a different sequence)");
    return GMG_OK;
} GMG_CATCH_H

// Complete x on this rank (every rank's rows) and copy it out (host, natural numbering).  Collective.
int gmg_p2p_fetch(gmg_handle h, double* x) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected || !h->bound || !x) return fail(h, GMG_ERR_STATE, "no distributed problem loaded (gmg_p2p_load)");
    Level& l = h->lv[0];
    // the residual-rows exchange moves every rank's own rows of a level-0 vector: reuse it for x
    int rc = p2p_exchange(h, l.ord.n_colors + 1, l.x);
    if (rc) return rc;
    return to_host(h, 0, l.x, p->d, x);
} GMG_CATCH_H

// Average duration (ms) of one halo exchange of colour 0 (push + wait + pull, one launch), `reps` back to back; collective.
int gmg_p2p_bench_exchange(gmg_handle h, int reps, double* ms_avg) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected || !h->bound || !ms_avg || reps <= 0) return fail(h, GMG_ERR_STATE, "no distributed problem loaded (gmg_p2p_load)");
    Level& l = h->lv[0];
    for (int i = 0; i < 3; ++i) (void)p2p_exchange(h, 0, l.x);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) (void)p2p_exchange(h, 0, l.x);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_avg = (double)ms / reps;
    return GMG_OK;
} GMG_CATCH_H

int gmg_p2p_stat(gmg_handle h, const char* key, double* out) try {
    if (!h || !h->p2p || !key || !out) return GMG_ERR_INVALID;
    auto it = h->p2p->stats.find(key);
    if (it == h->p2p->stats.end()) return fail(h, GMG_ERR_INVALID, std::string("unknown key: ") + key);
    *out = it->second;
    return GMG_OK;
} GMG_CATCH_H

}  // extern "C"
