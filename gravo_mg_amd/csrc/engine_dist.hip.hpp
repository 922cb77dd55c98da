// engine_dist.hip.hpp -- part of libgravomg_hip.so's single translation unit (included by engine.hip after engine_cycle).
// Multi-GPU V-cycle driven by the ENGINE (one process per GPU): level 0 row-partitioned per colour, level 1 partitioned by
// blocks (cfg.dist_shard_levels >= 2; levels below it replicated), and every exchange a device-initiated store into the peer's
// mailbox (kernels.hip.hpp::p2p_exchange) -- no collective-library call and no Python between the launches of a cycle.  gravo_mg_amd/dist.py keeps the RCCL orchestration of the same cycle
// (torch.distributed all-gathers), which remains the fallback and the reference for the tests.
//
// Partition (BASELINE.json north_star): the device numbering of level 0 is colour-major, every colour class is
// padded to 64 * world rows and cut into `world` equal contiguous pieces; rank p owns piece p of every colour.  What moves:
//   * after every colour of every sweep: the entries of x this rank just updated that rows of ANOTHER rank read (the halo of
//     that colour; a few thousand values at 3 M vertices) -- per peer, only what that peer reads;
//   * once per cycle: the entries of the residual r0 that the restriction rows of another rank read (level 1 partitioned), or
//     this rank's rows of r0 to every peer (level 1 replicated: the coarse part restricts the whole r);
//   * after the prolongation: the halo of all colours at once;
//   * per residual check: 2 d partial sums, added in rank order on every rank (same bits everywhere).
// Level 1 (block-hybrid Gauss-Seidel: exact inside a 64-row block, couplings to other blocks taken from the previous sweep) is
// partitioned by blocks: a block goes to the rank that owns most of the fine rows its points prolong into, so that the two
// partitions cover the same part of the mesh and the transfers between them stay local.  A sweep reads the previous iterate at
// its off-block columns, so ONE exchange per sweep
// (the x1 entries other ranks' rows -- or their prolongation rows -- read) keeps it exact; the residual r1 is completed on every
// rank (its rows to all peers: 8 n1 / world bytes per link) for the replicated levels below.
// The iterates are those of the single-GPU engine: bitwise independent of the number of ranks.
#pragma once

struct P2PPeer {
    int rank = -1;
    void *mbox_base = nullptr, *flag_base = nullptr;      // the peer's mailbox / counters mapped here (hipIpcOpenMemHandle)
};

struct gmg_p2p_blob {                                     // what a rank publishes to the others (plain bytes, exchanged out of band)
    hipIpcMemHandle_t mbox, flags, coll;                  // coll: the gathered buffer of the emulated collective (dist_exchange = 2), else zero
    int rank, world, d, n_pad, n_colors, reserved;
    long long mbox_doubles, coll_doubles;
    char device_uuid[16];                                 // hipDeviceGetUuid of the rank's device: two ranks on one device are refused at connect time
};

// Collective exchange backend (gmg_config::dist_exchange = 1 / 2): every exchange of the cycle as pack -> all-gather -> unpack on the engine's
// stream (kernels.hip.hpp::coll_pack / coll_unpack), the all-gather being ncclAllGather (librccl, loaded at run time) or, between processes
// that share a device, its emulation through hipIpc mappings.
struct CollBackend {
    int mode = 0;                                         // 1: RCCL, 2: emulated over hipIpc
    std::vector<long long> chunk;                         // [kind] doubles a rank contributes (multiple of 8, the same on every rank)
    long long max_chunk = 0;
    double* send = nullptr;                               // max_chunk doubles: this rank's packed chunk
    double* recv = nullptr;                               // 2 x world x max_chunk doubles: the gathered chunks, double-buffered by the kind's parity
    gmgk::CollSeg* d_segs = nullptr;
    std::vector<int> pack_at, pack_n, unpack_at, unpack_n, blocks;      // ranges in d_segs: pack [kind], unpack [kind * 2 + parity]; blocks per segment [kind]
    int* d_idx = nullptr;
    void* lib = nullptr;                                  // librccl
    void* comm = nullptr;                                 // ncclComm_t
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    const char* (*error_string)(int) = nullptr;
    unsigned long long count = 0;                         // collective exchanges done: its parity picks the half of `recv` (see coll_exchange)
    gmgk::P2POp* d_ops = nullptr;                         // emulation: contiguous push of the chunk into every peer's gathered buffer, [(kind * 2 + parity) * n_peers + j]
    std::vector<void*> peer_recv;                         // the peers' gathered buffers mapped here (ascending rank)
};

struct DistP2P {
    bool planned = false, connected = false;
    int rank = 0, world = 1, d = 0, nk = 0;               // nk = exchange kinds: colours 0..C-1, C = halo of all colours, C+1 = level-0 rows, C+2 = norm sums,
                                                          // C+3 = x1 halo, C+4 = level-1 rows, C+5 = r0 halo of the restriction (the last three: level 1 partitioned)
    bool shard1 = false;
    bool hybrid = false;                                  // gmg_p2p_set_smoother: Gauss-Seidel inside a rank, Jacobi across ranks (one exchange per sweep)
    bool fold = false;                                    // gmg_p2p_set_smoother(2): the exchange of a colour folded into its sweep launch (gmgk::gs_color_push)
    bool fenced = true;                                   // gmg_p2p_set_fences: release / acquire fences around the sequence words (kernels.hip.hpp::publish_order)
    unsigned long long* d_plain_rows = nullptr;           // hybrid smoother: one word per level-0 slice, bit = the row couples to another rank's rows (takes omega = 1)
    int* d_pub = nullptr;                                 // per colour: pub_ptr (own slices + 1), then pub_ent (int2), gmgk::PushTail
    std::vector<size_t> pub_ptr_at, pub_ent_at;           // [colour]: offsets (ints) in d_pub
    std::vector<int> pub_waves;                           // [colour]: own slices with a published row
    unsigned long long exchange_launches = 0;             // launches spent on exchanges so far (gmg_p2p_stat "exchange_launches")
    std::vector<int> blk_owner;                           // [block of level 1] -> rank
    std::vector<std::vector<int>> own_blocks;             // [rank]: its blocks, ascending (block b = device rows 64 b .. 64 b + 63)
    int* d_l1 = nullptr;                                  // this rank's launch tables, one allocation:
    int *d_own_begin = nullptr, *d_own_ncolors = nullptr; //   first row / colour count of its blocks (the block sweep's tables)
    int* d_own_blocks = nullptr;                          //   their numbers (restrict_sweep0's block list)
    int *d_rsl = nullptr, *d_asl = nullptr, *d_psl = nullptr;   //   slices of U0^T, of A1 and of U1 that hold its rows
    int n_rsl = 0, n_asl = 0, n_psl = 0;
    std::vector<std::vector<int>> halo1, halo0r;          // [s * world + t]: x1 entries / r0 entries rank s publishes to rank t, ascending
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
    unsigned int* d_done = nullptr;                       // push blocks finished, per peer (multi-block exchanges)
    std::vector<int> kind_blocks;                         // [kind]: blocks per peer and direction of its exchange kernel
    double* d_sums = nullptr;                             // 2 d partial sums of this rank / the reduced sums
    std::vector<unsigned long long> kind_count;           // exchanges done per kind (parity = count & 1)
    unsigned long long seq = 0;                           // exchanges done in all (the arrival counters carry it)
    std::map<std::string, double> stats;
    CollBackend coll;
};

namespace {

void p2p_release(gmg_handle h) {
    DistP2P* p = h->p2p;
    if (!p) return;
    for (auto& peer : p->peers) {
        if (peer.mbox_base) (void)hipIpcCloseMemHandle(peer.mbox_base);
        if (peer.flag_base) (void)hipIpcCloseMemHandle(peer.flag_base);
    }
    if (p->mbox) (void)sync_hipFree(p->mbox);
    if (p->flags) (void)sync_hipFree(p->flags);
    if (p->d_idx) (void)sync_hipFree(p->d_idx);
    if (p->d_ops) (void)sync_hipFree(p->d_ops);
    if (p->d_err) (void)sync_hipFree(p->d_err);
    if (p->d_done) (void)sync_hipFree(p->d_done);
    if (p->d_sums) (void)sync_hipFree(p->d_sums);
    if (p->d_l1) (void)sync_hipFree(p->d_l1);
    if (p->d_pub) (void)sync_hipFree(p->d_pub);
    if (p->d_plain_rows) (void)sync_hipFree(p->d_plain_rows);
    CollBackend& cb = p->coll;
    for (void* q : cb.peer_recv) if (q) (void)hipIpcCloseMemHandle(q);
    if (cb.comm && cb.comm_destroy) (void)cb.comm_destroy(cb.comm);
    for (void* q : {(void*)cb.send, (void*)cb.recv, (void*)cb.d_segs, (void*)cb.d_idx, (void*)cb.d_ops}) if (q) (void)sync_hipFree(q);
    delete p;
    h->p2p = nullptr;
}

// The all-gather of the collective backend: every rank's chunk of `kind` (cb.send) into every rank's gathered buffer (parity half).
int coll_all_gather(gmg_handle h, int kind, int parity) {
    DistP2P* p = h->p2p;
    CollBackend& cb = p->coll;
    const int np = p->world - 1;
    const long long chunk = cb.chunk[kind];
    double* gathered = cb.recv + (size_t)parity * p->world * cb.max_chunk;
    if (cb.mode == 1) {
        const int rc = cb.all_gather(cb.send, gathered, (size_t)chunk, /*ncclFloat64*/ 8, cb.comm, h->stream);
        if (rc != 0) return fail(h, GMG_ERR_HIP, std::string("ncclAllGather: ") + (cb.error_string ? cb.error_string(rc) : "failed"));
        return GMG_OK;
    }
    // emulated: one launch -- per peer, blocks store this rank's chunk into the peer's gathered buffer and publish the sequence number; per
    // peer, a block waits for that peer's number (gmgk::p2p_exchange on contiguous ops with nothing to copy on the pull side)
    const int B = (int)std::min<long long>(64, std::max<long long>(1, (chunk + 4095) / 4096));
    hipLaunchKernelGGL(gmgk::p2p_exchange, dim3(2 * np * B), dim3(256), 0, h->stream, cb.d_ops + (size_t)(kind * 2 + parity) * np, np, cb.send, 0, 1, p->seq, p->d_err, B, p->d_done, p->fenced ? 1 : 0);
    return GMG_OK;
}

// One exchange through the collective backend: pack -> all-gather -> unpack, three launches on the engine's stream.
int coll_exchange(gmg_handle h, int kind, double* vec, int ld) {
    DistP2P* p = h->p2p;
    CollBackend& cb = p->coll;
    // The gathered buffer has two halves that ALL kinds share, used alternately by consecutive collective exchanges whatever their kind: a peer
    // writes exchange m + 2 into the half of exchange m only after it has seen this rank's contribution to m + 1, which this rank enqueued behind
    // its unpack of m.  (A parity per kind, as the mailbox regions have, would let two consecutive exchanges of different kinds share a half.)
    ++p->kind_count[kind];
    const int parity = (int)(cb.count++ & 1);
    ++p->seq;
    const int B = cb.blocks[kind];
    p->exchange_launches += (cb.pack_n[kind] > 0) + 1 + (cb.unpack_n[kind * 2 + parity] > 0);
    if (cb.pack_n[kind] > 0)
        hipLaunchKernelGGL(gmgk::coll_pack, dim3(cb.pack_n[kind] * B), dim3(256), 0, h->stream, cb.d_segs + cb.pack_at[kind], cb.pack_n[kind], (const double*)vec, ld, p->d, cb.send, B);
    int rc = coll_all_gather(h, kind, parity);
    if (rc) return rc;
    const int u = kind * 2 + parity;
    if (cb.unpack_n[u] > 0) {
        if (cb.mode == 2) hipLaunchKernelGGL(gmgk::coll_unpack<true>, dim3(cb.unpack_n[u] * B), dim3(256), 0, h->stream, cb.d_segs + cb.unpack_at[u], cb.unpack_n[u], (const double*)cb.recv, vec, ld, p->d, B);
        else hipLaunchKernelGGL(gmgk::coll_unpack<false>, dim3(cb.unpack_n[u] * B), dim3(256), 0, h->stream, cb.d_segs + cb.unpack_at[u], cb.unpack_n[u], (const double*)cb.recv, vec, ld, p->d, B);
    }
    return GMG_OK;
}

// One exchange of kind `kind` on vector `vec` (leading dimension ld) -- or nothing with a single rank.
int p2p_exchange(gmg_handle h, int kind, double* vec, int ld) {
    DistP2P* p = h->p2p;
    if (p->coll.mode != 0) return coll_exchange(h, kind, vec, ld);
    const int np = (int)p->peers.size();
    if (np == 0) return GMG_OK;
    const int parity = (int)(p->kind_count[kind]++ & 1);
    ++p->seq;
    const int B = p->kind_blocks[kind];
    ++p->exchange_launches;
    hipLaunchKernelGGL(gmgk::p2p_exchange, dim3(2 * np * B), dim3(256), 0, h->stream, p->d_ops + (size_t)(kind * 2 + parity) * np, np, vec, ld, p->d,
                       p->seq, p->d_err, B, p->d_done, p->fenced ? 1 : 0);
    return GMG_OK;
}

// One colour of a level-0 sweep on this rank's rows WITH its exchange (gmgk::gs_color_push): false when the colour has to take the two-launch
// form (no rows of this colour here, collective backend, tables missing).
bool p2p_smooth_color_folded(gmg_handle h, int c) {
    DistP2P* p = h->p2p;
    const int np = (int)p->peers.size();
    if (!p->fold || p->hybrid || p->coll.mode != 0 || np == 0 || !p->d_pub || h->dist_all_rows || h->loaded_d != p->d) return false;
    Level& l = h->lv[0];
    int sb, se;
    own_range(h, c, sb, se);
    if (se <= sb) return false;
    const int parity = (int)(p->kind_count[c]++ & 1);
    ++p->seq;
    gmgk::PushTail pt;
    pt.pub_ptr = p->d_pub + p->pub_ptr_at[c];
    pt.pub_ent = reinterpret_cast<const int2*>(p->d_pub + p->pub_ent_at[c]);
    pt.ops = p->d_ops + (size_t)(c * 2 + parity) * np;
    pt.n_peers = np; pt.n_pub_waves = p->pub_waves[c];
    pt.done = p->d_done + p->world;
    pt.seq = p->seq; pt.err = p->d_err; pt.fenced = p->fenced ? 1 : 0;
    const int ld = l.n_pad;
    const dim3 grid(grid_for(se - sb)), block(gmgk::kBlock);
    if (l.Aoff.c16_mode != 0) {
        DISPATCH_D(p->d, DISPATCH_C16(l.Aoff.c16_sel(), hipLaunchKernelGGL((gmgk::gs_color_push<D, C16 + 1>), grid, block, 0, h->stream, l.Aoff.slice_ptr, l.Aoff.col, l.Aoff.val, l.diag, l.b, l.x, ld, sb, se,
                                            h->cfg.gs_omega, l.Aoff.col16, l.Aoff.win_base, l.Aoff.c16_arg(), pt)));
    } else {
        DISPATCH_D(p->d, hipLaunchKernelGGL((gmgk::gs_color_push<D, 1>), grid, block, 0, h->stream, l.Aoff.slice_ptr, l.Aoff.col, l.Aoff.val, l.diag, l.b, l.x, ld, sb, se,
                                            h->cfg.gs_omega, (const unsigned*)nullptr, (const int*)nullptr, 0, pt));
    }
    return true;
}

// the index list rank s publishes to rank t for exchange kind k (null: the kind has no list)
const std::vector<int>* p2p_list(const DistP2P* p, int C, int s, int t, int k) {
    if (k <= C) return &p->halo[((size_t)s * p->world + t) * (C + 1) + k];
    if (k == C + 3 && p->shard1) return &p->halo1[(size_t)s * p->world + t];
    if (k == C + 5 && p->shard1) return &p->halo0r[(size_t)s * p->world + t];
    return nullptr;
}

// ---- collective backend: chunk sizes, segment tables and buffers (all ranks derive the same numbers from the plan)
int coll_build(gmg_handle h) {
    DistP2P* p = h->p2p;
    CollBackend& cb = p->coll;
    // (RCCL takes a communicator of one rank: dist_exchange = 1 on a single rank runs every exchange of the cycle through the real library --
    // dlopen, ncclCommInitRank, ncclAllGather on the engine's stream; the emulation needs a peer to store into)
    cb.mode = (p->world > 1 || h->cfg.dist_exchange == 1) ? h->cfg.dist_exchange : 0;
    if (cb.mode == 0) return GMG_OK;
    const int world = p->world, rank = p->rank, d = p->d, nk = p->nk, C = dist_classes(h->lv[0].ord);
    const LevelOrdering& o = h->lv[0].ord;
    const int own_rows = h->lv[0].n_pad / world;
    auto up8 = [](long long v) { return (v + 7) / 8 * 8; };
    // ---- index lists: every (source, destination) list of every listed kind, the level-0 rows of every rank, the level-1 rows of every rank
    std::vector<int> idx;
    std::vector<size_t> list_at((size_t)world * world * nk, 0), rows0_at(world, 0), rows1_at(world, 0);
    std::vector<int> rows1_n(world, 0);
    for (int k = 0; k < nk; ++k)
        for (int s = 0; s < world; ++s)
            for (int t = 0; t < world; ++t) {
                const std::vector<int>* l = s == t ? nullptr : p2p_list(p, C, s, t, k);
                if (!l || (s != rank && t != rank)) continue;                      // (this rank packs its own lists and unpacks the ones addressed to it)
                list_at[((size_t)s * world + t) * nk + k] = idx.size();
                idx.insert(idx.end(), l->begin(), l->end());
            }
    for (int s = 0; s < world; ++s) {
        rows0_at[s] = idx.size();
        for (int c = 0; c < C; ++c) {
            const int cnt = (o.color_begin[c + 1] - o.color_begin[c]) / world, lo = o.color_begin[c] + s * cnt;
            for (int i = 0; i < cnt; ++i) idx.push_back(lo + i);
        }
    }
    if (p->shard1)
        for (int s = 0; s < world; ++s) {
            rows1_at[s] = idx.size();
            for (int b : p->own_blocks[s]) for (int i = 0; i < 64; ++i) idx.push_back(64 * b + i);
            rows1_n[s] = 64 * (int)p->own_blocks[s].size();
        }
    // ---- chunk of every kind: the largest contribution of any rank
    cb.chunk.assign(nk, 0);
    for (int k = 0; k < nk; ++k) {
        long long most = 0;
        if (k == C + 1) most = (long long)own_rows * d;
        else if (k == C + 2) most = 8;                                              // 2 d norm sums (d <= 4)
        else if (k == C + 4) { if (p->shard1) for (int s = 0; s < world; ++s) most = std::max<long long>(most, (long long)rows1_n[s] * d); }
        else
            for (int s = 0; s < world; ++s) {
                long long tot = 0;
                for (int t = 0; t < world; ++t) if (t != s) if (const std::vector<int>* l = p2p_list(p, C, s, t, k)) tot += (long long)l->size() * d;
                most = std::max(most, tot);
            }
        cb.chunk[k] = up8(most);
        if (cb.mode == 1) cb.chunk[k] = std::max<long long>(cb.chunk[k], 8);      // (no zero-count ncclAllGather: a kind without payload still is one collective on every rank)
        cb.max_chunk = std::max(cb.max_chunk, cb.chunk[k]);
    }
    cb.max_chunk = std::max<long long>(cb.max_chunk, 8);
    // ---- segment tables
    struct Seg { size_t at; int n; long long off; };
    std::vector<Seg> segs;
    cb.pack_at.assign(nk, 0); cb.pack_n.assign(nk, 0); cb.unpack_at.assign((size_t)nk * 2, 0); cb.unpack_n.assign((size_t)nk * 2, 0); cb.blocks.assign(nk, 1);
    for (int k = 0; k < nk; ++k) {
        long long most = 0;
        const bool rows0 = k == C + 1, rows1 = k == C + 4 && p->shard1, listed = p2p_list(p, C, rank, (rank + 1) % world, k) != nullptr;
        cb.pack_at[k] = (int)segs.size();
        if (rows0) segs.push_back({rows0_at[rank], own_rows, 0});
        else if (rows1) segs.push_back({rows1_at[rank], rows1_n[rank], 0});
        else if (listed) {
            long long off = 0;
            for (int t = 0; t < world; ++t) {
                if (t == rank) continue;
                const std::vector<int>* l = p2p_list(p, C, rank, t, k);
                segs.push_back({list_at[((size_t)rank * world + t) * nk + k], (int)l->size(), off});
                off += (long long)l->size() * d;
            }
        }
        cb.pack_n[k] = (int)segs.size() - cb.pack_at[k];
        for (int par = 0; par < 2; ++par) {
            const long long base = (long long)par * world * cb.max_chunk;
            cb.unpack_at[(size_t)k * 2 + par] = (int)segs.size();
            for (int s = 0; s < world && (rows0 || rows1 || listed); ++s) {
                if (s == rank) continue;
                if (rows0) segs.push_back({rows0_at[s], own_rows, base + (long long)s * cb.chunk[k]});
                else if (rows1) segs.push_back({rows1_at[s], rows1_n[s], base + (long long)s * cb.chunk[k]});
                else {
                    long long off = 0;
                    for (int t = 0; t < rank; ++t) if (t != s) off += (long long)p2p_list(p, C, s, t, k)->size() * d;
                    const std::vector<int>* l = p2p_list(p, C, s, rank, k);
                    segs.push_back({list_at[((size_t)s * world + rank) * nk + k], (int)l->size(), base + (long long)s * cb.chunk[k] + off});
                }
            }
            cb.unpack_n[(size_t)k * 2 + par] = (int)segs.size() - cb.unpack_at[(size_t)k * 2 + par];
        }
        for (int q = cb.pack_at[k]; q < (int)segs.size(); ++q) most = std::max<long long>(most, (long long)segs[q].n * d);
        cb.blocks[k] = (int)std::min<long long>(64, std::max<long long>(1, (most + 1023) / 1024));
    }
    HIPCHK(hipMalloc((void**)&cb.d_idx, sizeof(int) * std::max<size_t>(idx.size(), 1)));
    HIPCHK(hipMemcpy(cb.d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    std::vector<gmgk::CollSeg> dev(segs.size());
    for (size_t q = 0; q < segs.size(); ++q) dev[q] = gmgk::CollSeg{cb.d_idx + segs[q].at, segs[q].n, segs[q].off};
    HIPCHK(hipMalloc((void**)&cb.d_segs, sizeof(gmgk::CollSeg) * std::max<size_t>(dev.size(), 1)));
    HIPCHK(hipMemcpy(cb.d_segs, dev.data(), sizeof(gmgk::CollSeg) * dev.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&cb.send, sizeof(double) * (size_t)cb.max_chunk));
    const size_t recv_doubles = (size_t)2 * world * cb.max_chunk;
    if (cb.mode == 2) HIPCHK(hipExtMallocWithFlags((void**)&cb.recv, sizeof(double) * recv_doubles, hipDeviceMallocFinegrained));
    else HIPCHK(hipMalloc((void**)&cb.recv, sizeof(double) * recv_doubles));
    HIPCHK(hipMemsetAsync(cb.send, 0, sizeof(double) * (size_t)cb.max_chunk, h->stream));
    HIPCHK(hipMemsetAsync(cb.recv, 0, sizeof(double) * recv_doubles, h->stream));
    p->stats["collective_chunk_doubles_max"] = (double)cb.max_chunk;
    return GMG_OK;
}

// librccl, loaded at run time (the library does not link against it: a box without RCCL still runs the peer-to-peer path)
struct RcclApi {
    void* lib = nullptr;
    struct UniqueId { char internal[128]; };
    int (*get_unique_id)(UniqueId*) = nullptr;
    int (*comm_init_rank)(void**, int, UniqueId, int) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    const char* (*error_string)(int) = nullptr;
    bool ok() const { return lib && get_unique_id && comm_init_rank && all_gather && comm_destroy; }
};
RcclApi& rccl_api() {
    static RcclApi* api = [] {
        RcclApi* a = new RcclApi();
        // the copy a host application (PyTorch) has loaded already, else the system's
        for (const char* name : {"librccl.so", "librccl.so.1"}) if (!a->lib) a->lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if (!a->lib) a->lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (a->lib) {
            a->get_unique_id = (int (*)(RcclApi::UniqueId*))dlsym(a->lib, "ncclGetUniqueId");
            a->comm_init_rank = (int (*)(void**, int, RcclApi::UniqueId, int))dlsym(a->lib, "ncclCommInitRank");
            a->all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(a->lib, "ncclAllGather");
            a->comm_destroy = (int (*)(void*))dlsym(a->lib, "ncclCommDestroy");
            a->error_string = (const char* (*)(int))dlsym(a->lib, "ncclGetErrorString");
        }
        return a;
    }();
    return *api;
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
    p->fenced = !EnvSwitches::get().p2p_fence_free;
    Level& l = h->lv[0];
    const LevelOrdering& o = l.ord;
    const int C = dist_classes(o);
    p->nk = C + 6;
    p->kind_count.assign(p->nk, 0);
    p->own_lo.resize(C); p->own_cnt.resize(C);
    for (int c = 0; c < C; ++c) { p->own_cnt[c] = (o.color_begin[c + 1] - o.color_begin[c]) / world; p->own_lo[c] = o.color_begin[c] + rank * p->own_cnt[c]; }
    // ---- who owns and who reads what (engine_part.hip.hpp::build_dist_plan).  A partitioned set-up (gmg_dist_partition) made the plan while it
    // had the patterns at hand -- its level-0 / level-1 operators hold this rank's rows only, the natural copies are gone; otherwise it is made
    // now from host copies of the patterns
    std::shared_ptr<DistPlan> plan;
    if (h->partitioned) {
        if (!h->plan || h->plan->rank != rank || h->plan->world != world) return fail(h, GMG_ERR_STATE, "the system was partitioned for another rank / world size (gmg_dist_partition)");
        plan = h->plan;
    } else {
        plan = std::make_shared<DistPlan>();
        const bool shard1 = plan_can_shard_level1(h, world, true);
        if (world > 1) {
            if ((rc = ensure_host_A(h, 0, false))) return rc;
            if (shard1 && (rc = ensure_host_A(h, 1, false))) return rc;
        }
        const Compressed& A0 = l.A;
        if ((rc = build_dist_plan(h, *plan, rank, world, PatternView{A0.n_outer, A0.ptr.data(), A0.idx.data()}, shard1 ? &h->lv[1].A : nullptr, shard1))) return rc;
    }
    p->shard1 = plan->shard1;
    p->halo = plan->halo; p->halo1 = plan->halo1; p->halo0r = plan->halo0r;
    p->blk_owner = plan->blk_owner; p->own_blocks = plan->own_blocks;
    {   // rows of this rank that read another rank's rows = (the pattern is symmetric) the rows it publishes: they take the plain Gauss-Seidel
        // update in the hybrid smoother, whose coupling across ranks is a Jacobi one (gmgk::gs_color<..., OM = 1>)
        std::vector<unsigned long long> words((size_t)l.n_pad / 64, 0ull);
        double cnt = 0;
        for (int t = 0; t < world; ++t)
            if (t != rank)
                for (int row : p->halo[((size_t)rank * world + t) * (C + 1) + C]) { unsigned long long& w = words[(size_t)row >> 6]; const unsigned long long bit = 1ull << (row & 63); cnt += !(w & bit); w |= bit; }
        HIPCHK(hipMalloc((void**)&p->d_plain_rows, sizeof(unsigned long long) * std::max<size_t>(words.size(), 1)));
        if (!words.empty()) HIPCHK(hipMemcpy(p->d_plain_rows, words.data(), sizeof(unsigned long long) * words.size(), hipMemcpyHostToDevice));
        p->stats["boundary_rows"] = cnt;
    }
    if (p->shard1) {
        // this rank's launch tables
        Level& l1 = h->lv[1];
        const LevelOrdering& o1 = l1.ord;
        const std::vector<int>& mine = p->own_blocks[rank];
        const int rps_r = 64 / l.R.lpr, rps_a = 64 / l1.Aoff.lpr, rps_p = 64 / l1.P.lpr;
        std::vector<int> tab;
        for (int b : mine) tab.push_back(o1.blk_begin[b]);
        for (int b : mine) tab.push_back(o1.blk_ncolors[b]);
        for (int b : mine) tab.push_back(b);
        for (int rps : {rps_r, rps_a, rps_p}) for (int b : mine) for (int q = 0; q < 64 / rps; ++q) tab.push_back(b * (64 / rps) + q);
        HIPCHK(hipMalloc((void**)&p->d_l1, sizeof(int) * std::max<size_t>(tab.size(), 1)));
        if (!tab.empty()) HIPCHK(hipMemcpy(p->d_l1, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
        const int nm = (int)mine.size();
        p->d_own_begin = p->d_l1; p->d_own_ncolors = p->d_l1 + nm; p->d_own_blocks = p->d_l1 + 2 * nm;
        p->n_rsl = nm * (64 / rps_r); p->n_asl = nm * (64 / rps_a); p->n_psl = nm * (64 / rps_p);
        p->d_rsl = p->d_l1 + 3 * nm; p->d_asl = p->d_rsl + p->n_rsl; p->d_psl = p->d_asl + p->n_asl;
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
                    long long cnt = 0;
                    if (const std::vector<int>* v = p2p_list(p, C, src, dst, k)) cnt = src == dst ? 0 : (long long)v->size() * d;
                    else if (k == C + 1) cnt = src == dst ? 0 : own_rows * d;
                    else if (k == C + 2) cnt = 2LL * d;                    // norm sums: a slot for every source, the own one included
                    else if (k == C + 4 && p->shard1) cnt = src == dst ? 0 : 64LL * (long long)p->own_blocks[src].size() * d;
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
    {
        const double sec = EnvSwitches::get().p2p_timeout_s;      // GMG_P2P_TIMEOUT_S
        if (sec > 0.0) { const unsigned long long ticks = (unsigned long long)(sec * 1e8); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(gmgk::g_p2p_timeout_ticks), &ticks, sizeof(ticks))); }
    }
    HIPCHK(hipMalloc((void**)&p->d_sums, sizeof(double) * 4 * d));
    if (!p->d_done) { HIPCHK(hipMalloc((void**)&p->d_done, sizeof(unsigned int) * (world + 1))); HIPCHK(hipMemset(p->d_done, 0, sizeof(unsigned int) * (world + 1))); }      // (+ 1: gs_color_push)
    if ((rc = coll_build(h))) return rc;                  // gmg_config::dist_exchange != 0: chunk sizes, segment tables, buffers
    HIPCHK(hipStreamSynchronize(h->stream));
    p->planned = true;
    auto published = [&](int k) { double n = 0; for (int t = 0; t < world; ++t) if (t != rank) if (const std::vector<int>* v = p2p_list(p, C, rank, t, k)) n += (double)v->size(); return n; };
    p->stats["halo_rows_published"] = published(C);
    p->stats["level1_partitioned"] = p->shard1 ? 1.0 : 0.0;
    p->stats["x1_halo_rows_published"] = published(C + 3);
    p->stats["r0_halo_rows_published"] = published(C + 5);
    p->stats["level1_own_rows"] = p->shard1 ? 64.0 * (double)p->own_blocks[rank].size() : (double)h->lv[std::min(1, h->L)].n_pad;
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
    b.rank = p->rank; b.world = p->world; b.d = p->d; b.n_pad = h->lv[0].n_pad; b.n_colors = dist_classes(h->lv[0].ord); b.mbox_doubles = p->box_total[p->rank];
    b.reserved = (p->shard1 ? 1 : 0) | (p->coll.mode << 4);
    if (p->coll.mode == 2) { HIPCHK(hipIpcGetMemHandle(&b.coll, p->coll.recv)); b.coll_doubles = 2LL * p->world * p->coll.max_chunk; }
    {
        hipUUID id;
        std::memset(&id, 0, sizeof(id));
        if (hipDeviceGetUuid(&id, h->cfg.device) != hipSuccess) { (void)hipGetLastError(); std::memset(&id, 0, sizeof(id)); }
        static_assert(sizeof(id.bytes) == sizeof(b.device_uuid), "hipUUID is 16 bytes");
        std::memcpy(b.device_uuid, id.bytes, sizeof(b.device_uuid));
    }
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
    const int world = p->world, rank = p->rank, C = dist_classes(h->lv[0].ord), nk = p->nk, d = p->d;
    const gmg_p2p_blob* bl = (const gmg_p2p_blob*)blobs;
    // (a second connect: the mappings of the first one are closed, not dropped -- re-opening a handle that is still mapped can fail)
    auto close_peer = [](P2PPeer& peer) {
        if (peer.mbox_base) (void)hipIpcCloseMemHandle(peer.mbox_base);
        if (peer.flag_base) (void)hipIpcCloseMemHandle(peer.flag_base);
        peer.mbox_base = peer.flag_base = nullptr;
    };
    for (auto& peer : p->peers) close_peer(peer);
    p->peers.clear();
    for (void*& q : p->coll.peer_recv) { if (q) (void)hipIpcCloseMemHandle(q); q = nullptr; }
    p->coll.peer_recv.clear();
    p->connected = false;
    if (p->coll.mode == 1) return fail(h, GMG_ERR_STATE, "this handle exchanges through RCCL (gmg_config::dist_exchange = 1): connect it with gmg_p2p_connect_rccl");
    // one device per rank: the exchange kernels of ranks that share a device wait for each other ON that device (2.5 ms per cycle with four
    // ranks on one GPU, half a second with eight) -- refused unless the caller says it is meant (GMG_P2P_SHARED_DEVICE=1: functional tests)
    if (!EnvSwitches::get().p2p_shared_device) {
        const char zero[16] = {0};
        for (int a = 0; a < world; ++a)
            for (int b = a + 1; b < world; ++b)
                if (std::memcmp(bl[a].device_uuid, zero, 16) != 0 && std::memcmp(bl[a].device_uuid, bl[b].device_uuid, 16) == 0)
                    return fail(h, GMG_ERR_STATE, "ranks " + std::to_string(a) + " and " + std::to_string(b) + " are on the same device (same UUID): one device per rank "
                                "(check HIP_VISIBLE_DEVICES / LOCAL_RANK); GMG_P2P_SHARED_DEVICE=1 allows it for functional tests");
    }
    {   // peer access to every other visible device (the IPC mapping below enables it lazily as well; "already enabled" and "not
        // supported" are both fine here -- an unreachable peer shows up in hipIpcOpenMemHandle)
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess)
            for (int dev = 0; dev < ndev; ++dev)
                if (dev != h->cfg.device) { int can = 0; if (hipDeviceCanAccessPeer(&can, h->cfg.device, dev) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(dev, 0); }
        (void)hipGetLastError();
    }
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        P2PPeer peer;
        peer.rank = q;
        if (bl[q].rank != q || bl[q].world != world || bl[q].d != d || bl[q].n_pad != h->lv[0].n_pad || bl[q].n_colors != C || bl[q].mbox_doubles != p->box_total[q] ||
            bl[q].reserved != ((p->shard1 ? 1 : 0) | (p->coll.mode << 4)) || (p->coll.mode == 2 && bl[q].coll_doubles != 2LL * world * p->coll.max_chunk))
            return fail(h, GMG_ERR_INVALID, "peer " + std::to_string(q) + " published a different partition plan (different system / ordering / configuration?)");
        // (every mapping that was opened is in p->peers -- or closed -- before an error leaves this function: p2p_release sees it)
        if (hipIpcOpenMemHandle(&peer.mbox_base, bl[q].mbox, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            return fail(h, GMG_ERR_HIP, "hipIpcOpenMemHandle failed for the mailbox of rank " + std::to_string(q));
        }
        if (hipIpcOpenMemHandle(&peer.flag_base, bl[q].flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            close_peer(peer);
            return fail(h, GMG_ERR_HIP, "hipIpcOpenMemHandle failed for the arrival counters of rank " + std::to_string(q));
        }
        p->peers.push_back(peer);
        if (p->coll.mode == 2) {
            void* base = nullptr;
            if (hipIpcOpenMemHandle(&base, bl[q].coll, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError();
                return fail(h, GMG_ERR_HIP, "hipIpcOpenMemHandle failed for the gathered buffer of rank " + std::to_string(q));
            }
            p->coll.peer_recv.push_back(base);
        }
    }
    const int np = (int)p->peers.size();
    if (np == 0) { p->connected = true; return GMG_OK; }
    if (p->coll.mode == 2) {
        // the emulated all-gather: per (kind, parity, peer) one contiguous push of this rank's chunk into slot `rank` of the peer's gathered buffer
        CollBackend& cb = p->coll;
        std::vector<gmgk::P2POp> cops((size_t)nk * 2 * np);
        for (int k = 0; k < nk; ++k)
            for (int par = 0; par < 2; ++par)
                for (int j = 0; j < np; ++j) {
                    gmgk::P2POp& op = cops[(size_t)(k * 2 + par) * np + j];
                    const int q = p->peers[j].rank;
                    op.send_idx = op.recv_idx = nullptr; op.send_lo = op.recv_lo = 0; op.n_recv = 0; op.local_box = nullptr;
                    op.n_send = (int)cb.chunk[k];
                    op.remote_box = (double*)cb.peer_recv[j] + (size_t)par * world * cb.max_chunk + (size_t)rank * cb.chunk[k];
                    op.remote_flag = (unsigned long long*)p->peers[j].flag_base + 64 * (size_t)rank;
                    op.local_flag = p->flags + 64 * (size_t)q;
                }
        if (cb.d_ops) { (void)sync_hipFree(cb.d_ops); cb.d_ops = nullptr; }
        HIPCHK(hipMalloc((void**)&cb.d_ops, sizeof(gmgk::P2POp) * cops.size()));
        HIPCHK(hipMemcpy(cb.d_ops, cops.data(), sizeof(gmgk::P2POp) * cops.size(), hipMemcpyHostToDevice));
    }
    // ---- all index lists in one device array: per (peer, listed kind) the rows sent and the rows received; then the level-0 rows
    // of every rank (piece s of every colour, in device order)
    std::vector<int> idx;
    std::vector<size_t> send_at((size_t)np * nk, 0), recv_at((size_t)np * nk, 0);
    for (int j = 0; j < np; ++j)
        for (int k = 0; k < nk; ++k) {
            const std::vector<int>* sl = p2p_list(p, C, rank, p->peers[j].rank, k);
            const std::vector<int>* rl = p2p_list(p, C, p->peers[j].rank, rank, k);
            if (!sl) continue;
            send_at[(size_t)j * nk + k] = idx.size(); idx.insert(idx.end(), sl->begin(), sl->end());
            recv_at[(size_t)j * nk + k] = idx.size(); idx.insert(idx.end(), rl->begin(), rl->end());
        }
    const int own_rows = h->lv[0].n_pad / world;
    const size_t rows_at = idx.size();
    idx.resize(rows_at + (size_t)world * own_rows);
    for (int s = 0; s < world; ++s) {
        size_t at = rows_at + (size_t)s * own_rows;
        for (int c = 0; c < C; ++c) {
            const int cnt = (h->lv[0].ord.color_begin[c + 1] - h->lv[0].ord.color_begin[c]) / world, lo = h->lv[0].ord.color_begin[c] + s * cnt;
            for (int i = 0; i < cnt; ++i) idx[at++] = lo + i;
        }
    }
    // ... and the level-1 rows of every rank (its blocks, ascending)
    std::vector<size_t> rows1_at(world + 1, idx.size());
    if (p->shard1)
        for (int s = 0; s < world; ++s) {
            rows1_at[s] = idx.size();
            for (int b : p->own_blocks[s]) for (int i = 0; i < 64; ++i) idx.push_back(64 * b + i);
            rows1_at[s + 1] = idx.size();
        }
    if (p->d_idx) { (void)sync_hipFree(p->d_idx); p->d_idx = nullptr; }
    HIPCHK(hipMalloc((void**)&p->d_idx, sizeof(int) * std::max<size_t>(idx.size(), 1)));
    HIPCHK(hipMemcpy(p->d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
    // ---- one op table per (kind, parity)
    std::vector<gmgk::P2POp> ops((size_t)nk * 2 * np);
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
                if (const std::vector<int>* sl = p2p_list(p, C, rank, q, k)) {
                    op.n_send = (int)sl->size();
                    op.n_recv = (int)p2p_list(p, C, q, rank, k)->size();
                    op.send_idx = p->d_idx + send_at[(size_t)j * nk + k];
                    op.recv_idx = p->d_idx + recv_at[(size_t)j * nk + k];
                } else if (k == C + 1) {
                    // level-0 rows: the pieces of a rank (one per colour) are not contiguous -> the generated lists
                    op.n_send = own_rows; op.n_recv = own_rows;
                    op.send_idx = p->d_idx + rows_at + (size_t)rank * own_rows;
                    op.recv_idx = p->d_idx + rows_at + (size_t)q * own_rows;
                } else if (k == C + 4 && p->shard1) {
                    op.n_send = 64 * (int)p->own_blocks[rank].size(); op.n_recv = 64 * (int)p->own_blocks[q].size();
                    op.send_idx = p->d_idx + rows1_at[rank];
                    op.recv_idx = p->d_idx + rows1_at[q];
                }
            }
    // blocks per peer and direction: one per 1024 values of the largest transfer of the kind (all ranks derive the same number)
    p->kind_blocks.assign(nk, 1);
    for (int k = 0; k < nk; ++k) {
        long long most = 0;
        for (int par = 0; par < 1; ++par)
            for (int j = 0; j < np; ++j) { const gmgk::P2POp& op = ops[(size_t)(k * 2 + par) * np + j]; most = std::max<long long>(most, (long long)std::max(op.n_send, op.n_recv) * d); }
        p->kind_blocks[k] = (int)std::min<long long>(64, std::max<long long>(1, (most + 1023) / 1024));       // (a block per 1 024 values: four dependent index -> value -> store chains per thread)
    }
    if (p->d_ops) { (void)sync_hipFree(p->d_ops); p->d_ops = nullptr; }
    HIPCHK(hipMalloc((void**)&p->d_ops, sizeof(gmgk::P2POp) * ops.size()));
    HIPCHK(hipMemcpy(p->d_ops, ops.data(), sizeof(gmgk::P2POp) * ops.size(), hipMemcpyHostToDevice));
    // ---- tables of the folded exchange (gmgk::gs_color_push): per colour, the entries of this rank's send lists by slice of its piece
    {
        std::vector<int> tab;
        p->pub_ptr_at.assign(C, 0); p->pub_ent_at.assign(C, 0); p->pub_waves.assign(C, 0);
        bool fits = true;
        for (int c = 0; c < C; ++c) {
            int sb, se;
            own_range(h, c, sb, se);
            const int ns = std::max(se - sb, 0);
            std::vector<std::array<int, 2>> ent;
            for (int j = 0; j < np; ++j) {
                const std::vector<int>& sl = *p2p_list(p, C, rank, p->peers[j].rank, c);
                if (sl.size() >= ((size_t)1 << 24)) fits = false;
                for (size_t k = 0; k < sl.size(); ++k) {
                    if (sl[k] < sb * 64 || sl[k] >= se * 64) { fits = false; continue; }
                    ent.push_back({sl[k], (int)((unsigned)j << 24 | (unsigned)k)});
                }
            }
            std::stable_sort(ent.begin(), ent.end(), [](const std::array<int, 2>& a, const std::array<int, 2>& b) { return a[0] < b[0]; });
            if (tab.size() & 1) tab.push_back(0);
            p->pub_ptr_at[c] = tab.size();
            std::vector<int> ptr((size_t)ns + 1, 0);
            for (const auto& e : ent) ++ptr[(size_t)(e[0] / 64 - sb) + 1];
            for (int q = 0; q < ns; ++q) { p->pub_waves[c] += ptr[q + 1] > 0; ptr[q + 1] += ptr[q]; }
            tab.insert(tab.end(), ptr.begin(), ptr.end());
            if (tab.size() & 1) tab.push_back(0);                          // (int2 entries: 8-byte aligned)
            p->pub_ent_at[c] = tab.size();
            for (const auto& e : ent) { tab.push_back(e[0]); tab.push_back(e[1]); }
        }
        if (p->d_pub) { (void)sync_hipFree(p->d_pub); p->d_pub = nullptr; }
        if (fits && np < 64) {
            HIPCHK(hipMalloc((void**)&p->d_pub, sizeof(int) * std::max<size_t>(tab.size(), 2)));
            HIPCHK(hipMemcpy(p->d_pub, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
        }
    }
    p->connected = true;
    return GMG_OK;
} GMG_CATCH_H

// RCCL transport of the collective backend (gmg_config::dist_exchange = 1).  Rank 0 makes the id, the caller hands its 128 bytes to every rank
// (torch.distributed broadcast_object_list, MPI_Bcast, a file), every rank calls gmg_p2p_connect_rccl after gmg_p2p_prepare.  No hipIpc involved.
int gmg_p2p_rccl_unique_id(void* id_out) try {
    if (!id_out) return GMG_ERR_INVALID;
    RcclApi& api = rccl_api();
    if (!api.ok()) return GMG_ERR_UNSUPPORTED;
    RcclApi::UniqueId id;
    if (api.get_unique_id(&id) != 0) return GMG_ERR_HIP;
    std::memcpy(id_out, &id, sizeof(id));
    return GMG_OK;
} GMG_CATCH_0

int gmg_p2p_connect_rccl(gmg_handle h, const void* id_in) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->planned || !id_in) return fail(h, GMG_ERR_STATE, "call gmg_p2p_prepare first");
    if (p->coll.mode != 1) {
        if (p->world > 1) return fail(h, GMG_ERR_STATE, "create the handle with gmg_config::dist_exchange = 1 to exchange through RCCL");
        p->connected = true;                                                // one rank without the collective backend: nothing to connect
        return GMG_OK;
    }
    RcclApi& api = rccl_api();
    if (!api.ok()) return fail(h, GMG_ERR_UNSUPPORTED, "librccl could not be loaded (ncclGetUniqueId / ncclCommInitRank / ncclAllGather)");
    HIPCHK(hipSetDevice(h->cfg.device));
    CollBackend& cb = p->coll;
    if (cb.comm && cb.comm_destroy) { (void)cb.comm_destroy(cb.comm); cb.comm = nullptr; }
    RcclApi::UniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    const int rc = api.comm_init_rank(&cb.comm, p->world, id, p->rank);
    if (rc != 0) { cb.comm = nullptr; return fail(h, GMG_ERR_HIP, std::string("ncclCommInitRank: ") + (api.error_string ? api.error_string(rc) : "failed")); }
    cb.all_gather = api.all_gather; cb.comm_destroy = api.comm_destroy; cb.error_string = api.error_string;
    p->connected = true;
    return GMG_OK;
} GMG_CATCH_H

// Every rank loads the whole right-hand side and initial guess (host, natural numbering), like gmg_load_problem.
int gmg_p2p_load(gmg_handle h, const double* b, const double* x0) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected) return fail(h, GMG_ERR_STATE, "call gmg_p2p_prepare / gmg_p2p_connect first");
    unbind_level0(h);
    // a time-out of an earlier problem (rank skew at first use, a peer that came late) must not poison this one: the exchange
    // kernels give up at once while the word is set
    HIPCHK(hipMemsetAsync(p->d_err, 0, sizeof(int), h->stream));
    int rc = gmg_load_problem(h, b, x0, p->d);
    if (rc) return rc;
    // the distributed steps address the level-0 vectors through the `bound` state of the gmg_dist_* entry points
    Level& l = h->lv[0];
    h->own_x0 = l.x; h->own_b0 = l.b; h->own_r0 = l.r;
    h->bound = true;
    return GMG_OK;
} GMG_CATCH_H

namespace {

// Level-0 sweeps on this rank's rows.  Default: an exchange after every colour -- colours are global, so the iterates are those of the
// single-GPU engine whatever the number of ranks is.  Hybrid (gmg_p2p_set_smoother): the colours of a sweep run back to
// back on the rank's rows with the peers' values of the PREVIOUS sweep, then ONE exchange of the halo of all colours -- Gauss-Seidel
// inside a rank, Jacobi across ranks: a quarter of the exchanges at four colours, iterates (and possibly the cycle count) depend on P.
int p2p_smooth(gmg_handle h, int iters) {
    Level& l = h->lv[0];
    const bool hybrid = h->p2p->hybrid;
    int rc;
    if (l.ord.blocked) {
        // a blocked level 0 (gmg_config::block_fine: kNN operators -- one launch per sweep on one GPU too): this rank's run of whole 64-row blocks,
        // one halo exchange per sweep.  The block-hybrid sweep takes the couplings to other blocks from the previous iterate, so the iterates are
        // those of the single-GPU engine whatever the number of ranks is, exactly like the partitioned level 1 below.
        int sb, se;
        own_range(h, 0, sb, se);                           // slices of 64 rows = blocks
        const int d = h->p2p->d;
        double* in = l.x;
        double* out = l.tmp;
        for (int it = 0; it < iters; ++it) {
            launch_block_sweep_range<double>(h, l, d, in, out, sb, se - sb);
            if ((rc = p2p_exchange(h, 1, out, l.n_pad))) return rc;      // kind 1 of a one-class level: the halo of all rows
            std::swap(in, out);
        }
        if (in != l.x) HIPCHK(hipMemcpyAsync(l.x, in, sizeof(double) * (size_t)l.n_pad * d, hipMemcpyDeviceToDevice, h->stream));
        return GMG_OK;
    }
    for (int it = 0; it < iters; ++it) {
        for (int c = 0; c < l.ord.n_colors; ++c) {
            if (p2p_smooth_color_folded(h, c)) continue;
            if ((rc = dist_smooth_color_impl(h, c, hybrid ? h->p2p->d_plain_rows : nullptr))) return rc;
            if (!hybrid && (rc = p2p_exchange(h, c, l.x, l.n_pad))) return rc;
        }
        if (hybrid && (rc = p2p_exchange(h, l.ord.n_colors, l.x, l.n_pad))) return rc;
    }
    return GMG_OK;
}

// `iters` block sweeps on this rank's blocks of level 1, each followed by the x1 halo exchange; the result ends in l1.x
// (from_zero: the iterate is the zero vector and the first sweep takes no input, like launch_block_sweeps).
// first_fused: the first (from-zero) sweep already ran inside the restriction's launch (restrict_sweep0 on this rank's blocks): its result is in tmp
int p2p_smooth_level1(gmg_handle h, int iters, bool from_zero, bool first_fused = false) {
    DistP2P* p = h->p2p;
    Level& l1 = h->lv[1];
    const int C = dist_classes(h->lv[0].ord), d = p->d, nb = (int)p->own_blocks[p->rank].size();
    double* in = from_zero ? nullptr : l1.x;
    double* out = l1.tmp;
    int rc;
    const double* before_last = nullptr;
    for (int it = 0; it < iters; ++it) {
        if (!(it == 0 && first_fused)) launch_block_sweep_range<double>(h, l1, d, in, out, 0, nb, p->d_own_begin, p->d_own_ncolors);
        if ((rc = p2p_exchange(h, C + 3, out, l1.n_pad))) return rc;
        before_last = in;
        if (it == 0 && from_zero) { in = out; out = l1.x; }
        else std::swap(in, out);
    }
    if (iters > 0 && in != l1.x) HIPCHK(hipMemcpyAsync(l1.x, in, sizeof(double) * (size_t)l1.n_pad * d, hipMemcpyDeviceToDevice, h->stream));
    // (both buffers carry the halo entries this rank's rows read: each was exchanged right after the sweep that wrote it -- what the
    // residual from the sweep's explicit part needs, launch_residual_delta, exactly as the single-GPU engine forms it)
    h->sweep_prev_valid = iters > 0 && before_last != l1.x;
    h->sweep_prev = (const void*)before_last;
    return GMG_OK;
}

// Levels >= 1 with level 1 partitioned by blocks: b1 = U0^T r0 on this rank's rows, sweeps with one halo exchange each, r1
// completed on every rank, levels >= 2 replicated, and back up to x1 (own rows + halo) for the level-0 prolongation.
int p2p_coarse_cycle_sharded(gmg_handle h) {
    DistP2P* p = h->p2p;
    Level &l0 = h->lv[0], &l1 = h->lv[1];
    const int C = dist_classes(l0.ord), d = p->d;
    int rc;
    if ((rc = p2p_exchange(h, C + 5, l0.r, l0.n_pad))) return rc;                      // r0 entries my restriction rows read
    const bool from_zero = smooth_from_zero_ok(h, l1, h->cfg.pre_iters);
    // :1069 on my rows of level 1 -- together with the first pre-sweep of my blocks where the layouts allow it (restrict_sweep0 over this rank's
    // block list: a workgroup's four restriction slices are one block; same bits as the two launches)
    const bool fuse = from_zero && p->n_rsl > 0 && d <= 4 && restrict_sweep0_kind<double>(h, l0, l1, d, false) == 1 && l0.R.lpr == 4 && p->n_rsl == 4 * (int)p->own_blocks[p->rank].size();
    if (fuse) {
        const int nbk = (int)p->own_blocks[p->rank].size();
        const int vgrid = (nbk + 7) / 8 * 8;
        const size_t lds_sweep = gmgk::ep_lds_bytes<double>(d, 0, l1.ep_cap_l);
        const size_t lds = lds_sweep + (size_t)d * 64 * sizeof(double);
        DISPATCH_D(d, DISPATCH_C16(l0.R.c16_sel(), {
            if (ep_streams(l1))
                hipLaunchKernelGGL((gmgk::restrict_sweep0<double, D, true, C16, 0>), dim3(vgrid), dim3(256), lds, h->stream, l0.R.slice_ptr, l0.R.col, l0.R.val, l0.R.row_of, l0.r, l0.n_pad,
                                   l0.R.col16, l0.R.win_base, l0.R.c16_arg(), l1.b, l1.d_blk_ncolors, l1.d_row_color, l1.ep_ptr, l1.ep_col, l1.ep_val, l1.diag, l1.tmp, l1.n_pad, nbk, vgrid,
                                   (int)lds_sweep, (const int*)p->d_own_blocks);
            else
                hipLaunchKernelGGL((gmgk::restrict_sweep0<double, D, false, C16, 0>), dim3(vgrid), dim3(256), lds, h->stream, l0.R.slice_ptr, l0.R.col, l0.R.val, l0.R.row_of, l0.r, l0.n_pad,
                                   l0.R.col16, l0.R.win_base, l0.R.c16_arg(), l1.b, l1.d_blk_ncolors, l1.d_row_color, l1.ep_ptr, l1.ep_col, l1.ep_val, l1.diag, l1.tmp, l1.n_pad, nbk, vgrid,
                                   (int)lds_sweep, (const int*)p->d_own_blocks);
        }));
    } else if (p->n_rsl > 0)
        for (int c0 = 0; c0 < d; c0 += 4) {
            int dc = std::min(4, d - c0);
            if (l0.R.lpr == 4) {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::transfer_list<double, D, 0, 4>), dim3(grid_for(p->n_rsl)), dim3(gmgk::kBlock), 0, h->stream, l0.R.slice_ptr,
                                                  l0.R.col, l0.R.val, l0.R.row_of, l0.r + (size_t)c0 * l0.n_pad, l0.n_pad, l1.b + (size_t)c0 * l1.n_pad, l1.n_pad,
                                                  p->d_rsl, p->n_rsl));
            } else {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::transfer_list<double, D, 0, 1>), dim3(grid_for(p->n_rsl)), dim3(gmgk::kBlock), 0, h->stream, l0.R.slice_ptr,
                                                  l0.R.col, l0.R.val, l0.R.row_of, l0.r + (size_t)c0 * l0.n_pad, l0.n_pad, l1.b + (size_t)c0 * l1.n_pad, l1.n_pad,
                                                  p->d_rsl, p->n_rsl));
            }
        }
    if (!from_zero) HIPCHK(hipMemsetAsync(l1.x, 0, sizeof(double) * (size_t)l1.n_pad * d, h->stream));       // :1072-1073
    if ((rc = p2p_smooth_level1(h, h->cfg.pre_iters, from_zero, fuse))) return rc;        // :1063
    const bool from_sweep = launch_residual_delta<double>(h, l1, d, l1.r, p->d_own_begin, (int)p->own_blocks[p->rank].size());      // :1066 on my rows ...
    if (!from_sweep && p->n_asl > 0)                                                  // ... or with the residual SpMV
        for (int c0 = 0; c0 < d; c0 += 4) {
            int dc = std::min(4, d - c0);
            if (l1.Aoff.lpr == 4) {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::spmv_full_list<double, D, 1, 4>), dim3(grid_for(p->n_asl)), dim3(gmgk::kBlock), 0, h->stream, l1.Aoff.slice_ptr,
                                                  l1.Aoff.col, l1.Aoff.val, l1.diag, l1.b + (size_t)c0 * l1.n_pad, l1.x + (size_t)c0 * l1.n_pad,
                                                  l1.r + (size_t)c0 * l1.n_pad, l1.n_pad, p->d_asl, p->n_asl));
            } else {
                DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::spmv_full_list<double, D, 1, 1>), dim3(grid_for(p->n_asl)), dim3(gmgk::kBlock), 0, h->stream, l1.Aoff.slice_ptr,
                                                  l1.Aoff.col, l1.Aoff.val, l1.diag, l1.b + (size_t)c0 * l1.n_pad, l1.x + (size_t)c0 * l1.n_pad,
                                                  l1.r + (size_t)c0 * l1.n_pad, l1.n_pad, p->d_asl, p->n_asl));
            }
        }
    if ((rc = p2p_exchange(h, C + 4, l1.r, l1.n_pad))) return rc;                      // everybody's rows -> complete r1 on every rank
    h->first_sweep_fused = false;
    restrict_into<double>(h, 1, d, false);                                            // :1069, replicated from here down (+ level 2's first sweep where fused)
    enqueue_down<double>(h, d, 2);
    if (h->coarse_device) enqueue_coarse_device<double>(h, d);
    else if ((rc = coarse_host_begin<double>(h, d))) return rc;       // the host half is served at the end of the cycle's enqueue (p2p_vcycle)
    enqueue_up<double>(h, d, 2);
    for (int c0 = 0; c0 < d && p->n_psl > 0; c0 += 4) {                               // :1082 into my rows of level 1
        int dc = std::min(4, d - c0);
        DISPATCH_D(dc, hipLaunchKernelGGL((gmgk::transfer_list<double, D, 1, 1>), dim3(grid_for(p->n_psl)), dim3(gmgk::kBlock), 0, h->stream, l1.P.slice_ptr, l1.P.col,
                                          l1.P.val, (const int*)nullptr, h->lv[2].x + (size_t)c0 * h->lv[2].n_pad, h->lv[2].n_pad, l1.x + (size_t)c0 * l1.n_pad,
                                          l1.n_pad, p->d_psl, p->n_psl));
    }
    if ((rc = p2p_exchange(h, C + 3, l1.x, l1.n_pad))) return rc;
    return p2p_smooth_level1(h, h->cfg.post_iters, false);                            // :1085; leaves x1 current on my rows + halo
}

int p2p_vcycle_enqueue(gmg_handle h) {
    Level& l = h->lv[0];
    const int C = dist_classes(l.ord);
    int rc;
    if ((rc = p2p_smooth(h, h->cfg.pre_iters))) return rc;                // :1063
    if ((rc = gmg_dist_residual_own(h))) return rc;                       // :1066, own rows
    if (h->p2p->shard1) {
        if ((rc = p2p_coarse_cycle_sharded(h))) return rc;
    } else {
        if ((rc = p2p_exchange(h, C + 1, l.r, l.n_pad))) return rc;       //        everybody's rows -> complete r on every rank
        if ((rc = dist_coarse_cycle_enqueue(h))) return rc;               // :1069-1079, replicated
    }
    if ((rc = gmg_dist_prolong_own(h))) return rc;                        // :1082, own rows
    if ((rc = p2p_exchange(h, C, l.x, l.n_pad))) return rc;               //        halo of all colours
    return p2p_smooth(h, h->cfg.post_iters);                              // :1085
}

// the whole cycle is queued before the host turns to the coarsest solve: the way up waits in the stream for its answer
int p2p_vcycle(gmg_handle h) {
    const int rc = p2p_vcycle_enqueue(h);
    const int served = coarse_host_serve(h);
    return rc ? rc : served;
}

// exchange kind by name: "color<k>", "halo_all", "rows0", "x1_halo", "rows1", "r0_halo" -> kind, vector and its leading dimension
int p2p_kind_by_name(gmg_handle h, const std::string& name, int* kind, double** vec, int* ld) {
    DistP2P* p = h->p2p;
    Level& l0 = h->lv[0];
    const int C = dist_classes(l0.ord);
    *vec = l0.x; *ld = l0.n_pad;
    if (name.rfind("color", 0) == 0) { *kind = std::atoi(name.c_str() + 5); return *kind >= 0 && *kind < C ? GMG_OK : GMG_ERR_INVALID; }
    if (name == "halo_all") { *kind = C; return GMG_OK; }
    if (name == "rows0") { *kind = C + 1; *vec = l0.r; return GMG_OK; }
    if (!p->shard1) return GMG_ERR_INVALID;
    if (name == "x1_halo") { *kind = C + 3; *vec = h->lv[1].tmp; *ld = h->lv[1].n_pad; return GMG_OK; }
    if (name == "rows1") { *kind = C + 4; *vec = h->lv[1].r; *ld = h->lv[1].n_pad; return GMG_OK; }
    if (name == "r0_halo") { *kind = C + 5; *vec = l0.r; return GMG_OK; }
    return GMG_ERR_INVALID;
}

}  // namespace

// n V-cycles, each followed by the residual check (stop_type >= 0): every rank calls this with the same arguments.
int gmg_p2p_cycles(gmg_handle h, int n_cycles, int stop_type, double* residues) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected || !h->bound) return fail(h, GMG_ERR_STATE, "no distributed problem loaded (gmg_p2p_load)");
    int rc;
    if (stop_type >= 0 && (rc = check_norm_type(h, stop_type))) return rc;
    const int d = p->d, np = (int)p->peers.size(), C = dist_classes(h->lv[0].ord);
    HelperScope helper_scope(h, d);
    for (int i = 0; i < n_cycles; ++i) {
        if ((rc = p2p_vcycle(h))) return rc;
        if (stop_type < 0) continue;
        double sums[8];
        if ((rc = dist_norm_launch(h, stop_type))) return rc;                // this rank's rows -> h->d_norm
        const double* d_result = h->d_norm;
        if (p->coll.mode != 0) {
            // collective backend: every rank's 2 d sums all-gathered, then added in rank order (the same bits everywhere)
            CollBackend& cb = p->coll;
            const int kind = C + 2, parity = (int)(cb.count++ & 1);
            ++p->kind_count[kind];
            ++p->seq;
            HIPCHK(hipMemcpyAsync(cb.send, h->d_norm, sizeof(double) * 2 * d, hipMemcpyDeviceToDevice, h->stream));
            if ((rc = coll_all_gather(h, kind, parity))) return rc;
            const double* gathered = cb.recv + (size_t)parity * p->world * cb.max_chunk;
            if (cb.mode == 2) hipLaunchKernelGGL(gmgk::coll_sum_ranks<true>, dim3(1), dim3(64), 0, h->stream, gathered, cb.chunk[kind], p->world, p->rank, (const double*)h->d_norm, 2 * d, p->d_sums);
            else hipLaunchKernelGGL(gmgk::coll_sum_ranks<false>, dim3(1), dim3(64), 0, h->stream, gathered, cb.chunk[kind], p->world, p->rank, (const double*)h->d_norm, 2 * d, p->d_sums);
            d_result = p->d_sums;
        } else if (np > 0) {
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
    // the level-0 rows exchange moves every rank's own rows of a level-0 vector
    int rc = p2p_exchange(h, dist_classes(l.ord) + 1, l.x, l.n_pad);
    if (rc) return rc;
    int herr = 0;
    HIPCHK(hipMemcpyAsync(&herr, p->d_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (herr) return fail(h, GMG_ERR_STATE, "a peer-to-peer exchange timed out (a rank is missing or ran a different sequence): x is incomplete");
    return to_host(h, 0, l.x, p->d, x);
} GMG_CATCH_H

// gmg_solve over the ranks: x0 in, solution out (complete on every rank).  The reference's loop, multigrid_solver.cpp:1408-1419:
// do { V-cycle; residualCheck } while (residue > tol && it < maxIter) -- every rank sees the same residues (the norm sums are
// all-reduced in rank order), so every rank leaves the loop in the same iteration and takes the same decision below.  Collective.
// The safeguards of the single-GPU solve (engine.hip::solve_common) apply unchanged: the loop stops at a residue that is not
// finite or 1e4 x the smallest seen; an iteration that ends above the tolerance with such a residue, or with one larger than
// after its first cycle, returns GMG_DIVERGED (x still receives the last iterate, as in the reference) and the caller can repeat
// the solve with Gauss-Seidel in colour order on every level (block_rows = 0, gs_omega = 1).  Timing keys as in gmg_solve.
int gmg_p2p_solve(gmg_handle h, const double* b, double* x, double tol, int stop_type, int max_iter, int* iters_out, double* residue_out) try {
    if (!h) return GMG_ERR_INVALID;
    if (!b || !x) return fail(h, GMG_ERR_INVALID, "bad arguments");
    if (max_iter < 1) max_iter = 1;      // do { } while: at least one cycle
    auto t_all = clk::now();
    int rc = gmg_p2p_load(h, b, x);
    if (rc) return rc;
    h->timing["solve_load"] = ms_since(t_all);
    auto t0 = clk::now();
    int it = 0;
    double residue = 0.0, first_residue = 0.0, least_residue = 0.0;
    bool blown = false;
    do {
        if ((rc = gmg_p2p_cycles(h, 1, stop_type, &residue))) return rc;
        if (it == 0) first_residue = least_residue = residue;
        if (residue < least_residue) least_residue = residue;
        ++it;
        blown = !std::isfinite(residue) || (it >= 3 && residue > 1e4 * least_residue);
    } while (residue > tol && it < max_iter && !blown);
    h->timing["cycles"] = ms_since(t0);
    const bool diverged = !(residue <= tol) && (blown || (it > 1 && residue > first_residue));
    h->timing["diverged"] = diverged ? 1.0 : 0.0;
    h->timing["blown_up"] = blown ? 1.0 : 0.0;
    h->timing["iterations"] = it;
    h->timing["residue"] = residue;
    if (iters_out) *iters_out = it;
    if (residue_out) *residue_out = residue;
    rc = gmg_p2p_fetch(h, x);
    h->timing["solve_call"] = ms_since(t_all);
    if (rc) return rc;
    return diverged ? GMG_DIVERGED : GMG_OK;
} GMG_CATCH_H

// Average duration (ms) of one exchange of the named kind (push + wait + pull, one launch), `reps` back to back; collective --
// every rank calls it with the same arguments.  Kinds: "color<k>" (halo of colour k), "halo_all", "rows0" (every rank's rows of
// a level-0 vector), and with level 1 partitioned "x1_halo", "rows1", "r0_halo".  The values moved are whatever the vectors
// hold: call it between problems (gmg_p2p_load afterwards).
int gmg_p2p_bench_kind(gmg_handle h, const char* kind_name, int reps, double* ms_avg) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected || !h->bound || !ms_avg || !kind_name || reps <= 0) return fail(h, GMG_ERR_STATE, "no distributed problem loaded (gmg_p2p_load)");
    int kind = 0, ld = 0;
    double* vec = nullptr;
    if (p2p_kind_by_name(h, kind_name, &kind, &vec, &ld)) return fail(h, GMG_ERR_INVALID, std::string("unknown exchange kind: ") + kind_name);
    for (int i = 0; i < 3; ++i) (void)p2p_exchange(h, kind, vec, ld);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < reps; ++i) (void)p2p_exchange(h, kind, vec, ld);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_avg = (double)ms / reps;
    int herr = 0;
    HIPCHK(hipMemcpy(&herr, p->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (herr) return fail(h, GMG_ERR_STATE, "a peer-to-peer exchange timed out (a rank is missing or ran a different sequence)");
    return GMG_OK;
} GMG_CATCH_H

// 0 (default): exact multicolour Gauss-Seidel on level 0, one exchange per colour (iterates independent of the rank count);
// 1: hybrid -- Gauss-Seidel inside a rank, Jacobi across ranks, ONE exchange per sweep.  Collective in the sense that every rank must choose the same.
int gmg_p2p_set_smoother(gmg_handle h, int hybrid) try {
    if (!h || !h->p2p) return h ? fail(h, GMG_ERR_STATE, "no distributed plan (gmg_p2p_prepare)") : GMG_ERR_INVALID;
    if (hybrid < 0 || hybrid > 2) return fail(h, GMG_ERR_INVALID, "smoother must be 0 (exact), 1 (hybrid) or 2 (exact, exchange folded into the colour launches)");
    h->p2p->hybrid = hybrid == 1;
    h->p2p->fold = hybrid == 2;
    return GMG_OK;
} GMG_CATCH_H

// Test hook (include/gravomg_hip_internal.h): one exchange of every rank's level-0 rows of x through the collective backend, then the gathered
// buffer's slot of THIS rank against what it packed -- what ncclAllGather delivered, seen from the host.
int gmg_p2p_debug_collective_roundtrip(gmg_handle h, double* max_abs_diff, long long* doubles) try {
    NEED_DEVICE();
    DistP2P* p = h->p2p;
    if (!p || !p->connected || !h->bound || !max_abs_diff || !doubles) return fail(h, GMG_ERR_STATE, "no distributed problem loaded (gmg_p2p_load)");
    CollBackend& cb = p->coll;
    if (cb.mode == 0) return fail(h, GMG_ERR_STATE, "this handle has no collective backend (gmg_config::dist_exchange)");
    Level& l = h->lv[0];
    const int kind = dist_classes(l.ord) + 1;
    const int parity = (int)(cb.count & 1);                               // the half coll_exchange is about to use
    int rc = coll_exchange(h, kind, l.x, l.n_pad);
    if (rc) return rc;
    const long long n = (long long)(l.n_pad / p->world) * p->d;
    std::vector<double> sent((size_t)n), got((size_t)n);
    HIPCHK(hipMemcpyAsync(sent.data(), cb.send, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(got.data(), cb.recv + ((size_t)parity * p->world + p->rank) * cb.max_chunk, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    double worst = 0.0;
    for (long long i = 0; i < n; ++i) worst = std::max(worst, std::fabs(sent[(size_t)i] - got[(size_t)i]));
    *max_abs_diff = worst; *doubles = n;
    return GMG_OK;
} GMG_CATCH_H

int gmg_p2p_set_fences(gmg_handle h, int fenced) try {
    if (!h || !h->p2p) return h ? fail(h, GMG_ERR_STATE, "no distributed plan (gmg_p2p_prepare)") : GMG_ERR_INVALID;
    h->p2p->fenced = fenced != 0;
    return GMG_OK;
} GMG_CATCH_H

int gmg_p2p_stat(gmg_handle h, const char* key, double* out) try {
    if (!h || !h->p2p || !key || !out) return GMG_ERR_INVALID;
    if (std::string(key) == "fenced") { *out = h->p2p->fenced ? 1.0 : 0.0; return GMG_OK; }
    if (std::string(key) == "collective_mode") { *out = (double)h->p2p->coll.mode; return GMG_OK; }
    if (std::string(key) == "collective_exchanges") { *out = (double)h->p2p->coll.count; return GMG_OK; }
    if (std::string(key) == "device_bytes") { *out = (double)h->pool.live_bytes; return GMG_OK; }      // device memory this rank's handle holds (pool blocks in use)
    if (std::string(key) == "exchange_launches") { *out = (double)h->p2p->exchange_launches; return GMG_OK; }      // launches spent on exchanges so far (the folded colour exchanges: none)
    if (std::string(key) == "device_bytes_peak") { *out = (double)h->pool.peak_live_bytes; return GMG_OK; }      // ... and its high-water mark since the last gmg_set_system began
    auto it = h->p2p->stats.find(key);
    if (it == h->p2p->stats.end()) return fail(h, GMG_ERR_INVALID, std::string("unknown key: ") + key);
    *out = it->second;
    return GMG_OK;
} GMG_CATCH_H

}  // extern "C"
