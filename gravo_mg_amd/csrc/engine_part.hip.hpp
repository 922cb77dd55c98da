// engine_part.hip.hpp -- part of libgravomg_hip.so's single translation unit (included by engine.hip after engine_cycle, before the entry points).
// The partition plan of a multi-GPU job (one process per GPU): who owns which rows of levels 0 / 1 and which entries every rank
// publishes to every other rank.  Computed identically on every rank -- from the orderings, the level-0 / level-1 sparsity patterns and U_0 --
// so that mailbox layouts need no negotiation (engine_dist.hip.hpp), and early enough for a PARTITIONED set-up (gmg_dist_partition) to lay out
// this rank's rows only: the masks below turn everybody else's rows into "padding rows" of the layout builders.
#pragma once

namespace {

inline int p2p_owner(const LevelOrdering& o, int world, int row, int* colour_out) {
    int c = (int)(std::upper_bound(o.color_begin.begin(), o.color_begin.end(), row) - o.color_begin.begin()) - 1;
    const int piece = (o.color_begin[c + 1] - o.color_begin[c]) / world;
    if (colour_out) *colour_out = c;
    return piece > 0 ? (row - o.color_begin[c]) / piece : 0;
}

// Can level 1 be partitioned by blocks?  It must run the entry-parallel block sweep (big level: one lane per row, 64-row blocks, block b =
// rows 64 b ..) with replicated levels below it, and the restriction's sorting windows must not straddle blocks.  `use_ep_known`: the level's
// layout exists and says whether its blocks fit the sweep's LDS (l1.use_ep); before the layout the configuration decides, and the caller
// re-checks afterwards.
inline bool plan_can_shard_level1(gmg_handle h, int world, bool use_ep_known) {
    if (!(world > 1 && h->cfg.dist_shard_levels >= 2 && h->L >= 2 && h->cfg.smoother == GMG_SMOOTHER_MULTICOLOR_GS &&
          (h->cfg.restrict_sigma == 0 || h->cfg.restrict_sigma == 64))) return false;
    const Level& l1 = h->lv[1];
    if (!l1.ord.blocked) return false;
    if (use_ep_known) { if (!l1.use_ep) return false; }
    else {
        const int lpr = h->cfg.block_lanes ? h->cfg.block_lanes : (l1.n < kQuadLevelRows ? 4 : 1);
        if (!wants_block_ep(h, lpr)) return false;
    }
    const LevelOrdering& o1 = l1.ord;
    for (int b = 0; b <= o1.n_blocks(); ++b) if (o1.blk_begin[b] != 64 * b) return false;
    return true;
}

// A0: the level-0 operator's pattern in natural numbering (the caller's arrays or a host copy); A1: level 1's (needed with shard1 only).
// Threaded; the lists come out sorted and duplicate-free, so the result does not depend on the number of threads.
int build_dist_plan(gmg_handle h, DistPlan& P, int rank, int world, const PatternView& A0, const Compressed* A1, bool shard1) {
    Level& l = h->lv[0];
    const LevelOrdering& o = l.ord;
    const int C = dist_classes(o);
    P = DistPlan();
    P.rank = rank; P.world = world; P.n_colors = C; P.shard1 = shard1 && world > 1;
    P.halo.assign((size_t)world * world * (C + 1), std::vector<int>());
    P.halo1.assign((size_t)world * world, std::vector<int>());
    P.halo0r.assign((size_t)world * world, std::vector<int>());
    if (world <= 1) return GMG_OK;
    auto sort_unique = [](std::vector<int>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    const int T = std::max(1, std::min(h->cfg.host_threads, 16));
    const size_t W2 = (size_t)world * world;
    using Lists = std::vector<std::vector<int>>;
    auto merge = [&](std::vector<Lists>& part, size_t slot, std::vector<int>& out) {
        size_t total = 0;
        for (auto& pw : part) total += pw[slot].size();
        out.clear(); out.reserve(total);
        for (auto& pw : part) { out.insert(out.end(), pw[slot].begin(), pw[slot].end()); std::vector<int>().swap(pw[slot]); }
        sort_unique(out);
    };
    // ---- level 0: for every row r (owner t) and every entry (r, c) with owner(c) = s != t, s publishes c to t
    std::vector<int> owner(l.n_pad);
    std::vector<unsigned char> colour(l.n_pad);
    parallel_ranges(l.n_pad, h->cfg.host_threads, [&](int lo, int hi, int) { for (int r = lo; r < hi; ++r) { int c; owner[r] = p2p_owner(o, world, r, &c); colour[r] = (unsigned char)c; } });
    {
        std::vector<Lists> part(T, Lists(W2));
        parallel_ranges(l.n, T, [&](int lo, int hi, int t) {
            Lists& mine = part[std::min(t, T - 1)];
            for (int i = lo; i < hi; ++i) {
                const int tr = owner[o.old2new[i]];
                for (int q = A0.ptr[i]; q < A0.ptr[i + 1]; ++q) {
                    const int c = o.old2new[A0.idx[q]], s = owner[c];
                    if (s != tr) mine[(size_t)s * world + tr].push_back(c);
                }
            }
        }, 1);
        std::vector<int> all;
        for (int s = 0; s < world; ++s)
            for (int t = 0; t < world; ++t) {
                if (s == t) continue;
                merge(part, (size_t)s * world + t, all);
                for (int c : all) P.halo[((size_t)s * world + t) * (C + 1) + colour[c]].push_back(c);
                P.halo[((size_t)s * world + t) * (C + 1) + C] = all;
            }
    }
    if (!P.shard1) return GMG_OK;
    // ---- level 1 by blocks: a block -> the rank that owns most of the fine rows its points prolong into (ties: the lowest rank)
    if (!A1) return fail(h, GMG_ERR_STATE, "partition plan: the level-1 pattern is missing");
    Level& l1 = h->lv[1];
    const LevelOrdering& o1 = l1.ord;
    const Compressed& U0 = h->U[0];                       // CSC: one column per coarse point, rows = fine points
    const int nb = o1.n_blocks();
    {
        std::vector<std::vector<int>> votes(T, std::vector<int>((size_t)nb * world, 0));
        parallel_ranges(U0.n_outer, T, [&](int lo, int hi, int t) {
            std::vector<int>& v = votes[std::min(t, T - 1)];
            for (int jc = lo; jc < hi; ++jc) {
                const int blk = o1.old2new[jc] >> 6;
                for (int q = U0.ptr[jc]; q < U0.ptr[jc + 1]; ++q) ++v[(size_t)blk * world + owner[o.old2new[U0.idx[q]]]];
            }
        }, 1);
        P.blk_owner.resize(nb);
        P.own_blocks.assign(world, std::vector<int>());
        for (int b = 0; b < nb; ++b) {
            int best = 0; long best_votes = -1;
            for (int r = 0; r < world; ++r) { long s = 0; for (int t = 0; t < T; ++t) s += votes[t][(size_t)b * world + r]; if (s > best_votes) { best_votes = s; best = r; } }
            P.blk_owner[b] = best;
            P.own_blocks[best].push_back(b);
        }
    }
    auto owner1 = [&](int row) { return P.blk_owner[row >> 6]; };
    // x1 entries read through A1 (sweeps, residual) or through U0 (prolongation into another rank's fine rows), and r0 entries read
    // through U0^T (restriction into another rank's coarse rows)
    {
        std::vector<Lists> p1(T, Lists(W2)), p0r(T, Lists(W2));
        parallel_ranges(l1.n, T, [&](int lo, int hi, int t) {
            Lists& mine = p1[std::min(t, T - 1)];
            for (int i = lo; i < hi; ++i) {
                const int tr = owner1(o1.old2new[i]);
                for (int q = A1->ptr[i]; q < A1->ptr[i + 1]; ++q) {
                    const int c = o1.old2new[A1->idx[q]], s = owner1(c);
                    if (s != tr) mine[(size_t)s * world + tr].push_back(c);
                }
            }
        }, 1);
        parallel_ranges(U0.n_outer, T, [&](int lo, int hi, int t) {
            Lists& m1 = p1[std::min(t, T - 1)];
            Lists& m0 = p0r[std::min(t, T - 1)];
            for (int jc = lo; jc < hi; ++jc) {
                const int c = o1.old2new[jc], s1 = owner1(c);
                for (int q = U0.ptr[jc]; q < U0.ptr[jc + 1]; ++q) {
                    const int rf = o.old2new[U0.idx[q]], t0 = owner[rf];
                    if (t0 == s1) continue;
                    m1[(size_t)s1 * world + t0].push_back(c);        // rank t0 prolongs into fine row rf: reads x1[c]
                    m0[(size_t)t0 * world + s1].push_back(rf);       // rank s1 restricts into coarse row c: reads r0[rf]
                }
            }
        }, 1);
        for (size_t slot = 0; slot < W2; ++slot) { merge(p1, slot, P.halo1[slot]); merge(p0r, slot, P.halo0r[slot]); }
    }
    return GMG_OK;
}

// Masked row maps of a partitioned set-up: new2old of level 0 / level 1 with -1 for the rows of other ranks (gmgs::mask_rows_*).
int make_row_masks(gmg_handle h, const DistPlan& P, int** d_mask0, int** d_mask1) {
    Level& l0 = h->lv[0];
    const LevelOrdering& o = l0.ord;
    DevTmp<int> d_cb;
    int rc;
    if ((rc = d_cb.alloc(h, (size_t)dist_classes(o) + 1))) return rc;
    HIPCHK(hipMemcpyAsync(d_cb.p, o.color_begin.data(), sizeof(int) * ((size_t)dist_classes(o) + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(dev_malloc((void**)d_mask0, sizeof(int) * (size_t)l0.n_pad));
    hipLaunchKernelGGL(gmgs::mask_rows_by_colour, dim3((l0.n_pad + 255) / 256), dim3(256), 0, h->stream, l0.d_new2old, l0.n_pad, d_cb.p, dist_classes(o), P.world, P.rank, *d_mask0);
    if (P.shard1) {
        Level& l1 = h->lv[1];
        DevTmp<int> d_own;
        if ((rc = d_own.alloc(h, P.blk_owner.size()))) return rc;
        HIPCHK(hipMemcpyAsync(d_own.p, P.blk_owner.data(), sizeof(int) * P.blk_owner.size(), hipMemcpyHostToDevice, h->stream));
        HIPCHK(dev_malloc((void**)d_mask1, sizeof(int) * (size_t)l1.n_pad));
        hipLaunchKernelGGL(gmgs::mask_rows_by_block, dim3((l1.n_pad + 255) / 256), dim3(256), 0, h->stream, l1.d_new2old, l1.n_pad, d_own.p, P.rank, *d_mask1);
        HIPCHK(hipStreamSynchronize(h->stream));      // (the small host arrays above are pageable; d_own / d_cb die at scope end)
    } else HIPCHK(hipStreamSynchronize(h->stream));
    return GMG_OK;
}

}  // namespace
