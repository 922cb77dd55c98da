"""Pure-Python restatement of the reference's default hierarchy construction -- TEST INFRASTRUCTURE ONLY.

Follows gravomg/src/multigrid_solver.cpp line by line (FASTDISK sampling, BARYCENTRIC weights, check_voronoi, not nested):
    constructProlongation :62-469      computeAverageEdgeLength :695-711      fastDiskSample :975-1013
    constructDijkstraWithCluster :1015-1056      inTriangle :471-507      inverseDistanceWeights :515-526
It shares no code with gravo_mg_amd/csrc/host_hierarchy.hpp (the product's C++ builder) and is only practical for a
few thousand vertices; tests/test_hierarchy_restatement.py compares the two entry for entry.
Parity status: unpinned by the reference (it has no tests and cannot be built here); two independent restatements
agreeing is what pins the builder.
"""
import heapq
import math

import numpy as np
import scipy.sparse as sp


def _norm(v):
    return math.sqrt(float(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]))


def _normalized(v):
    z = float(v @ v)
    return v / math.sqrt(z) if z > 0 else v


def average_edge_length(pos, neigh):                       # :695-711
    s, cnt = 0.0, 0
    for i in range(pos.shape[0]):
        for j in range(neigh.shape[1]):
            g = neigh[i, j]
            if g < 0:
                continue
            d = _norm(pos[i] - pos[g])
            if d > 0:
                s += d
                cnt += 1
    return s / cnt


def fast_disk_sample(pos, neigh, radius, D, nearest):     # :975-1013
    n, K = neigh.shape
    visited = np.zeros(n, bool)
    sel = []
    for i in range(n):
        if visited[i]:
            continue
        sidx = len(sel)
        sel.append(i)
        nearest[i] = sidx
        for j in range(K):
            g = neigh[i, j]
            if g < 0:
                break
            d1 = _norm(pos[i] - pos[g])
            if d1 < radius:
                visited[g] = True
                if d1 < D[g]:
                    D[g] = d1
                    nearest[g] = sidx
                for j2 in range(K):
                    g2 = neigh[g, j2]
                    if g2 < 0:
                        break
                    d2 = d1 + _norm(pos[g] - pos[g2])
                    if d2 < radius:
                        visited[g2] = True
                        if d2 < D[g2]:
                            D[g2] = d2
                            nearest[g2] = sidx
    return sel


def dijkstra_with_cluster(pos, sources, neigh, D, nearest):   # :1015-1056 (no stale-entry check upstream either)
    heap = []
    for i, s in enumerate(sources):
        D[s] = 0.0
        heapq.heappush(heap, (0.0, len(heap), s))
        nearest[s] = i
    tick = len(heap)
    while heap:
        dist, _, v = heapq.heappop(heap)
        owner = nearest[v]
        for j in range(neigh.shape[1]):
            g = neigh[v, j]
            if g < 0:
                continue
            cand = dist + _norm(pos[g] - pos[v])
            if cand < D[g]:
                D[g] = cand
                tick += 1
                heapq.heappush(heap, (cand, tick, g))
                nearest[g] = owner


def in_triangle(p, tri, nrm, pos, inside_edge):            # :471-507
    v1, v2, v3 = pos[tri[0]], pos[tri[1]], pos[tri[2]]
    v1p, e12, e13 = p - v1, v2 - v1, v3 - v1
    plane = float((p - v1) @ nrm)
    q = p - plane * nrm
    area2 = float(np.cross(v2 - v1, v3 - v1) @ nrm)
    b0 = float(np.cross(v3 - v2, q - v2) @ nrm) / area2
    b1 = float(np.cross(v1 - v3, q - v3) @ nrm) / area2
    b2 = 1.0 - b0 - b1
    if tri[1] not in inside_edge:
        inside_edge[tri[1]] = float(np.float32(_norm(v1p - float(v1p @ e12) * e12)))    # std::map<int, float>
    if tri[2] not in inside_edge:
        inside_edge[tri[2]] = float(np.float32(_norm(v1p - float(v1p @ e13) * e13)))
    if b0 < 0.0 or b1 < 0.0:
        inside_edge[tri[1]] = -1.0
    if b0 < 0.0 or b2 < 0.0:
        inside_edge[tri[2]] = -1.0
    if b0 >= 0.0 and b1 >= 0.0 and b2 >= 0.0:
        return abs(plane), (b0, b1, b2)
    return -1.0, (b0, b1, b2)


def inverse_distance_weights(pos, p, ids):                 # :515-526
    w = [1.0 / max(1e-8, _norm(p - pos[i])) for i in ids]
    s = sum(w)
    return [x / s for x in w]


def build(pos, neigh, ratio=8.0, lower_bound=1000):
    """Returns the list of prolongation matrices U_k (scipy CSC, n_k x n_{k+1})."""
    P = np.asarray(pos, dtype=np.float64)
    NB = np.asarray(neigh, dtype=np.int64)
    Us = []
    level = 0
    while P.shape[0] > lower_bound and level < 10:         # :103
        nf = P.shape[0]
        radius = ratio ** (1.0 / 3.0) * average_edge_length(P, NB)      # :104
        D = np.full(nf, np.finfo(np.float64).max)
        nearest = np.zeros(nf, np.int64)
        sample = fast_disk_sample(P, NB, radius, D, nearest)            # :128
        if len(sample) < lower_bound:                                    # :156-159
            break
        nc = len(sample)
        dijkstra_with_cluster(P, sample, NB, D, nearest)                 # :170
        cadj = [set() for _ in range(nc)]                               # :178-187
        for f in range(nf):
            for j in range(NB.shape[1]):
                g = NB[f, j]
                if g < 0:
                    break
                if nearest[f] != nearest[g]:
                    cadj[nearest[f]].add(int(nearest[g]))
        cadj = [sorted(a) for a in cadj]
        max_nb = max((len(a) for a in cadj), default=0)
        Kc = max(max_nb, 1)
        NBc = -np.ones((nc, Kc), np.int64)                              # :196-205
        if max_nb > 0:
            for i in range(nc):
                NBc[i, 0] = i
                cnt = 1
                for node in cadj[i]:
                    if node == i:
                        continue
                    if cnt >= max_nb:
                        break
                    NBc[i, cnt] = node
                    cnt += 1
        Pc = np.zeros((nc, 3))                                           # :216-240 (not nested)
        csize = np.zeros(nc, np.int64)
        for f in range(nf):
            Pc[nearest[f]] += P[f]
            csize[nearest[f]] += 1
        for c in range(nc):
            if csize[c] == 1:
                s = P[sample[c]].copy()
                for nb in cadj[c]:
                    s += P[sample[nb]]
                Pc[c] = s / (len(cadj[c]) + 1.0)
            else:
                Pc[c] = Pc[c] / csize[c]
        tris, normals, tris_of = [], [], [[] for _ in range(nc)]        # :247-281
        for c in range(nc):
            a = cadj[c]
            for i2 in range(len(a)):
                v2 = a[i2]
                if v2 < c:
                    continue
                for i3 in range(i2 + 1, len(a)):
                    v3 = a[i3]
                    if v3 < c:
                        continue
                    if v3 in set(cadj[v2]):
                        t = len(tris)
                        tris.append((c, v2, v3))
                        normals.append(_normalized(np.cross(Pc[v2] - Pc[c], Pc[v3] - Pc[c])))
                        tris_of[c].append(t); tris_of[v2].append(t); tris_of[v3].append(t)
        rows, cols, vals = [], [], []                                    # :291-452
        for f in range(nf):
            p = P[f]
            c = int(nearest[f])
            pc = Pc[c]

            def edge_row(other):
                e = Pc[other] - pc
                ln = max(_norm(e), 1e-8)
                w2 = float((p - pc) @ _normalized(e)) / ln
                w2 = min(max(w2, 0.0), 1.0)
                rows.extend([f, f]); cols.extend([c, other]); vals.extend([1.0 - w2, w2])

            if not cadj[c]:
                rows.append(f); cols.append(c); vals.append(1.0)
                continue
            if len(cadj[c]) == 1:
                edge_row(cadj[c][0])
                continue
            inside_edge = {}
            found = None
            for t in tris_of[c]:
                tri = list(tris[t])
                while tri[0] != c:
                    tri = tri[1:] + tri[:1]
                dist, bary = in_triangle(p, tri, normals[t], Pc, inside_edge)
                if dist >= 0.0:
                    found = (tri, bary)
                    break
            if found:
                tri, bary = found
                for j in range(3):
                    rows.append(f); cols.append(tri[j]); vals.append(bary[j])
                continue
            edge_to = None
            for key in sorted(inside_edge):                               # std::map iterates in key order; first hit wins
                if inside_edge[key] >= 0.0:
                    edge_to = key
                    break
            if edge_to is not None:
                edge_row(edge_to)
                continue
            cand = sorted((_norm(p - Pc[nb]), int(nb)) for nb in NBc[c] if nb >= 0 and nb != c)
            ids = [c] + [nb for _, nb in cand[:2]]
            w = inverse_distance_weights(Pc, p, ids)
            for i, wi in zip(ids, w):
                rows.append(f); cols.append(i); vals.append(wi)
        U = sp.coo_matrix((vals, (rows, cols)), shape=(nf, nc)).tocsc()   # setFromTriplets: duplicates summed
        U.sort_indices()
        Us.append(U)
        P, NB = Pc, NBc
        level += 1
    return Us
