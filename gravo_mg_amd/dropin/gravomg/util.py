"""Input helpers with the names of the reference's gravomg.util (gravomg_bindings/src/gravomg/util.py).  Written
for this build (vectorised numpy); outputs are checked against captured outputs of the reference helpers in
tests/golden/util_neigh.npz.  `neighbors_from_stiffness` additionally accepts any scipy format (the upstream
helper is only correct for CSC input)."""
import numpy as np
import scipy.sparse as sp
from scipy.spatial import cKDTree


def _pad_rows(owner, member, n_rows=None):
    """owner/member: edge list sorted by (owner, member), unique -> (n, kmax) table padded with -1."""
    n = int(owner.max()) + 1 if n_rows is None else n_rows
    deg = np.bincount(owner, minlength=n)
    start = np.concatenate([[0], np.cumsum(deg)[:-1]])
    table = -np.ones((n, int(deg.max())), dtype=np.int32)
    table[owner, np.arange(owner.shape[0]) - start[owner]] = member
    return table


def _unique_edges(i, j):
    key = np.unique(np.stack([i, j], axis=1), axis=0)
    return key[:, 0], key[:, 1]


def neighbors_from_stiffness(S):
    """Row i = ascending ids of the stored entries of row i of S (the vertex itself included), padded with -1."""
    S = sp.csr_matrix(S)
    S.sort_indices()
    owner = np.repeat(np.arange(S.shape[0]), np.diff(S.indptr))
    return _pad_rows(owner, S.indices, S.shape[0])


def neighbors_from_faces(F):
    """One-ring of every vertex from a triangle list (self excluded), ascending, padded with -1."""
    F = np.asarray(F)
    i = np.concatenate([F[:, 0], F[:, 0], F[:, 1], F[:, 1], F[:, 2], F[:, 2]])
    j = np.concatenate([F[:, 1], F[:, 2], F[:, 0], F[:, 2], F[:, 0], F[:, 1]])
    return _pad_rows(*_unique_edges(i, j))


def knn(V, k):
    return cKDTree(V).query(V, k + 1)[1][:, 1:]


def knn_undirected(V, k):
    """Symmetrised k-nearest-neighbour table."""
    n = V.shape[0]
    i = np.repeat(np.arange(n), k)
    j = knn(V, k).ravel()
    return _pad_rows(*_unique_edges(np.concatenate([i, j]), np.concatenate([j, i])), n_rows=n)


def face_area(pos, F):
    return np.linalg.norm(np.cross(pos[F[:, 1]] - pos[F[:, 0]], pos[F[:, 2]] - pos[F[:, 0]]), axis=1) / 2


def normalize_area(pos, F):
    pos = pos / np.sqrt(face_area(pos, F).sum())
    return pos - pos.mean(axis=0, keepdims=True)


def normalize_bounding_box(pos):
    pos = pos - pos.mean(axis=0, keepdims=True)
    return pos * (0.5 / np.abs(pos).max())


def normalize_axes(pos):
    return pos[:, np.argsort(np.std(pos, axis=0))]
