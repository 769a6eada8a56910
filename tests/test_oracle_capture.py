"""CPU: the oracle's dist2 producer (reference src/capture.cpp:46-99) against an independently
formulated numpy computation (tests/golden/make_golden_capture.py)."""
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cap():
    return np.load(os.path.join(HERE, "golden", "capture_golden.npz"))


def test_distances_match_independent_formulation(oracle, cap):
    out = oracle.capture_dist2(cap["P"], cap["tris"], radius2=1e30, dofalloff=True)
    ref = cap["d2"]
    assert np.abs(out - ref).max() <= 1e-6 * max(1.0, ref.max()) and (out >= 0).all()
    assert np.all(out[:150] <= 1e-12)            # points placed on vertices, faces and edges


def test_capture_semantics(oracle, cap):
    P, tris, ref, mask = cap["P"], cap["tris"], cap["d2"], cap["mask"]
    r2 = np.float32(0.09)
    out = oracle.capture_dist2(P, tris, r2, True, mask)
    inside = mask.astype(bool)
    assert np.all(out[~inside] == 0.0)                                   # attribute default (capture.cpp:31)
    near = inside & (ref.astype(np.float32) < r2)
    far = inside & ~(ref.astype(np.float32) < r2)
    assert near.any() and far.any()
    assert np.all(out[far] == -1.0)                                      # nothing within the radius (:76,88)
    assert np.allclose(out[near], ref[near], rtol=1e-6, atol=1e-9)
    assert np.all(oracle.capture_dist2(P, tris, r2, False, mask) == 0.0)   # dofalloff off (:71-75)
    assert np.all(oracle.capture_dist2(P[:10], tris[:0], r2, True) == -1.0)   # no rig surface at all


def _grid_mesh(nx, ny):
    """nx x ny grid in the z = 0 plane with 4-neighbour edges, as a CSR adjacency."""
    idx = np.arange(nx * ny).reshape(ny, nx)
    P = np.stack([np.tile(np.arange(nx), ny), np.repeat(np.arange(ny), nx), np.zeros(nx * ny)], axis=1).astype(np.float32)
    nbrs = [[] for _ in range(nx * ny)]
    for y in range(ny):
        for x in range(nx):
            for dx, dy in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                if 0 <= x + dx < nx and 0 <= y + dy < ny:
                    nbrs[idx[y, x]].append(idx[y + dy, x + dx])
    offsets = np.zeros(nx * ny + 1, np.int64)
    offsets[1:] = np.cumsum([len(n) for n in nbrs])
    return P, offsets, np.concatenate([np.array(n, np.int32) for n in nbrs])


def test_islands_on_a_grid_are_manhattan_balls(oracle):
    """On a 4-connected grid the points within k edges of a seed are its Manhattan ball."""
    P, offsets, nb = _grid_mesh(31, 23)
    rig = np.array([[5.2, 4.9, 0.3], [25.0, 17.6, -0.2]], np.float32)     # nearest grid points (5,5) and (25,18)
    for k in (0, 1, 4):
        mask = oracle.capture_islands(P, offsets, nb, rig, k)
        man = np.minimum(np.abs(P[:, 0] - 5) + np.abs(P[:, 1] - 5), np.abs(P[:, 0] - 25) + np.abs(P[:, 1] - 18))
        assert np.array_equal(mask.astype(bool), man <= k), k
    assert oracle.capture_islands(P, offsets, nb, rig[:0], 3).sum() == 0
