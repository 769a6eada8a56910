"""Golden vectors for the dist2 producer (reference src/capture.cpp:46-99), written by an
independent numpy formulation: the squared distance from a point to a triangle as the minimum of
(a) the distance to the plane when the projection falls inside the triangle and (b) the distances
to the three edge segments -- not the Voronoi-region walk the oracle and the kernel use.

    python tests/golden/make_golden_capture.py      -> tests/golden/capture_golden.npz
"""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def seg_d2(p, a, b):
    ab = b - a
    den = np.einsum("ij,ij->i", ab, ab)
    t = np.where(den > 0, np.einsum("ij,ij->i", p - a, ab) / np.where(den > 0, den, 1), 0.0)
    t = np.clip(t, 0.0, 1.0)
    q = a + t[:, None] * ab
    return np.einsum("ij,ij->i", p - q, p - q)


def tri_d2(p, tri):
    """p (n,3) float64, tri (9,) -> (n,) squared distances."""
    a, b, c = tri[0:3][None], tri[3:6][None], tri[6:9][None]
    n = np.cross(b - a, c - a)
    nn = np.einsum("ij,ij->i", n, n)
    best = np.minimum(np.minimum(seg_d2(p, a, b), seg_d2(p, b, c)), seg_d2(p, c, a))
    if nn[0] > 0:
        dist = np.einsum("ij,ij->i", p - a, n) / nn                 # signed distance / |n|
        q = p - dist[:, None] * n                                    # projection onto the plane
        # inside test by barycentric coordinates
        v0, v1, v2 = b - a, c - a, q - a
        d00, d01, d11 = (v0 * v0).sum(), (v0 * v1).sum(), (v1 * v1).sum()
        d20, d21 = np.einsum("ij,ij->i", v2, v0), np.einsum("ij,ij->i", v2, v1)
        den = d00 * d11 - d01 * d01
        v = (d11 * d20 - d01 * d21) / den
        w = (d00 * d21 - d01 * d20) / den
        inside = (v >= 0) & (w >= 0) & (v + w <= 1)
        plane = dist * dist * nn
        best = np.where(inside, np.minimum(best, plane), best)
    return best


def main():
    rng = np.random.default_rng(20261004)
    # rig: a small triangulated patch of a sphere plus two degenerate triangles
    M = 60
    rig = rng.normal(size=(M, 3)); rig /= np.linalg.norm(rig, axis=1, keepdims=True)
    rig = rig.astype(np.float32)
    tris = []
    for _ in range(90):
        i = rng.integers(M)
        d = np.linalg.norm(rig - rig[i], axis=1)
        j, k = np.argsort(d)[1:3]
        tris.append(np.concatenate([rig[i], rig[j], rig[k]]))
    tris.append(np.concatenate([rig[0], rig[0], rig[0]]))             # a point
    tris.append(np.concatenate([rig[1], rig[2], rig[1]]))             # a segment
    tris = np.array(tris, np.float32)
    N = 4000
    P = (rng.normal(size=(N, 3)) * 0.8).astype(np.float32)
    P[:50] = rig[:50]                                                 # on the surface
    P[50:100] = (tris[:50, 0:3] + tris[:50, 3:6] + tris[:50, 6:9]) / np.float32(3)   # face interiors
    P[100:150] = (tris[:50, 0:3] + tris[:50, 3:6]) / np.float32(2)    # on edges
    d2 = np.full(N, np.inf)
    p64 = P.astype(np.float64)
    for t in tris.astype(np.float64):
        d2 = np.minimum(d2, tri_d2(p64, t))
    mask = (rng.random(N) < 0.7).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "capture_golden.npz"), P=P, tris=tris, d2=d2, mask=mask)
    print("wrote capture_golden.npz", N, len(tris), float(d2.min()), float(d2.max()))


if __name__ == "__main__":
    main()
