"""How often does partial pivoting leave the diagonal on the QNN model's systems?  (VERDICT r1 #6a.)

(Phi + lambda I) w = r with Phi_ij = exp(-|c_i - c_j|^2 / R_j^2), R_j = min(q nn_j, z median(nn)) -- the SOP's
default model (q = 1, z = 5).  For rigs like the benchmark's (and harder ones: clustered points, near-duplicates)
run LU with partial pivoting in fp64 and count, per column, whether the pivot row is the diagonal row, whether it
stays inside the column's 32-row diagonal block, and how large the unpivoted multipliers would have been.
CPU only (numpy + scipy):  python tests/tools/qnn_pivot_stats.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from facedeform_amd import synth


def qnn_matrix(c, q=1.0, z=5.0, lam=0.0):
    d2 = ((c[:, None, :] - c[None, :, :]) ** 2).sum(-1)
    nn = np.sqrt(np.where(np.eye(len(c), dtype=bool), np.inf, d2).min(axis=1))
    med = np.sort(nn)[len(c) // 2]
    R = np.minimum(q * nn, z * med)
    return np.exp(-d2 / (R[None, :] ** 2)) + lam * np.eye(len(c)), R


def lu_stats(A, nb=32):
    A = A.copy()
    n = len(A)
    off_diag = out_of_block = 0
    worst_mult = 0.0        # largest |multiplier| an unpivoted elimination would use
    growth = np.abs(A).max()
    a0 = growth
    for k in range(n):
        col = np.abs(A[k:, k])
        p = int(col.argmax()) + k
        worst_mult = max(worst_mult, col.max() / max(abs(A[k, k]), 1e-300))
        if p != k:
            off_diag += 1
            if p >= (k // nb + 1) * nb:
                out_of_block += 1
            A[[k, p]] = A[[p, k]]
        A[k + 1:, k] /= A[k, k]
        A[k + 1:, k + 1:] -= np.outer(A[k + 1:, k], A[k, k + 1:])
        growth = max(growth, np.abs(A[k + 1:, k + 1:]).max() if k + 1 < n else 0.0)
    return off_diag, out_of_block, worst_mult, growth / a0


def main():
    rng = np.random.default_rng(0)
    cases = []
    for M in (64, 256, 512, 1024):
        cases.append((f"head rig M={M}", synth.control_points(M, "head").astype(np.float64)))
    cases.append(("sphere rig M=256", synth.control_points(256, "sphere").astype(np.float64)))
    c = synth.control_points(256, "head").astype(np.float64)
    cl = c.copy(); cl[:64] = c[0] + 0.02 * rng.normal(size=(64, 3))                 # a tight cluster of 64 points
    cases.append(("head rig M=256, 64 points clustered (2% of the head)", cl))
    nd = c.copy(); nd[1::2] = nd[0::2] + 1e-4 * rng.normal(size=(128, 3))            # pairs 1e-4 apart
    cases.append(("head rig M=256, every other point 1e-4 from its neighbour", nd))
    cases.append(("uniform random cube M=512", rng.random((512, 3))))
    print(f"{'rig':62s} {'cond':>9s} {'pivot != diagonal':>18s} {'pivot outside 32-block':>23s} {'max |mult| unpivoted':>21s} {'growth':>7s}")
    for name, pts in cases:
        for q, z in ((1.0, 5.0), (2.0, 5.0)):
            A, R = qnn_matrix(pts, q, z)
            od, ob, wm, gr = lu_stats(A)
            print(f"{name + f' q={q:g}':62s} {np.linalg.cond(A):9.2e} {od:8d} / {len(A):<7d} {ob:10d} / {len(A):<10d} {wm:21.3g} {gr:7.2f}")


if __name__ == "__main__":
    main()
