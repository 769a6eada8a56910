#!/usr/bin/env python3
"""Generate tests/golden/rbf_golden.npz.

The reference (symek/facedeform) has no tests or fixtures and cannot be built
here (needs Houdini HDK + ALGLIB, neither vendored), so these vectors come from
an INDEPENDENT implementation of the same dense RBF system, run in the build
container: SciPy 1.15.3 scipy.interpolate.RBFInterpolator (LAPACK dgesv), and,
for the per-centre-radius Gaussian that SciPy cannot express, a numpy
linalg.solve of the explicitly assembled system.  Nothing here touches
oracle/ or facedeform_amd/csrc: the fixtures pin both.

Run:  python tests/golden/make_golden.py     (rewrites rbf_golden.npz)
"""
import os
import sys
import warnings

import numpy as np
from scipy.interpolate import RBFInterpolator

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from facedeform_amd import synth  # noqa: E402  (input recipes only; no compute)

SCIPY_KERNEL = {"gaussian": "gaussian", "thin_plate": "thin_plate_spline",
                "biharmonic": "linear", "cubic": "cubic"}
DEGREE = {"linear": 1, "const": 0, "zero": -1}


def scipy_case(rest, deform, x, kernel, term, radius=1.0, lam=0.0):
    y = rest.astype(np.float64)
    d = (deform - rest).astype(np.float32).astype(np.float64)  # fp32 delta, reference :278
    eps = 1.0 / radius if kernel == "gaussian" else 1.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        interp = RBFInterpolator(y, d, smoothing=lam, kernel=SCIPY_KERNEL[kernel],
                                 epsilon=eps, degree=DEGREE[term])
        out = interp(x.astype(np.float64))
    return out, np.array(interp._coeffs[: y.shape[0]])


def qnn_radii(y, q, z):
    d = np.linalg.norm(y[:, None, :] - y[None, :, :], axis=2)
    np.fill_diagonal(d, np.inf)
    r = q * d.min(axis=1)
    med = np.sort(r)[len(r) // 2]        # ALGLIB's tmp[n/2] (recollection)
    return np.minimum(r, z * med)


def numpy_qnn_case(rest, deform, x, term, q, z, lam=0.0):
    y = rest.astype(np.float64)
    f = (deform - rest).astype(np.float32).astype(np.float64)
    M = y.shape[0]
    R = qnn_radii(y, q, z)
    d2 = ((y[:, None, :] - y[None, :, :]) ** 2).sum(axis=2)
    Phi = np.exp(-d2 / (R[None, :] ** 2)) + lam * np.eye(M)
    T = {"linear": 4, "const": 1, "zero": 0}[term]
    # ALGLIB's order for its Gaussian models: the term's polynomial is a
    # least-squares fit to the deltas, removed first; the Gaussians fit what is left.
    P = np.concatenate([np.ones((M, 1)), y], axis=1)[:, :T]
    a = np.linalg.lstsq(P, f, rcond=None)[0] if T else np.zeros((0, 3))
    w = np.linalg.solve(Phi, f - P @ a)
    xx = x.astype(np.float64)
    e2 = ((xx[:, None, :] - y[None, :, :]) ** 2).sum(axis=2)
    out = np.exp(-e2 / (R[None, :] ** 2)) @ w
    if T:
        out = out + np.concatenate([np.ones((xx.shape[0], 1)), xx], axis=1)[:, :T] @ a
    return out, w, R


def main():
    out = {}
    names = []
    # evaluation points: a coarse sphere scaled a little off the surface plus points
    # sitting exactly on control points (d2 == 0 branch of thin-plate)
    rest32 = synth.control_points(32, "sphere")
    deform32 = synth.deformed_rig(rest32, frame=0)
    x_small = np.concatenate([1.07 * synth.sphere_mesh(200), rest32[:5]], axis=0).astype(np.float32)

    for kernel in ("thin_plate", "gaussian", "biharmonic", "cubic"):
        for term in ("linear", "const", "zero"):
            for lam in (0.0, 0.1):
                if kernel == "thin_plate" and term == "zero" and lam == 0.0:
                    continue  # indefinite, zero diagonal: ill-posed, nothing to pin
                name = f"{kernel}_{term}_lam{lam:g}_M32"
                radius = 0.6
                vals, w = scipy_case(rest32, deform32, x_small, kernel, term, radius, lam)
                out[name + "/rest"] = rest32
                out[name + "/deform"] = deform32
                out[name + "/x"] = x_small
                out[name + "/delta"] = vals
                out[name + "/w"] = w
                out[name + "/params"] = np.array([radius, lam] if kernel == "gaussian" else [lam])
                names.append(name)

    # QNN (the SOP's default model: q=1, z=5), all three terms
    for term in ("linear", "const", "zero"):
        name = f"gaussian_qnn_{term}_lam0_M32"
        vals, w, R = numpy_qnn_case(rest32, deform32, x_small, term, 1.0, 5.0)
        out[name + "/rest"] = rest32
        out[name + "/deform"] = deform32
        out[name + "/x"] = x_small
        out[name + "/delta"] = vals
        out[name + "/w"] = w
        out[name + "/radii"] = R
        out[name + "/params"] = np.array([1.0, 5.0])
        names.append(name)

    # C2-shaped conditioning check: 256 head control points, thin-plate + linear
    rest256 = synth.control_points(256, "head")
    deform256 = synth.deformed_rig(rest256, frame=0)
    x_mid = synth.head_mesh(400)
    for kernel, term in (("thin_plate", "linear"), ("gaussian", "linear")):
        name = f"{kernel}_{term}_lam0_M256"
        vals, w = scipy_case(rest256, deform256, x_mid, kernel, term, 0.25, 0.0)
        out[name + "/rest"] = rest256
        out[name + "/deform"] = deform256
        out[name + "/x"] = x_mid
        out[name + "/delta"] = vals
        out[name + "/w"] = w
        out[name + "/params"] = np.array([0.25, 0.0] if kernel == "gaussian" else [0.0])
        names.append(name)

    out["names"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rbf_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(names)} cases, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
