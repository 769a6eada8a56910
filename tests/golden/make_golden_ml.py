"""Golden vectors for the multilayer Gaussian model (FD_KERNEL_GAUSSIAN_ML), written with an
independent implementation: the polynomial by numpy's least squares, every layer by
scipy.interpolate.RBFInterpolator(kernel='gaussian', epsilon=1/R_l, smoothing=lambda, degree=-1),
which solves (Phi_l + lambda I) w = r_l; evaluation by the interpolators themselves.
SciPy 1.15.3 / numpy 2.2.  Run from the repository root: python tests/golden/make_golden_ml.py"""
import os, sys
import numpy as np
from scipy.interpolate import RBFInterpolator

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facedeform_amd import synth

out = {}
cases = [("m40_lin", 40, 0.6, 4, 0.1, 0), ("m96_const", 96, 0.45, 3, 0.05, 1), ("m64_zero", 64, 0.5, 2, 0.01, 2),
         ("m150_lin_1layer", 150, 0.3, 1, 0.1, 0)]
for name, M, R, L, lam, term in cases:
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 1)
    c = rest.astype(np.float64)
    f = (deform - rest).astype(np.float32).astype(np.float64)       # fp32 subtraction as the SOP does (:278)
    T = (4, 1, 0)[term]
    aff = np.zeros((4, 3))
    if T:
        Pm = np.hstack([np.ones((M, 1)), c])[:, :T]
        aff[:T] = np.linalg.lstsq(Pm, f, rcond=None)[0]
        f = f - Pm @ aff[:T]
    x = synth.head_mesh(400).astype(np.float64)
    disp = (np.hstack([np.ones((len(x), 1)), x]) @ aff) if T else np.zeros((len(x), 3))
    W = np.zeros((M * L + 4, 3))
    for l in range(L):
        Rl = R / 2.0 ** l
        itp = RBFInterpolator(c, f, kernel="gaussian", epsilon=1.0 / Rl, smoothing=lam, degree=-1)
        W[l * M:(l + 1) * M] = itp._coeffs[:M]
        disp = disp + itp(x)
        f = f - itp(c)                                   # itp(c) = Phi_l w_l: SciPy adds no smoothing when evaluating
    W[M * L:] = aff
    out[name + "_rest"] = rest; out[name + "_deform"] = deform
    out[name + "_params"] = np.array([R, L, lam, term], np.float64)
    out[name + "_W"] = W; out[name + "_x"] = x.astype(np.float32); out[name + "_disp"] = disp
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ml_golden.npz"), **out)
print("wrote ml_golden.npz:", sorted(k for k in out if k.endswith("_W")))
