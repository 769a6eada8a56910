"""Golden vectors for the morph-space row (reference src/dbse.cpp): written by LAPACK/numpy, not by
the oracle.  scipy.linalg.qr(mode='raw') returns LAPACK dgeqrf's packed QR and tau -- the storage
and reflector convention of Eigen::HouseholderQR::matrixQR() that dbse.cpp:55 sums over.

    python tests/golden/make_golden_morph.py      -> tests/golden/morph_golden.npz
"""
import os
import numpy as np
import scipy, scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))


def case(rng, N, S, clamp, add_delta, falloffradius):
    f32 = np.float32
    rest = rng.normal(size=(N, 3)).astype(f32)
    shapes = [(rest + (0.1 * rng.normal(size=(N, 3)) * (rng.random((N, 1)) < 0.4)).astype(f32)).astype(f32) for _ in range(S)]
    # dbse.cpp:16-31
    A = np.stack([(s - rest).astype(f32).reshape(-1) for s in shapes], axis=1).astype(np.float64)
    (qr, tau), _ = scipy.linalg.qr(A, mode="raw")
    qr = np.asfortranarray(qr)
    # a deformed mesh: rest + a mix of the shapes + something outside their span
    mix = rng.normal(size=S) * 0.5
    P = (rest + (A @ mix).reshape(N, 3).astype(f32) + (0.01 * rng.normal(size=(N, 3))).astype(f32)).astype(f32)
    delta = (P - rest).astype(f32).reshape(-1).astype(np.float64)          # dbse.cpp:49-51
    w = (delta[:, None] * qr).sum(axis=0)                                    # :55-56
    # dbse.cpp:62-77 + SOP_FaceDeform.cpp:458-473, fp32, columns in order
    disp = np.zeros((N, 3), f32)
    for s in range(S):
        ws = f32(w[s] * 3)
        cw = ws if clamp is None else f32(min(max(ws, f32(clamp[0])), f32(clamp[1])))
        disp = (disp + (A[:, s].astype(f32).reshape(N, 3) * cw).astype(f32)).astype(f32)
    if add_delta:
        disp = (disp + ((P - rest).astype(f32) * f32(falloffradius)).astype(f32)).astype(f32)
    P_out = (rest + disp).astype(f32)
    return dict(rest=rest, shapes=np.stack(shapes), qr=qr, tau=tau, P=P, w=w, P_out=P_out,
                clamp=np.array([np.nan, np.nan] if clamp is None else clamp, f32),
                add_delta=np.int32(add_delta), falloffradius=f32(falloffradius))


def main():
    rng = np.random.default_rng(20261003)
    cases = {
        "small": case(rng, 40, 7, None, False, 0.0),
        "clamped": case(rng, 257, 12, (-0.5, 0.75), False, 0.0),
        "delta_term": case(rng, 1000, 33, (-1.0, 1.0), True, 0.35),
        "single_shape": case(rng, 64, 1, None, True, 1.0),
    }
    out = {"names": np.array(sorted(cases)), "scipy_version": np.array(scipy.__version__)}
    for name, c in cases.items():
        for k, v in c.items():
            out[f"{name}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "morph_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "morph_golden.npz"), {k: v["qr"].shape for k, v in cases.items()})


if __name__ == "__main__":
    main()
