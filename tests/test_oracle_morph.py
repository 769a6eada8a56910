"""CPU: the oracle's morph-space restatement (reference src/dbse.cpp, SOP_FaceDeform.cpp:444-473)
against vectors written by LAPACK (SciPy raw QR) and numpy -- tests/golden/make_golden_morph.py."""
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def morph_golden():
    return np.load(os.path.join(HERE, "golden", "morph_golden.npz"))


def _case(g, name):
    clamp = g[name + "/clamp"]
    return dict(rest=g[name + "/rest"], shapes=list(g[name + "/shapes"]), qr=g[name + "/qr"], tau=g[name + "/tau"],
                P=g[name + "/P"], w=g[name + "/w"], P_out=g[name + "/P_out"],
                clamp=None if np.isnan(clamp[0]) else clamp, add_delta=bool(g[name + "/add_delta"]),
                falloffradius=float(g[name + "/falloffradius"]))


def test_packed_qr_matches_lapack(oracle, morph_golden):
    for name in [str(n) for n in morph_golden["names"]]:
        c = _case(morph_golden, name)
        A = oracle.morph_shapes_matrix(c["rest"], c["shapes"])
        # the matrix itself: fp32 deltas, exactly
        ref_A = np.stack([(s - c["rest"]).astype(np.float32).reshape(-1) for s in c["shapes"]], axis=1)
        assert np.array_equal(A, ref_A.astype(np.float64)), name
        QR, tau = oracle.morph_qr(A)
        scale = np.abs(c["qr"]).max()
        assert np.abs(QR - c["qr"]).max() <= 1e-12 * scale, (name, np.abs(QR - c["qr"]).max() / scale)
        assert np.abs(tau - c["tau"]).max() <= 1e-13, name
        # and it is a QR: R^T R = A^T A
        R = np.triu(QR[: QR.shape[1]])
        assert np.allclose(R.T @ R, A.T @ A, rtol=1e-10, atol=1e-12 * scale * scale), name


def test_weights_and_displacement_match_numpy(oracle, morph_golden):
    for name in [str(n) for n in morph_golden["names"]]:
        c = _case(morph_golden, name)
        A = oracle.morph_shapes_matrix(c["rest"], c["shapes"])
        w = oracle.morph_weights(np.asfortranarray(c["qr"]), c["P"], c["rest"])
        assert np.abs(w - c["w"]).max() <= 1e-12 * max(1.0, np.abs(c["w"]).max()), name
        out = oracle.morph_displace(A, c["w"], c["P"], c["rest"], c["clamp"], c["add_delta"], c["falloffradius"])
        # same fp32 operations in the same order: bit-identical
        assert np.array_equal(out, c["P_out"]), (name, np.abs(out - c["P_out"]).max())


def test_degenerate_columns(oracle):
    """A blendshape equal to the rest pose gives a zero column: tau = 0, R entry 0, nothing NaN
    (Eigen's makeHouseholder small-tail branch, which LAPACK shares)."""
    rng = np.random.default_rng(3)
    rest = rng.normal(size=(30, 3)).astype(np.float32)
    shapes = [rest.copy(), (rest + rng.normal(size=rest.shape).astype(np.float32) * 0.1).astype(np.float32), rest.copy()]
    A = oracle.morph_shapes_matrix(rest, shapes)
    QR, tau = oracle.morph_qr(A)
    assert np.isfinite(QR).all() and np.isfinite(tau).all()
    assert tau[0] == 0.0 and np.all(QR[:, 0] == 0.0)
    w = oracle.morph_weights(QR, shapes[1], rest)
    assert np.isfinite(w).all()
