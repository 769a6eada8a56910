"""CPU: the oracle against SciPy's RBFInterpolator (LAPACK) run live on randomised
configurations -- sizes, point clouds, kernels, terms, smoothing -- on top of the 28 committed
golden cases.  Compares the interpolant (what the path delivers) and, where the system is well
conditioned, the weights."""
import warnings

import numpy as np
import pytest

scipy_interp = pytest.importorskip("scipy.interpolate")
from oracle import fd_oracle as fo  # noqa: E402

SCIPY_KERNEL = {fo.KERNEL_GAUSSIAN: "gaussian", fo.KERNEL_THIN_PLATE: "thin_plate_spline",
                fo.KERNEL_BIHARMONIC: "linear", fo.KERNEL_CUBIC: "cubic"}
DEGREE = {fo.TERM_LINEAR: 1, fo.TERM_CONST: 0, fo.TERM_ZERO: -1}


def _cases():
    rng = np.random.default_rng(777)
    out = []
    for i in range(24):
        M = int(rng.integers(5, 140))
        kind = [fo.KERNEL_GAUSSIAN, fo.KERNEL_THIN_PLATE, fo.KERNEL_BIHARMONIC, fo.KERNEL_CUBIC][i % 4]
        term = [fo.TERM_LINEAR, fo.TERM_CONST, fo.TERM_ZERO][int(rng.integers(3))]
        # thin-plate / cubic are conditionally positive definite of order 2: SciPy insists on degree >= 1
        if kind in (fo.KERNEL_THIN_PLATE, fo.KERNEL_CUBIC):
            term = fo.TERM_LINEAR
        lam = float(rng.choice([0.0, 0.0, 1e-3, 0.1]))
        scale = float(rng.choice([0.3, 1.0, 7.0]))
        out.append((i, M, kind, term, lam, scale))
    return out


@pytest.mark.parametrize("i,M,kind,term,lam,scale", _cases())
def test_oracle_matches_scipy(oracle, i, M, kind, term, lam, scale):
    rng = np.random.default_rng(1000 + i)
    rest = (rng.normal(size=(M, 3)) * scale).astype(np.float32)
    deform = (rest + rng.normal(size=(M, 3)).astype(np.float32) * np.float32(0.05 * scale)).astype(np.float32)
    x = (rng.normal(size=(200, 3)) * scale).astype(np.float32)
    # Gaussian radius comparable to the spacing keeps the system solvable in fp64
    radius = float(scale * 1.5 / M ** (1 / 3))
    params = ([radius, lam] if kind == fo.KERNEL_GAUSSIAN else [lam])
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, kind, params, term)
    assert rc == 0 and tt == 1, (rc, tt)
    out, _ = oracle.deform(table, kind, radii, W, x)
    d = (deform - rest).astype(np.float32).astype(np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        interp = scipy_interp.RBFInterpolator(rest.astype(np.float64), d, smoothing=lam, kernel=SCIPY_KERNEL[kind],
                                              epsilon=(1.0 / radius if kind == fo.KERNEL_GAUSSIAN else 1.0),
                                              degree=DEGREE[term])
        ref = interp(x.astype(np.float64))
    got = out.astype(np.float64) - x
    # fdo_deform narrows the displacement to fp32 and adds it to the fp32 position (:415, :438)
    tol = 2e-6 * max(np.abs(ref).max(), 1e-30) + 4 * np.spacing(np.abs(x).max())
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)
    # weights: compare in the interpolant's own metric via the residual at the centres
    fit, _ = oracle.deform(table, kind, radii, W, rest)
    resid = (fit.astype(np.float64) - rest) - d
    if lam == 0.0:
        assert np.abs(resid).max() <= 2e-6 * np.abs(d).max() + 4 * np.spacing(np.abs(rest).max())
