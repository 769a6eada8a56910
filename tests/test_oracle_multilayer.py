"""CPU: the oracle's multilayer Gaussian model (oracle/fd_oracle.py build_multilayer, the dense
statement of the SOP's model = 1, reference src/SOP_FaceDeform.cpp:346-348) against
tests/golden/ml_golden.npz, which SciPy's RBFInterpolator wrote layer by layer
(tests/golden/make_golden_ml.py).  ALGLIB is absent, so this is what pins the restatement."""
import os
import numpy as np
import pytest

from oracle import fd_oracle as fo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ml_golden.npz")
CASES = ["m40_lin", "m96_const", "m64_zero", "m150_lin_1layer"]


@pytest.mark.parametrize("name", CASES)
def test_multilayer_oracle_matches_scipy_layers(oracle, name):
    g = np.load(GOLD)
    R, L, lam, term = g[name + "_params"]
    L, term = int(L), int(term)
    table = oracle.control_table(g[name + "_rest"], g[name + "_deform"])
    tt, table_ml, W, radii = oracle.build_multilayer(table, R, L, lam, term)
    assert tt == 1 and table_ml.shape[0] == table.shape[0] * L
    Wg = g[name + "_W"]
    assert np.abs(W - Wg).max() <= 1e-9 * np.abs(Wg).max()
    M = table.shape[0]
    assert np.array_equal(radii, np.repeat(R / 2.0 ** np.arange(L), M))
    disp = oracle.eval(table_ml, fo.KERNEL_GAUSSIAN_QNN, radii, W, g[name + "_x"].astype(np.float64))
    ref = g[name + "_disp"]
    assert np.abs(disp - ref).max() <= 1e-10 * np.abs(ref).max()


def test_multilayer_oracle_properties(oracle):
    """One layer with the linear term removed first is NOT the saddle-point Gaussian (the polynomial
    is fitted before, not together); with lambda = 0 the first layer interpolates and later layers
    get nothing; coincident centres report -5."""
    g = np.load(GOLD)
    table = oracle.control_table(g["m40_lin_rest"], g["m40_lin_deform"])
    M = table.shape[0]
    tt, tml, W, radii = oracle.build_multilayer(table, 0.6, 3, 0.0, fo.TERM_LINEAR)
    assert tt == 1
    assert np.abs(W[M:3 * M]).max() <= 1e-6 * np.abs(W[:M]).max()
    at_centres = oracle.eval(tml, fo.KERNEL_GAUSSIAN_QNN, radii, W, table[:, :3])
    assert np.abs(at_centres - table[:, 3:]).max() <= 1e-8 * np.abs(table[:, 3:]).max()
    rc, tt1, W1, _ = oracle.build(table, fo.KERNEL_GAUSSIAN, [0.6, 0.1], fo.TERM_LINEAR)
    tt2, _, W2, _ = oracle.build_multilayer(table, 0.6, 1, 0.1, fo.TERM_LINEAR)
    assert tt1 == 1 and tt2 == 1 and np.abs(W1[:M] - W2[:M]).max() > 1e-3 * np.abs(W1[:M]).max()
    dup = table.copy(); dup[7, :3] = dup[3, :3]
    assert oracle.build_multilayer(dup, 0.6, 2, 0.1, fo.TERM_LINEAR)[0] == -5
