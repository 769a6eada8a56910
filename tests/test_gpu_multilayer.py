"""GPU: FD_KERNEL_GAUSSIAN_ML -- the SOP's model = 1 (alglib::rbfsetalgomultilayer(model, radius,
layers, lambda), reference src/SOP_FaceDeform.cpp:346-348) in the dense form of
include/facedeform_hip.h: polynomial by least squares first, then one positive definite Gaussian
system per layer on the residual.  Checked against the SciPy-written golden vectors
(tests/golden/ml_golden.npz) and the numpy oracle (oracle/fd_oracle.py build_multilayer)."""
import os
import numpy as np
import pytest
import torch

from conftest import parity_ratio
from facedeform_amd import capi, synth
from facedeform_amd.sop import FaceDeformSOP
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ml_golden.npz")
CASES = ["m40_lin", "m96_const", "m64_zero", "m150_lin_1layer"]


def _engine(rest, deform, R, L, lam, term, precision=capi.EVAL_FP32):
    e = capi.Engine(precision=precision)
    e.set_kernel(capi.KERNEL_GAUSSIAN_ML, [R, L, lam]); e.set_term(term)
    e.set_points(rest, (deform - rest).astype(np.float32))
    return e


@pytest.mark.parametrize("name", CASES)
def test_golden_vectors(hip_lib, name):
    g = np.load(GOLD)
    R, L, lam, term = g[name + "_params"]
    L, term = int(L), int(term)
    rest, deform = g[name + "_rest"], g[name + "_deform"]
    M = rest.shape[0]
    for precision, tol in ((capi.EVAL_FP32, 1e-5), (capi.EVAL_FP64, 2e-7)):
        e = _engine(rest, deform, R, L, lam, term, precision)
        rep = e.build()
        T = (4, 1, 0)[term]
        assert rep.terminationtype == 1 and rep.n == M + T and rep.iterationscount == M + T
        assert hip_lib.fd_model_centres(e.ctx) == M * L
        W, radii = e.get_weights()
        Wg = g[name + "_W"]
        assert W.shape == Wg.shape and np.abs(W - Wg).max() <= 1e-8 * np.abs(Wg).max()
        assert np.array_equal(radii, np.repeat(R / 2.0 ** np.arange(L), M))
        x = g[name + "_x"]
        out, _ = e.deform(x)
        ref = x + g[name + "_disp"].astype(np.float32)       # fp64 result narrowed, then added in fp32 (reference :415, :438)
        assert parity_ratio(out, ref, x, tol) <= 1.0, precision
        e.close()


@pytest.mark.parametrize("M,L,term", [(256, 4, capi.TERM_LINEAR), (700, 3, capi.TERM_CONST), (33, 8, capi.TERM_ZERO)])
def test_against_the_oracle_at_rig_sizes(hip_lib, oracle, M, L, term):
    """The SOP's defaults (radius 1, lambda 0.1) at the benchmark rig size, a ranged
    back-substitution with a ragged last block, and the layer limit."""
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 2)
    table = oracle.control_table(rest, deform)
    tt, table_ml, Wo, radii_o = oracle.build_multilayer(table, 1.0, L, 0.1, term)
    assert tt == 1
    e = _engine(rest, deform, 1.0, L, 0.1, term)
    assert e.build().terminationtype == 1
    W, radii = e.get_weights()
    assert np.abs(W - Wo).max() <= 1e-8 * np.abs(Wo).max() and np.array_equal(radii, radii_o)
    P = synth.head_mesh(20000)
    dist2 = np.linspace(0.0, 1.2, P.shape[0]).astype(np.float32)          # gate + fall-off in play
    out, fall = e.deform(P, dist2=dist2, radius2=1.0, falloffrate=1.5)
    ref, rfall = oracle.deform(table_ml, fo.KERNEL_GAUSSIAN_QNN, radii_o, Wo, P, dist2=dist2, radius2=1.0, falloffrate=1.5)
    assert parity_ratio(out, ref, P, 1e-5) <= 1.0
    assert np.abs(fall - rfall).max() <= 3e-7                 # powf on the device vs libm: last-place differences
    assert np.array_equal(fall == 0.0, rfall == 0.0)          # the gate itself is exact
    e.close()


def test_boundary_behaviour(hip_lib):
    rest = synth.control_points(48, "head")
    deform = synth.deformed_rig(rest, 0)
    delta = (deform - rest).astype(np.float32)
    P = synth.head_mesh(3000)
    e = _engine(rest, deform, 0.7, 3, 0.1, capi.TERM_LINEAR)
    e.build()
    out, _ = e.deform(P)
    # the blob is the solved model -- M * layers Gaussians with their own radii -- and replicates bitwise
    blob = e.export_model()
    r = capi.Engine()
    r.import_model(blob)
    assert hip_lib.fd_model_centres(r.ctx) == 48 * 3
    out2, _ = r.deform(P)
    assert np.array_equal(out, out2)
    r.close()
    # no single factorisation to reuse
    with pytest.raises(capi.FdError) as ei:
        e.set_deltas(delta)
    assert ei.value.code == capi.FD_E_NOT_BUILT
    # parameters are checked; layers are clamped by the caller, not silently
    for bad in ([0.0, 2, 0.1], [1.0, 0, 0.1], [1.0, 9, 0.1], [1.0, 2, -0.1]):
        with pytest.raises(capi.FdError):
            e.set_kernel(capi.KERNEL_GAUSSIAN_ML, bad)
    # coincident centres: -5, and the context recovers
    dup = rest.copy(); dup[5] = dup[30]
    e.set_points(dup, delta)
    assert e.build(check=False).terminationtype == -5
    e.set_points(rest, delta)
    assert e.build().terminationtype == 1
    out3, _ = e.deform(P)
    assert np.array_equal(out, out3)
    # more layers than before on the same context: the record arrays grow
    e.set_kernel(capi.KERNEL_GAUSSIAN_ML, [0.7, 6, 0.1])
    assert e.build().terminationtype == 1 and hip_lib.fd_model_centres(e.ctx) == 48 * 6
    e.close()


def test_batched_builds_match_single_builds_bitwise(hip_lib):
    M, L = 120, 4
    rest = synth.control_points(M, "head")
    dev = torch.device("cuda:0")
    d_rest = torch.from_numpy(rest).to(dev)
    es, d_del = [], []
    for f in range(4):
        e = capi.Engine()
        e.set_kernel(capi.KERNEL_GAUSSIAN_ML, [0.5, L, 0.05]); e.set_term(capi.TERM_LINEAR)
        es.append(e)
        d_del.append(torch.from_numpy(synth.smooth_deltas(rest, f).astype(np.float32)).to(dev))
    b = capi.Batch(es)
    b.set_points_dev([d_rest.data_ptr()] * 4, [t.data_ptr() for t in d_del], M)
    b.build_async()
    assert all(r.terminationtype == 1 for r in b.build_result())
    for f, e in enumerate(es):
        s = capi.Engine()
        s.set_kernel(capi.KERNEL_GAUSSIAN_ML, [0.5, L, 0.05]); s.set_term(capi.TERM_LINEAR)
        s.set_points(rest, synth.smooth_deltas(rest, f).astype(np.float32)); s.build()
        assert np.array_equal(s.get_weights()[0], e.get_weights()[0]), f
        s.close()
    # one batched evaluation call gives every context what its own fd_deform_dev gives, bit for bit
    # (the multilayer contexts are launched one by one there: their kernel shares distances per centre)
    N = 4096
    d_P = torch.from_numpy(synth.head_mesh(N)).to(dev)
    outs = [torch.empty_like(d_P) for _ in es]
    b.deform_dev(N, [d_P.data_ptr()] * 4, [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    for e, o in zip(es, outs):
        single = torch.empty_like(d_P)
        e.deform_dev(N, d_P.data_ptr(), single.data_ptr()); e.synchronize()
        assert torch.equal(single, o)
    b.close()
    for e in es:
        e.close()


def test_sop_model_1_is_the_multilayer_model(hip_lib, oracle):
    """fdsop_cook with model = 1 passes radius, layers and lambda as the reference passes them to
    rbfsetalgomultilayer (src/SOP_FaceDeform.cpp:347), clamps included (:251-257)."""
    rest = synth.control_points(64, "head")
    deform = synth.deformed_rig(rest, 1)
    P = synth.head_mesh(4000)
    node = FaceDeformSOP()
    node.set("model", "1"); node.set("term", "0")
    node.set("radius", 0.8); node.set("layers", 3); node.set("lambda", 0.001)      # lambda clamps to 0.01
    res = node.cook(P, rest, deform, dist2=np.zeros(P.shape[0], np.float32))
    assert res.severity == capi.FDSOP_MESSAGE, res.messages
    assert res.infos == ["Termination type: 1, Iterations: 68"]
    table = oracle.control_table(rest, deform)
    tt, table_ml, W, radii = oracle.build_multilayer(table, float(np.float32(0.8)), 3, float(np.float32(0.01)), fo.TERM_LINEAR)
    ref, _ = oracle.deform(table_ml, fo.KERNEL_GAUSSIAN_QNN, radii, W, P, radius2=float(np.float32(0.8)) ** 2)
    assert parity_ratio(res.P, ref, P, 1e-5) <= 1.0
    node.close()
