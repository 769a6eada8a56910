"""GPU: morph-space reprojection (fd_morph_*, next row N1) against the LAPACK/numpy golden
vectors and the oracle (reference src/dbse.cpp:9-87, SOP_FaceDeform.cpp:444-473).

Bars: packed QR <= 1e-11 relative to max|QR| (fp64 on both sides, different summation order);
weights <= 1e-10 relative; displacement: the fp32 loop is the reference's own operation order,
so GIVEN THE SAME WEIGHTS the positions are bit-identical; with the device's own weights they
are checked to 1e-5 of the per-vertex displacement (north_star's tolerance)."""
import os
import numpy as np
import pytest
import torch

from facedeform_amd import capi, synth

pytestmark = pytest.mark.gpu
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


def _disp_ratio(out, ref, rest, tol=1e-5):
    d = out.astype(np.float64) - rest
    dr = ref.astype(np.float64) - rest
    err = np.linalg.norm(d - dr, axis=1)
    ulp = np.spacing(np.abs(ref).max(axis=1).astype(np.float32)).astype(np.float64)
    scale = np.maximum(np.linalg.norm(dr, axis=1), 1e-5 * np.linalg.norm(dr, axis=1).max())
    return float((err / (tol * scale + 2 * ulp)).max())


def test_golden_cases(hip_lib, morph_golden):
    for name in [str(n) for n in morph_golden["names"]]:
        c = _case(morph_golden, name)
        m = capi.Morph()
        assert not m.initialised
        m.init(c["rest"], c["shapes"])
        assert m.initialised and not m.computed
        QR, tau = m.qr()
        scale = np.abs(c["qr"]).max()
        assert np.abs(QR - c["qr"]).max() <= 1e-11 * scale, (name, np.abs(QR - c["qr"]).max() / scale)
        assert np.abs(tau - c["tau"]).max() <= 1e-12, name
        out, w = m.apply(c["P"], c["clamp"], c["add_delta"], c["falloffradius"])
        assert m.computed
        assert np.abs(w - c["w"]).max() <= 1e-10 * max(1.0, np.abs(c["w"]).max()), name
        assert _disp_ratio(out, c["P_out"], c["rest"]) <= 1.0, name
        m.close()


def test_against_oracle_ragged_sizes_and_device_pointers(hip_lib, oracle):
    """N not a multiple of anything, more shapes, device-resident inputs, a zero column."""
    rng = np.random.default_rng(11)
    N, S = 12_347, 41
    rest = synth.head_mesh(N)
    shapes = [(rest + (0.05 * rng.normal(size=(N, 3)) * (rng.random((N, 1)) < 0.3)).astype(np.float32)).astype(np.float32)
              for _ in range(S)]
    shapes[5] = rest.copy()                                   # a blendshape equal to the rest pose
    P = (rest + 0.3 * (shapes[2] - rest) - 0.2 * (shapes[17] - rest) + 0.002 * rng.normal(size=(N, 3))).astype(np.float32)
    A = oracle.morph_shapes_matrix(rest, shapes)
    QR_ref, tau_ref = oracle.morph_qr(A)
    w_ref = oracle.morph_weights(QR_ref, P, rest)
    dev = torch.device("cuda", 0)
    d_rest = torch.from_numpy(rest).to(dev)
    d_shapes = [torch.from_numpy(s).to(dev) for s in shapes]
    d_P = torch.from_numpy(P).to(dev)
    torch.cuda.synchronize()
    m = capi.Morph()
    m.init_dev(N, d_rest.data_ptr(), [t.data_ptr() for t in d_shapes])
    QR, tau = m.qr()
    scale = np.abs(QR_ref).max()
    assert np.isfinite(QR).all()
    assert np.abs(QR - QR_ref).max() <= 1e-11 * scale
    assert np.abs(tau - tau_ref).max() <= 1e-12 and tau[5] == 0.0
    stream = torch.cuda.Stream(device=dev)
    for clamp, add_delta, fr in ((None, False, 0.0), ((-0.4, 0.9), True, 0.5)):
        d_work = d_P.clone()
        torch.cuda.synchronize()
        m.compute_weights_dev(d_work.data_ptr(), stream.cuda_stream)
        m.displace_dev(d_work.data_ptr(), clamp, add_delta, fr, stream.cuda_stream)
        stream.synchronize()
        w = m.weights()
        assert np.abs(w - w_ref).max() <= 1e-10 * max(1.0, np.abs(w_ref).max())
        out = d_work.cpu().numpy()
        # same weights in, same fp32 operations in the same order: bit-identical to the oracle
        ref_same_w = oracle.morph_displace(A, w, P, rest, clamp, add_delta, fr)
        assert np.array_equal(out, ref_same_w)
        ref = oracle.morph_displace(A, w_ref, P, rest, clamp, add_delta, fr)
        assert _disp_ratio(out, ref, rest) <= 1.0
    m.close()


@pytest.mark.parametrize("N,S", [(3, 9), (5, 8), (700, 1), (700, 7), (700, 8), (700, 9), (700, 16), (700, 17), (1024 // 3 + 1, 33)])
def test_panel_edges_of_the_blocked_factorisation(hip_lib, oracle, N, S):
    """The factorisation works in panels of 8 columns with a block-reflector update behind each
    (csrc/fd_morph.hip); Eigen's own HouseholderQR is blocked the same way and the packed result is
    the column-by-column one up to rounding.  Shape counts around the panel width, a square matrix
    (as many rows as shapes: the last reflector has an empty tail), a single workgroup's worth of rows."""
    rng = np.random.default_rng(100 * N + S)
    rest = rng.normal(size=(N, 3)).astype(np.float32)
    shapes = [(rest + 0.1 * rng.normal(size=(N, 3))).astype(np.float32) for _ in range(S)]
    A = oracle.morph_shapes_matrix(rest, shapes)
    QR_ref, tau_ref = oracle.morph_qr(A)
    m = capi.Morph()
    m.init(rest, shapes)
    QR, tau = m.qr()
    assert np.abs(QR - QR_ref).max() <= 1e-11 * np.abs(QR_ref).max(), np.abs(QR - QR_ref).max() / np.abs(QR_ref).max()
    assert np.abs(tau - tau_ref).max() <= 1e-12
    m.close()


def test_full_size_properties(hip_lib):
    """N = 1M, S = 24 (the oracle's QR would take minutes): size-independent properties.
    (a) R^T R = A^T A (it is a QR of the shapes matrix); (b) a mesh equal to the rest pose gives
    zero weights and comes back bit-identical; (c) the weights are linear in the deformation;
    (d) re-initialising with the same shapes reproduces the factorisation bit for bit."""
    rng = np.random.default_rng(2)
    N, S = 1_000_000, 24
    rest = synth.head_mesh(N)
    bumps = rng.normal(size=(S, 3)).astype(np.float32)
    shapes = []
    for s in range(S):
        centre = rest[rng.integers(N)]
        wgt = np.exp(-np.sum((rest - centre) ** 2, axis=1) / 0.05).astype(np.float32)
        shapes.append((rest + wgt[:, None] * bumps[s] * np.float32(0.1)).astype(np.float32))
    m = capi.Morph()
    m.init(rest, shapes)
    QR, tau = m.qr()
    R = np.triu(QR[:S])
    A = np.stack([(s - rest).astype(np.float32).reshape(-1) for s in shapes], axis=1).astype(np.float64)
    G = A.T @ A
    assert np.abs(R.T @ R - G).max() <= 1e-10 * np.abs(G).max()
    out, w = m.apply(rest)
    assert np.all(w == 0.0) and np.array_equal(out, rest)
    P1 = (rest + 0.5 * (shapes[3] - rest)).astype(np.float32)
    P2 = (rest + 0.25 * (shapes[9] - rest)).astype(np.float32)
    d1, d2 = (P1 - rest).astype(np.float32), (P2 - rest).astype(np.float32)
    P12 = (rest + (d1 + d2)).astype(np.float32)
    _, w1 = m.apply(P1)
    _, w2 = m.apply(P2)
    _, w12 = m.apply(P12)
    d12 = (P12 - rest).astype(np.float32)
    # linear in the fp32 delta actually formed: compare with the weights of d1 + d2 as rounded
    resid = (d12.astype(np.float64) - d1 - d2).reshape(-1)
    assert np.abs(w12 - (w1 + w2) - resid @ QR).max() <= 1e-9 * max(1.0, np.abs(w12).max())
    m2 = capi.Morph()
    m2.init(rest, shapes)
    QR2, tau2 = m2.qr()
    assert np.array_equal(QR, QR2) and np.array_equal(tau, tau2)
    print(f"QR of {3 * N} x {S}: {m.last_init_ms:.1f} ms")
    m.close(); m2.close()


def test_cook_with_morph_space_follows_the_reference_sequence(hip_lib, oracle):
    """fdsop_cook with blendshapes on inputs 3..: setupBlends (:175-213), RBF pass, then the
    morph-space loop (:444-482) -- against the oracle chaining the same steps; plus the
    reference's messages and its isComputed() state machine (weights only on the first cook
    after the blendshapes were (re)initialised)."""
    from facedeform_amd.sop import FaceDeformSOP
    from oracle import fd_oracle as fo
    rng = np.random.default_rng(9)
    N, M, S = 20_011, 48, 9
    P = synth.head_mesh(N)
    rig_rest = synth.control_points(M, "head")
    rig_deform = synth.deformed_rig(rig_rest, 1)
    shapes = [(P + (0.08 * rng.normal(size=(N, 3)) * (rng.random((N, 1)) < 0.5)).astype(np.float32)).astype(np.float32)
              for _ in range(S)]
    node = FaceDeformSOP()
    node.set("kernel", 1)
    node.set("morphspace", 1)
    node.set("doclampweight", 1)
    node.set("weightrange", -0.01, 0)
    node.set("weightrange", 0.02, 1)
    node.set("dofalloff", 1)
    node.set("falloffradius", 0.25)
    # no blendshapes connected: the reference's warning, plain RBF result
    res0 = node.cook(P, rig_rest, rig_deform)
    assert "No blendshapes found. Ignoring morphspace deformation." in res0.warnings
    # first cook with blendshapes: init + weights + displacement
    res = node.cook(P, rig_rest, rig_deform, shapes=shapes, blends_changed=True)
    assert not [w for w in res.warnings if "morph" in w.lower() or "weights" in w.lower()], res.warnings
    assert res.weights.shape == (S,)
    # oracle: RBF deform, then DirectBSEdit on the deformed mesh with rest = incoming P
    table = oracle.control_table(rig_rest, rig_deform)
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [0.0], 0)
    assert rc == 0 and tt == 1
    P_def, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    A = oracle.morph_shapes_matrix(P, shapes)
    QR, _ = oracle.morph_qr(A)
    w_ref = oracle.morph_weights(QR, P_def, P)
    # the weights see the fp32 RBF displacement (1e-5 relative); they are sums of O(1e-2) terms
    assert np.abs(res.weights - w_ref).max() <= 2e-5 * max(1.0, np.abs(w_ref).max())
    assert (np.abs(3 * w_ref) > 0.02).any(), "the clamp must be exercised"
    ref = oracle.morph_displace(A, res.weights, res0.P, P, (-0.01, 0.02), True, 0.25)
    assert np.array_equal(res.P, ref)                         # same weights, same fp32 order
    ref2 = oracle.morph_displace(A, w_ref, P_def, P, (-0.01, 0.02), True, 0.25)
    assert _disp_ratio(res.P, ref2, P, 2e-5) <= 1.0
    # second cook, blendshapes unchanged: isComputed() is still true -> the warning branch (:448-453)
    res2 = node.cook(P, rig_rest, rig_deform, shapes=shapes)
    assert "Can't compute weights for morphspace deformation. Ingoring it." in res2.warnings
    assert res2.weights.size == 0 and np.array_equal(res2.P, res0.P)
    # a blendshape with another point count is dropped with the reference's warning (:200-204)
    res3 = node.cook(P, rig_rest, rig_deform, shapes=shapes[:3] + [shapes[3][:100]], blends_changed=True)
    assert "Some blendshapes don't match rest pose point count. Ignoring them." in res3.warnings
    assert res3.weights.shape == (3,)
    # input 0 carrying its own rest attribute: weights and the final position use it (:445-447, 471)
    own_rest = (P + np.float32(0.001)).astype(np.float32)
    res4 = node.cook(P, rig_rest, rig_deform, shapes=shapes, blends_changed=True, rest=own_rest)
    w4 = oracle.morph_weights(QR, res0.P, own_rest)
    assert np.abs(res4.weights - w4).max() <= 1e-9 * max(1.0, np.abs(w4).max())
    assert np.array_equal(res4.P, oracle.morph_displace(A, res4.weights, res0.P, own_rest, (-0.01, 0.02), True, 0.25))
