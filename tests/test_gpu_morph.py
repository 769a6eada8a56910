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
