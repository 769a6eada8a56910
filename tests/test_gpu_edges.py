"""GPU: edge cases of the boundary -- empty and tiny inputs, one control point (closed form), the
largest supported system and one past it, degenerate morph / capture inputs."""
import numpy as np
import pytest

from conftest import parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu


def test_empty_and_tiny_meshes(hip_lib, oracle):
    rest = synth.control_points(20, "sphere")
    deform = synth.deformed_rig(rest)
    e = capi.Engine()
    e.set_points(rest, (deform - rest).astype(np.float32)); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
    e.build()
    out, fall = e.deform(np.zeros((0, 3), np.float32))
    assert out.shape == (0, 3) and fall.shape == (0,)
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    for n in (1, 2, 63, 65, 257):                       # below / across a wave, a tile group, a workgroup
        P = synth.sphere_mesh(1000)[:n].copy()
        tu, tv, nn = synth.tangent_frames(P)
        d2 = np.linspace(0.0, 0.6, n).astype(np.float32)
        out, fall = e.deform(P, dist2=d2, tangents=(tu, tv, nn), radius2=0.25, falloffrate=2.0)
        ref, ref_fall = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, dist2=d2, tangents=(tu, tv, nn),
                                      radius2=0.25, falloffrate=2.0)
        plain, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, dist2=d2, radius2=0.25, falloffrate=2.0)
        assert parity_ratio(out, ref, P, 1e-5, scale_out=plain) <= 1.0, n
        assert np.allclose(fall, ref_fall, rtol=2e-6, atol=1e-7), n
    e.close()


def test_single_control_point_closed_form(hip_lib):
    """One Gaussian centre, zero term: the weight is the delta itself and the displacement is
    delta * exp(-d^2 / R^2).  One centre, constant term, thin-plate: phi(0) = 0, so the system is
    [[0, 1], [1, 0]] -- weight 0, constant = delta: a rigid translation."""
    c = np.array([[0.1, -0.2, 0.3]], np.float32)
    delta = np.array([[0.05, 0.02, -0.04]], np.float32)
    P = synth.sphere_mesh(500)
    e = capi.Engine()
    e.set_points(c, delta); e.set_kernel(capi.KERNEL_GAUSSIAN, [0.8]); e.set_term(capi.TERM_ZERO)
    rep = e.build()
    assert rep.terminationtype == 1 and rep.n == 1
    W, radii = e.get_weights()
    assert np.allclose(W[0], delta[0].astype(np.float64), rtol=1e-15) and radii[0] == 0.8
    out, _ = e.deform(P)
    d2 = ((P.astype(np.float64) - c[0]) ** 2).sum(axis=1)
    ref = P + (np.exp(-d2 / 0.64)[:, None] * delta[0].astype(np.float64)).astype(np.float32)
    assert parity_ratio(out, ref.astype(np.float32), P, 1e-5) <= 1.0
    e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_CONST)
    rep = e.build()
    assert rep.terminationtype == 1 and rep.n == 2
    W, _ = e.get_weights()
    assert np.allclose(W[0], 0.0, atol=1e-18) and np.allclose(W[1], delta[0].astype(np.float64))
    out, _ = e.deform(P)
    assert np.abs(out - (P + delta[0])).max() <= 1e-7
    e.close()


def test_largest_supported_system_and_one_past_it(hip_lib):
    """M + 4 = 5632 is the order the single-workgroup panel and the LDS-resident back
    substitution are sized for (fd_internal.h kMaxOrder).  No oracle at this size (minutes on a
    CPU): the interpolation property -- every control point moves by its own delta -- checks the
    whole chain (NB = 4/8/16/32 panels, three trailing-update widths, ranged back-substitution)."""
    M = 5628
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 3)
    delta = (deform - rest).astype(np.float32)
    e = capi.Engine()
    e.set_points(rest, delta); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
    rep = e.build()
    assert rep.terminationtype == 1 and rep.n == M + 4 and rep.iterationscount == M + 4
    out, _ = e.deform(rest)
    got = out.astype(np.float64) - rest
    err = np.linalg.norm(got - delta, axis=1) / np.maximum(np.linalg.norm(delta, axis=1), 1e-5 * np.linalg.norm(delta, axis=1).max())
    assert err.max() <= 3e-5, err.max()          # fp32 evaluation over 5628 centres at the centres themselves
    print(f"order {rep.n}: assemble {rep.t_assemble_ms:.2f} ms + solve {rep.t_solve_ms:.2f} ms, max interpolation error {err.max():.2e}")
    with pytest.raises(capi.FdError) as ei:
        e.set_points(np.zeros((M + 1, 3), np.float32), np.zeros((M + 1, 3), np.float32))
    assert ei.value.code == capi.FD_E_INVALID and "exceeds" in str(ei.value)
    e.close()


def test_morph_without_shapes_and_capture_without_points(hip_lib):
    rest = synth.sphere_mesh(300)
    P = (rest + np.float32(0.01)).astype(np.float32)
    m = capi.Morph()
    m.init(rest, [])
    assert m.initialised
    out, w = m.apply(P)                                   # no shapes: P = rest + 0
    assert w.size == 0 and np.array_equal(out, rest)
    out, _ = m.apply(P, add_delta=True, falloffradius=0.5)
    assert np.array_equal(out, (rest + ((P - rest) * np.float32(0.5))).astype(np.float32))
    with pytest.raises(capi.FdError):
        m.init(rest[:2], [rest[:2]] * 7)                  # more shapes than rows
    m.close()
    e = capi.Engine()
    assert e.capture_dist2(np.zeros((0, 3), np.float32), np.zeros((1, 9), np.float32), 1.0).shape == (0,)
    e.close()


def test_poisoned_systems_fail_cleanly(hip_lib):
    """Matrices full of NaN must end in terminationtype -4 / -5, not in a wild write: coincident
    centres under the QNN rule give radius 0 and exp(-0/0) = NaN everywhere in two columns; a NaN
    or Inf control point poisons a whole row.  (The panel's pivot search sees no candidate in an
    all-NaN column; it used to decode the empty key into logical row 65535.)  The context must
    stay usable afterwards."""
    rest = synth.control_points(40, "sphere")
    deform = synth.deformed_rig(rest)
    delta = (deform - rest).astype(np.float32)
    e = capi.Engine()
    e.set_kernel(capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0]); e.set_term(0)
    bad = rest.copy(); bad[7] = bad[3]; bad[21] = bad[3]
    e.set_points(bad, delta)
    assert e.build(check=False).terminationtype == -5
    for poison in (np.nan, np.inf):
        bad = rest.copy(); bad[11, 1] = poison
        for kind, params in ((capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0]), (capi.KERNEL_THIN_PLATE, [])):
            e.set_kernel(kind, params)
            e.set_points(bad, delta)
            assert e.build(check=False).terminationtype in (-4, -5), (poison, kind)
    e.set_kernel(capi.KERNEL_THIN_PLATE)
    e.set_points(rest, delta)
    assert e.build().terminationtype == 1
    out, _ = e.deform(rest)
    assert np.abs((out - rest) - delta).max() <= 1e-5 * np.abs(delta).max() + 1e-7
    e.close()
