"""GPU: fd_set_deltas -- new right-hand sides through the stored factorisation (the animated-rig
case: rest rig fixed, deformed rig moving).  The bar is bit-identity with a full rebuild: the
right-hand-side block goes through the same kernels with the same operands in the same order."""
import time
import numpy as np
import pytest
import torch

from facedeform_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,kind,params,term", [
    (32, capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR),
    (256, capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR),
    (300, capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0], capi.TERM_CONST),
    (1100, capi.KERNEL_BIHARMONIC, [1e-3], capi.TERM_ZERO),      # 32- and 16-wide panels, ranged back-substitution
    (2044, capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR),       # the largest order the fast path takes
])
def test_resolve_is_bit_identical_to_rebuild(hip_lib, M, kind, params, term):
    rest = synth.control_points(M, "head")
    d = [synth.smooth_deltas(rest, f).astype(np.float32) for f in range(3)]
    P = synth.head_mesh(5000)
    fast, full = capi.Engine(), capi.Engine()
    for e in (fast, full):
        e.set_kernel(kind, params); e.set_term(term)
    fast.set_points(rest, d[0])
    assert fast.build().terminationtype == 1
    for f in (1, 2, 0):
        fast.set_deltas(d[f])
        rep = fast.build()
        assert rep.terminationtype == 1 and rep.n == M + (4, 1, 0)[term]
        full.set_points(rest, d[f])
        assert full.build().terminationtype == 1
        Wf, rf = fast.get_weights()
        Wr, rr = full.get_weights()
        assert np.array_equal(Wf, Wr) and np.array_equal(rf, rr), (M, f)
        of, _ = fast.deform(P)
        orr, _ = full.deform(P)
        assert np.array_equal(of, orr)
    fast.close(); full.close()


def test_when_the_factorisation_cannot_be_reused(hip_lib):
    rest = synth.control_points(40, "sphere")
    d0 = synth.smooth_deltas(rest, 0).astype(np.float32)
    e = capi.Engine()
    e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
    with pytest.raises(capi.FdError) as ei:
        e.set_deltas(d0)                                  # nothing built yet
    assert ei.value.code == capi.FD_E_NOT_BUILT
    e.set_points(rest, d0); e.build()
    e.set_deltas(d0); e.build()                           # fine
    with pytest.raises(capi.FdError):
        e.set_deltas(d0[:30])                             # another M
    e.set_term(capi.TERM_CONST)                           # the matrix changes: factorisation discarded
    with pytest.raises(capi.FdError):
        e.set_deltas(d0)
    e.set_points(rest, d0); e.build()
    e.set_kernel(capi.KERNEL_CUBIC)
    with pytest.raises(capi.FdError):
        e.set_deltas(d0)
    # a failed factorisation stays failed through fd_set_deltas (its flags are kept)
    bad = rest.copy(); bad[5] = bad[1]
    e.set_kernel(capi.KERNEL_THIN_PLATE)
    e.set_points(bad, d0)
    assert e.build(check=False).terminationtype == -5
    e.set_deltas(d0)
    assert e.build(check=False).terminationtype == -5
    # too large for the fast path
    big = synth.control_points(2100, "head")
    e.set_points(big, synth.smooth_deltas(big, 0).astype(np.float32)); e.build()
    with pytest.raises(capi.FdError) as ei:
        e.set_deltas(synth.smooth_deltas(big, 1).astype(np.float32))
    assert ei.value.code == capi.FD_E_INVALID and "2048" in str(ei.value)
    e.close()


def test_device_pointers_batch_factorisation_and_latency(hip_lib):
    """Deltas from device memory; a factorisation left by a BATCHED build is reused as well."""
    M, N = 256, 1_000_000
    dev = torch.device("cuda", 0)
    rest = synth.control_points(M, "head")
    deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(4)]).astype(np.float32)
    d_rest = torch.from_numpy(rest).to(dev); d_deltas = torch.from_numpy(deltas).to(dev)
    d_P = torch.from_numpy(synth.head_mesh(N)).to(dev); d_out = torch.empty_like(d_P)
    torch.cuda.synchronize()
    es = [capi.Engine() for _ in range(3)]
    for e in es:
        e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
    b = capi.Batch(es)
    b.set_points_dev([d_rest.data_ptr()] * 3, [d_deltas[f].data_ptr() for f in range(3)], M)
    b.build_async(); b.build_result()
    ref = capi.Engine(); ref.set_kernel(capi.KERNEL_THIN_PLATE); ref.set_term(0)
    ref.set_points(rest, deltas[3]); ref.build()
    es[1].set_deltas_dev(d_deltas[3].data_ptr(), M)
    assert es[1].build().terminationtype == 1
    assert np.array_equal(es[1].get_weights()[0], ref.get_weights()[0])
    # single-cook latency with and without the factorisation (device-resident inputs)
    e = es[0]
    def cook(full, f):
        if full:
            e.set_points_dev(d_rest.data_ptr(), d_deltas[f % 4].data_ptr(), M)
        else:
            e.set_deltas_dev(d_deltas[f % 4].data_ptr(), M)
        e.build_async()
        e.deform_dev(N, d_P.data_ptr(), d_out.data_ptr())
        e.build_result(); e.synchronize()
    res = {}
    for full in (True, False):
        cook(True, 0)
        ts = []
        for f in range(12):
            t0 = time.perf_counter(); cook(full, f); ts.append(time.perf_counter() - t0)
        res[full] = sorted(ts)[len(ts) // 2] * 1e3
    print(f"single cook, 1M vertices x 256 control points: full rebuild {res[True]:.3f} ms, fd_set_deltas {res[False]:.3f} ms")
    assert res[False] < res[True]
    b.close(); ref.close()
    for e in es:
        e.close()


def test_cook_with_unchanged_rest_rig(hip_lib):
    """fdsop_cook with rig_rest_unchanged: the second cook takes fd_set_deltas and gives exactly
    what a fresh node computes; a changed kernel parm or point count silently falls back."""
    from facedeform_amd.sop import FaceDeformSOP
    P = synth.head_mesh(20_000)
    rest = synth.control_points(64, "head")
    node = FaceDeformSOP(); node.set("kernel", 1)
    node.cook(P, rest, synth.deformed_rig(rest, 0))
    res = node.cook(P, rest, synth.deformed_rig(rest, 2), rig_rest_unchanged=True)
    fresh = FaceDeformSOP(); fresh.set("kernel", 1)
    ref = fresh.cook(P, rest, synth.deformed_rig(rest, 2))
    assert res.severity == ref.severity and res.infos == ref.infos
    assert np.array_equal(res.P, ref.P) and np.array_equal(res.fd_falloff, ref.fd_falloff)
    # the flag is a promise about the rest rig only: other changes are caught by the engine
    node.set("kernel", 2)
    res2 = node.cook(P, rest, synth.deformed_rig(rest, 1), rig_rest_unchanged=True)
    fresh.set("kernel", 2)
    assert np.array_equal(res2.P, fresh.cook(P, rest, synth.deformed_rig(rest, 1)).P)
    rest2 = synth.control_points(50, "head")
    res3 = node.cook(P, rest2, synth.deformed_rig(rest2, 1), rig_rest_unchanged=True)
    assert np.array_equal(res3.P, fresh.cook(P, rest2, synth.deformed_rig(rest2, 1)).P)


def test_cook_with_unchanged_mesh(hip_lib):
    """mesh_unchanged: the second cook evaluates the device-resident mesh; same bits as a fresh
    node.  Toggling the tangent parm or the dist2 input changes what the cache must hold and
    re-uploads silently."""
    from facedeform_amd.sop import FaceDeformSOP
    P = synth.head_mesh(30_000)
    rest = synth.control_points(64, "head")
    tu, tv, nn = synth.tangent_frames(P)
    d2 = (0.5 * np.abs(P[:, 1])).astype(np.float32)
    node, fresh = FaceDeformSOP(), FaceDeformSOP()
    for n in (node, fresh):
        n.set("kernel", 1); n.set("radius", 0.6)
    node.cook(P, rest, synth.deformed_rig(rest, 0), dist2=d2)
    res = node.cook(P, rest, synth.deformed_rig(rest, 1), dist2=d2, mesh_unchanged=True, rig_rest_unchanged=True)
    ref = fresh.cook(P, rest, synth.deformed_rig(rest, 1), dist2=d2)
    assert np.array_equal(res.P, ref.P) and np.array_equal(res.fd_falloff, ref.fd_falloff)
    for n in (node, fresh):
        n.set("tangent", 1)
    res = node.cook(P, rest, synth.deformed_rig(rest, 2), dist2=d2, tangentu=tu, tangentv=tv, N=nn, mesh_unchanged=True)
    ref = fresh.cook(P, rest, synth.deformed_rig(rest, 2), dist2=d2, tangentu=tu, tangentv=tv, N=nn)
    assert np.array_equal(res.P, ref.P)
    res = node.cook(P, rest, synth.deformed_rig(rest, 2), tangentu=tu, tangentv=tv, N=nn, mesh_unchanged=True)   # dist2 gone
    ref = fresh.cook(P, rest, synth.deformed_rig(rest, 2), tangentu=tu, tangentv=tv, N=nn)
    assert np.array_equal(res.P, ref.P) and res.warnings == ref.warnings
