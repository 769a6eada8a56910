"""GPU: the batched build (fd_batch_*) against single builds and the oracle.

One launch chain factorises all contexts of a batch; the kernels and their arithmetic are the
ones a single fd_build runs, so the bar is bit-identical weights, not a tolerance."""
import numpy as np
import pytest
import torch

from conftest import parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu


def _rig(M, frame, shift=0.0):
    """Rest rig and the fp32 deltas the SOP would hand over (deformed - rest, src/SOP_FaceDeform.cpp:276-281)."""
    rest = (synth.control_points(M, "head") + np.float32(shift)).astype(np.float32)
    deform = (rest + synth.smooth_deltas(rest, frame)).astype(np.float32)
    return rest, (deform - rest).astype(np.float32)


@pytest.mark.parametrize("kind,params,term", [
    (capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR),
    (capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0], capi.TERM_LINEAR),
    (capi.KERNEL_GAUSSIAN, [0.7, 1e-3], capi.TERM_CONST),
    (capi.KERNEL_BIHARMONIC, [], capi.TERM_ZERO),
])
def test_batch_equals_single_builds(hip_lib, oracle, kind, params, term):
    M, nb = 77, 5
    singles, batched = [], []
    for f in range(nb):
        rest, delta = _rig(M, f, 0.003 * f)            # different systems, not only different RHS
        for lst in (singles, batched):
            e = capi.Engine()
            e.set_points(rest, delta); e.set_kernel(kind, params); e.set_term(term)
            lst.append(e)
    for e in singles:
        assert e.build().terminationtype == 1
    b = capi.Batch(batched)
    assert len(b) == nb
    b.build_async()
    reps = b.build_result()
    assert [r.terminationtype for r in reps] == [1] * nb
    assert all(r.n == M + (4, 1, 0)[term] for r in reps)
    for es, eb in zip(singles, batched):
        Ws, rs = es.get_weights()
        Wb, rb = eb.get_weights()
        assert np.array_equal(Ws, Wb) and np.array_equal(rs, rb)
    # and against the oracle for one of them
    rest, delta = _rig(M, 2, 0.003 * 2)
    table = oracle.control_table(rest, rest + delta)
    rc, tt, W_ref, _ = oracle.build(table, kind, params, term)
    assert rc == 0 and tt == 1
    Wb, _ = batched[2].get_weights()
    assert np.abs(Wb - W_ref).max() <= 1e-8 * np.abs(W_ref).max()
    b.close()
    for e in singles + batched:
        e.close()


def test_batch_points_in_place_other_stream_and_deform(hip_lib, oracle):
    """Control points read from caller-owned device arrays; build on a stream that is not the
    contexts' own; evaluation on each context's stream has to wait for the batch."""
    M, nb, N = 256, 8, 30_011
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(N)
    d_P = torch.from_numpy(P).to(dev)
    rest = synth.control_points(M, "head")
    d_rest = torch.from_numpy(rest).to(dev)
    deltas = np.stack([(rest + synth.smooth_deltas(rest, f)).astype(np.float32) - rest for f in range(nb)]).astype(np.float32)
    d_deltas = torch.from_numpy(deltas).to(dev)
    torch.cuda.synchronize()
    engines = []
    for _ in range(nb):
        e = capi.Engine()
        e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
        engines.append(e)
    b = capi.Batch(engines)
    side = torch.cuda.Stream(device=dev)
    outs = [torch.empty_like(d_P) for _ in range(nb)]
    for rnd in range(2):                                     # second round replays the captured graph
        order = list(range(nb)) if rnd == 0 else list(reversed(range(nb)))
        b.set_points_dev([d_rest.data_ptr()] * nb, [d_deltas[f].data_ptr() for f in order], M)
        b.build_async(side.cuda_stream)
        for e, o in zip(engines, outs):
            e.deform_dev(N, d_P.data_ptr(), o.data_ptr())     # on e's own stream
        reps = b.build_result()
        assert [r.terminationtype for r in reps] == [1] * nb
        for e in engines:
            e.synchronize()
        for k, f in enumerate(order):
            table = oracle.control_table(rest, rest + deltas[f])
            rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
            ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P[::7])
            out = outs[k].cpu().numpy()[::7]
            assert parity_ratio(out, ref, P[::7], 1e-5) <= 1.0, (rnd, k)
    # a context of the batch can still be rebuilt on its own: the batch mirrored its points
    e = engines[3]
    W_before, _ = e.get_weights()
    assert e.build().terminationtype == 1
    W_after, _ = e.get_weights()
    assert np.array_equal(W_before, W_after)
    b.close()
    for e in engines:
        e.close()


def test_batch_rejects_mismatched_contexts(hip_lib):
    rest, delta = _rig(40, 0)
    a, c = capi.Engine(), capi.Engine()
    a.set_points(rest, delta); a.set_kernel(capi.KERNEL_THIN_PLATE); a.set_term(0)
    c.set_points(rest[:30], delta[:30]); c.set_kernel(capi.KERNEL_THIN_PLATE); c.set_term(0)
    b = capi.Batch([a, c])
    with pytest.raises(capi.FdError) as ei:
        b.build_async()
    assert ei.value.code == capi.FD_E_INVALID and "differs" in str(ei.value)
    c.set_points(rest, delta); c.set_term(capi.TERM_CONST)
    with pytest.raises(capi.FdError):
        b.build_async()
    with pytest.raises(capi.FdError):
        capi.Batch([a, a])
    b.close(); a.close(); c.close()


def test_batch_reports_singular_member(hip_lib):
    """One coincident-point rig in the batch: that context reports -5, the others solve."""
    rest, delta = _rig(48, 1)
    bad = rest.copy(); bad[7] = bad[3]
    es = []
    for r in (rest, bad, rest):
        e = capi.Engine()
        e.set_points(r, delta); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
        es.append(e)
    b = capi.Batch(es)
    b.build_async()
    reps = b.build_result(check=False)
    assert [r.terminationtype for r in reps] == [1, -5, 1]
    b.close()
    for e in es:
        e.close()


@pytest.mark.parametrize("kind,params,M", [
    (capi.KERNEL_THIN_PLATE, [], 200),              # matrix-pipe kernel, batched
    (capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0], 200),    # packed-VALU kernel, batched (the SOP's default model)
    (capi.KERNEL_THIN_PLATE, [], 40),               # thin-plate below the matrix-pipe threshold
    (capi.KERNEL_CUBIC, [], 96),
])
def test_batched_evaluation_is_bit_identical_to_single_launches(hip_lib, kind, params, M):
    """fd_batch_deform_dev: one launch for all contexts (grid y = context).  Same kernel body as a
    single launch, so every context's result must match fd_deform_dev bit for bit -- with and
    without gate / fall-off / tangent frames, ragged N, and through the fallback (mixed kernels)."""
    nb, N = 5, 70_003
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    tu, tv, nn = synth.tangent_frames(P)
    d_P = torch.from_numpy(P).to(dev)
    d_d2 = torch.from_numpy((rng.random(N) * 0.6).astype(np.float32)).to(dev)
    d_tan = [torch.from_numpy(a).to(dev) for a in (tu, tv, nn)]
    engines = []
    for f in range(nb):
        e = capi.Engine()
        e.set_kernel(kind, params); e.set_term(0)
        e.set_points(rest, synth.smooth_deltas(rest, f).astype(np.float32))
        engines.append(e)
    b = capi.Batch(engines)
    b.build_async(); b.build_result()
    stream = torch.cuda.Stream(device=dev)
    for use_d2, use_tan in ((False, False), (True, False), (True, True)):
        outs = [torch.empty_like(d_P) for _ in range(nb)]
        falls = [torch.full((N,), 2.0, device=dev) for _ in range(nb)]
        refs = [torch.empty_like(d_P) for _ in range(nb)]
        ref_falls = [torch.full((N,), 2.0, device=dev) for _ in range(nb)]
        torch.cuda.synchronize()
        b.deform_dev(N, [d_P.data_ptr()] * nb, [o.data_ptr() for o in outs],
                     [d_d2.data_ptr()] * nb if use_d2 else None, [f.data_ptr() for f in falls],
                     [[t.data_ptr()] * nb for t in d_tan] if use_tan else None, radius2=0.3, falloffrate=1.5,
                     stream_ptr=stream.cuda_stream)
        stream.synchronize()
        for e, r, rf in zip(engines, refs, ref_falls):
            e.deform_dev(N, d_P.data_ptr(), r.data_ptr(), d_dist2=d_d2.data_ptr() if use_d2 else 0, d_falloff=rf.data_ptr(),
                         d_tu=d_tan[0].data_ptr() if use_tan else 0, d_tv=d_tan[1].data_ptr() if use_tan else 0,
                         d_nrm=d_tan[2].data_ptr() if use_tan else 0, radius2=0.3, falloffrate=1.5)
            e.synchronize()
        for k in range(nb):
            assert torch.equal(outs[k], refs[k]) and torch.equal(falls[k], ref_falls[k]), (use_d2, use_tan, k)
        assert not torch.equal(outs[0], outs[1])            # different models did give different results
    # fallback: one context on another kernel -> single launches, same results
    engines[2].set_kernel(capi.KERNEL_BIHARMONIC if kind != capi.KERNEL_BIHARMONIC else capi.KERNEL_CUBIC); engines[2].build()
    outs = [torch.empty_like(d_P) for _ in range(nb)]
    b.deform_dev(N, [d_P.data_ptr()] * nb, [o.data_ptr() for o in outs], stream_ptr=stream.cuda_stream)
    stream.synchronize()
    ref = torch.empty_like(d_P)
    for k in (1, 2):
        engines[k].deform_dev(N, d_P.data_ptr(), ref.data_ptr()); engines[k].synchronize()
        assert torch.equal(outs[k], ref)
    b.close()
    for e in engines:
        e.close()


def test_large_batches_group_panels(hip_lib, oracle):
    """From four systems up a batched build groups its panels under deep trailing updates (they
    are HBM-bound there).  Same pivots, another summation order: the weights agree with a single
    build to rounding, with the oracle as usual, and fd_set_deltas on such a factorisation replays
    the grouped sequence bit for bit."""
    M, nb = 1100, 4                  # 32-wide panels above 512 rows are paired, 16-wide ones grouped by four
    rest = synth.control_points(M, "head")
    # deltas as the SOP forms them: float32(deformed - rest)
    d = [((rest + synth.smooth_deltas(rest, f)).astype(np.float32) - rest).astype(np.float32) for f in range(nb + 1)]
    es = []
    for f in range(nb):
        e = capi.Engine(); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
        e.set_points(rest, d[f]); es.append(e)
    b = capi.Batch(es)
    b.build_async(); reps = b.build_result()
    assert [r.terminationtype for r in reps] == [1] * nb
    single = capi.Engine(); single.set_kernel(capi.KERNEL_THIN_PLATE); single.set_term(0)
    single.set_points(rest, d[2]); single.build()
    Wb, _ = es[2].get_weights(); Ws, _ = single.get_weights()
    scale = np.abs(Ws).max()
    assert np.abs(Wb - Ws).max() <= 1e-10 * scale
    table = oracle.control_table(rest, (rest + synth.smooth_deltas(rest, 2)).astype(np.float32))
    assert np.array_equal(table[:, 3:], d[2].astype(np.float64))
    rc, tt, W_ref, _ = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    assert np.abs(Wb - W_ref).max() <= 1e-8 * scale
    # new deltas through the grouped factorisation == a fresh batched build with those deltas
    es[1].set_deltas(d[nb]); assert es[1].build().terminationtype == 1
    W_fast, _ = es[1].get_weights()
    es2 = []
    for f in range(nb):
        e = capi.Engine(); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
        e.set_points(rest, d[nb] if f == 1 else d[f]); es2.append(e)
    b2 = capi.Batch(es2); b2.build_async(); b2.build_result()
    assert np.array_equal(W_fast, es2[1].get_weights()[0])
    b.close(); b2.close(); single.close()
    for e in es + es2:
        e.close()
