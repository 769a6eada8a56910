"""GPU: frames that share the mesh and the rest rig (fd_batch_deform_shared_dev, kernel
k_deform32_tps_shared): phi once per (vertex, centre), the weight contraction on the matrix pipe
from fp16 x 2 split operands.  Not bit-identical to the one-frame kernels by construction; the bar
is the oracle's, 1e-5 of the displacement (conftest.parity_ratio and the stricter l2_parity_ulp),
for every frame, with gate, fall-off, tangent frames, ragged sizes and vertices sitting on centres."""
import numpy as np
import pytest
import torch

from conftest import l2_parity, l2_parity_ulp, parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _setup(M, N, F, kind=capi.KERNEL_THIN_PLATE, params=(), noise=False):
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(max(N, 100_000))[:: max(N, 100_000) // N][:N].copy()
    rest = synth.control_points(M, "head")
    P[:6] = rest[:6]                                   # vertices on centres: d2 == 0
    # frames as SURVEY.md 8d lays them out: same mesh and rest rig, deltas phase-shifted (C4's eight
    # phases; further frames repeat them at another amplitude).  Phases whose displacement field passes
    # through zero at some vertices (0.3 * 19 rad does) are the province of
    # tests/tools/scaled_delta_parity.py: there EVERY fp32 kernel, this one, the one-frame ones and the
    # all-VALU one alike, is limited by cancellation (sum|phi w| / |d| up to 450) -- DESIGN.md 4.1.
    deltas = np.stack([synth.smooth_deltas(rest, f % 8) * np.float32(1.0 + 0.25 * (f // 8)) for f in range(F)]).astype(np.float32)
    d_P = torch.from_numpy(P).to(dev)
    d_rest = torch.from_numpy(rest).to(dev)
    d_deltas = torch.from_numpy(deltas).to(dev)
    engines = []
    for _ in range(F):
        e = capi.Engine()
        e.set_kernel(kind, params); e.set_term(capi.TERM_LINEAR)
        engines.append(e)
    batch = capi.Batch(engines)
    batch.set_points_dev([d_rest.data_ptr()] * F, [d_deltas.data_ptr() + f * M * 12 for f in range(F)], M)
    batch.build_async()
    assert [r.terminationtype for r in batch.build_result()] == [1] * F
    return dev, P, rest, deltas, d_P, (d_rest, d_deltas), engines, batch


def _close(engines, batch):
    batch.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("M,N,F", [(48, 4_099, 3), (256, 20_011, 8), (256, 70_000, 32), (100, 1_000, 1), (800, 3_001, 13), (256, 9_001, 20), (1400, 2_500, 27)])
def test_shared_frames_match_oracle(hip_lib, oracle, M, N, F):
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F)
    rng = np.random.default_rng(M + F)
    tu, tv, nn = synth.tangent_frames(P)
    r2 = np.float32(0.49)
    dist2 = (rng.random(N) * 0.6).astype(np.float32)
    dist2[::9] = -1.0
    d_d2 = torch.from_numpy(dist2).to(dev)
    d_t = [torch.from_numpy(a).to(dev) for a in (tu, tv, nn)]
    outs = [torch.empty_like(d_P) for _ in range(F)]
    falls = [torch.full((N,), 7.0, device=dev) for _ in range(F)]
    for mode in ("plain", "gate", "frames"):
        kw = {}
        okw = {}
        if mode != "plain":
            kw.update(d_dist2=d_d2.data_ptr(), radius2=r2, falloffrate=1.5)
            okw.update(dist2=dist2, radius2=r2, falloffrate=1.5)
        if mode == "frames":
            kw.update(d_tangents=[t.data_ptr() for t in d_t])
            okw.update(tangents=(tu, tv, nn))
        for fl in falls:
            fl.fill_(7.0)
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs], d_falloff=[f.data_ptr() for f in falls], **kw)
        torch.cuda.synchronize()
        for f in range(F):
            table = oracle.control_table(rest, (rest + deltas[f]).astype(np.float32))
            rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
            ref, ref_fall = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, **okw)
            plain = None
            if mode == "frames":
                plain, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, **{k: v for k, v in okw.items() if k != "tangents"})
            out = outs[f].cpu().numpy()
            ratio = parity_ratio(out, ref, P, TOL, scale_out=plain)
            assert ratio <= 1.0, (mode, f, ratio)
            if mode == "plain":
                assert l2_parity_ulp(out, ref, P, TOL) <= 1.0, (f, l2_parity_ulp(out, ref, P, TOL), l2_parity(out, ref, P))
            fall = falls[f].cpu().numpy()
            gated = dist2 > r2 if mode != "plain" else np.zeros(N, bool)
            assert np.array_equal(out[gated], P[gated]) and np.all(fall[gated] == 7.0)      # B2: untouched
            assert np.allclose(fall[~gated], ref_fall[~gated], rtol=2e-6, atol=1e-7)
    _close(engines, batch)


def test_shared_c2_full_size_and_against_the_one_frame_kernel(hip_lib, oracle):
    """C2's sizes, 8 frames: every frame sampled against the oracle (raw 8d metric where it holds,
    see tests/test_gpu_configs.py) and against the one-frame kernel's output, which it must agree
    with far inside the tolerance (both sit within ~5e-6 of the oracle)."""
    M, N, F = 256, 1_000_000, 8
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F)
    outs = [torch.empty_like(d_P) for _ in range(F)]
    single = [torch.empty_like(d_P) for _ in range(F)]
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs])
    batch.deform_dev(N, [d_P.data_ptr()] * F, [o.data_ptr() for o in single])
    torch.cuda.synchronize()
    idx = np.unique(np.concatenate([np.arange(0, N, 397), [0, 1, 5, 6, 63, 64, 511, 512, N - 1]]))
    for f in range(F):
        table = oracle.control_table(rest, (rest + deltas[f]).astype(np.float32))
        rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
        ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P[idx])
        out = outs[f].cpu().numpy()
        assert parity_ratio(out[idx], ref, P[idx], TOL) <= 1.0 and l2_parity_ulp(out[idx], ref, P[idx], TOL) <= 1.0, f
        one = single[f].cpu().numpy()
        d = np.linalg.norm(one.astype(np.float64) - P, axis=1)
        assert np.abs(out.astype(np.float64) - one).max() <= 1e-5 * d.max(), f
    _close(engines, batch)


def test_shared_needs_one_rest_rig_and_falls_back_for_other_kernels(hip_lib):
    M, N, F = 64, 5_000, 4
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F)
    outs = [torch.empty_like(d_P) for _ in range(F)]
    # a context whose rest points came from another array: refused, not silently wrong
    other = torch.from_numpy(rest.copy()).to(dev)
    batch.set_points_dev([keep[0].data_ptr()] * (F - 1) + [other.data_ptr()], [keep[1].data_ptr()] * F, M)
    batch.build_async(); batch.build_result()
    with pytest.raises(capi.FdError) as ei:
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs])
    assert ei.value.code == capi.FD_E_INVALID and "one rest rig" in ei.value.text
    _close(engines, batch)
    # a kernel the shared launch does not take (biharmonic: sqrt(d2) needs more of d2 near a centre than the
    # expanded form keeps) on shared arrays: the per-frame kernels, bit for bit
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F, kind=capi.KERNEL_BIHARMONIC, params=[])
    a = [torch.empty_like(d_P) for _ in range(F)]
    b = [torch.empty_like(d_P) for _ in range(F)]
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in a])
    batch.deform_dev(N, [d_P.data_ptr()] * F, [o.data_ptr() for o in b])
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    _close(engines, batch)


@pytest.mark.parametrize("F", [16, 32])       # 16: the 16-row kernel; 32: the 32-row kernel (what bench.py's pipeline launches)
def test_contexts_may_be_rebuilt_once_the_launch_has_its_copy(hip_lib, oracle, F):
    """fd_batch_wait_consumed: the shared-rig launch reads the contexts' models only in its first small kernel (weights,
    centre tiles and normalisation go into the batch's scratch).  A lane that waits for that point -- not for the
    evaluation -- and then builds the NEXT group's models on the same contexts must leave the evaluation in flight
    untouched: its frames still match the oracle for the FIRST group's deltas, and the second evaluation the second's."""
    N, M = 400_000, 256
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    d_P = torch.from_numpy(P).to(dev)
    d_rest = torch.from_numpy(rest).to(dev)
    groups = [np.stack([synth.smooth_deltas(rest, f % 8) * np.float32(1.0 + 0.25 * (f // 8)) for f in range(g * F, (g + 1) * F)])
              for g in range(2)]
    d_del = [torch.from_numpy(d).to(dev) for d in groups]
    lane, es = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    engines = []
    for _ in range(F):
        e = capi.Engine(); e.set_stream(lane.cuda_stream); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
        engines.append(e)
    batch = capi.Batch(engines)
    outs = [[torch.empty_like(d_P) for _ in range(F)] for _ in range(2)]
    falls = [[torch.zeros(N, device=dev) for _ in range(F)] for _ in range(2)]
    torch.cuda.synchronize()
    built = torch.cuda.Event()
    for g in range(2):
        batch.wait_consumed(lane.cuda_stream)                      # no-op the first time
        batch.set_points_dev([d_rest.data_ptr()] * F, [d_del[g].data_ptr() + f * M * 12 for f in range(F)], M)
        batch.build_async(lane.cuda_stream)
        built.record(lane)
        es.wait_event(built)
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs[g]], d_falloff=[x.data_ptr() for x in falls[g]],
                                stream_ptr=es.cuda_stream)
    torch.cuda.synchronize()
    idx = np.unique(np.concatenate([np.arange(0, N, 397), [0, 63, 64, N - 1]]))
    for g in range(2):
        for f in (0, 5, F - 1):
            table = oracle.control_table(rest, (rest + groups[g][f]).astype(np.float32))
            rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
            ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P[idx])
            out = outs[g][f].cpu().numpy()[idx]
            assert parity_ratio(out, ref, P[idx], 1e-5) <= 1.0, (g, f)
    batch.close()
    for e in engines:
        e.set_stream(None); e.close()


@pytest.mark.parametrize("kind,okind,params,M,N,F", [
    (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN, (1.0, 5.0), 256, 30_011, 32),      # the SOP's default model, dense layout
    (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN, (1.0, 5.0), 100, 4_099, 5),        # padded layout, ragged sizes
    (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN, (0.15,), 256, 10_000, 16),                 # one radius for all centres
    (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN, (1.0, 5.0), 256, 10_000, 16),      # 16 frames: three tiles per component block
    (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN, (2.0, 5.0), 512, 6_000, 24),
])
def test_shared_frames_of_the_gaussian_models(hip_lib, oracle, kind, okind, params, M, N, F):
    """The SOP's default model (QNN radii) and the fixed-radius Gaussian through the shared-rig launch: exp(-d2 / R_j^2)
    from direct coordinate differences, formed once per (vertex, centre) for all frames, contracted on the matrix
    pipe like the thin-plate phi.  Against the oracle (1e-5) with the gate and the fall-off in play, and against
    the one-frame kernel on the same models."""
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F, kind=kind, params=params)
    rng = np.random.default_rng(7 * M + F)
    r2 = np.float32(0.36)
    dist2 = (rng.random(N) * 0.5).astype(np.float32)
    dist2[::11] = -1.0
    d_d2 = torch.from_numpy(dist2).to(dev)
    outs = [torch.empty_like(d_P) for _ in range(F)]
    single = [torch.empty_like(d_P) for _ in range(F)]
    falls = [torch.full((N,), 7.0, device=dev) for _ in range(F)]
    for mode in ("plain", "gate"):
        kw, okw = {}, {}
        if mode == "gate":
            kw.update(d_dist2=d_d2.data_ptr(), radius2=r2, falloffrate=2.0)
            okw.update(dist2=dist2, radius2=r2, falloffrate=2.0)
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs], d_falloff=[f.data_ptr() for f in falls], **kw)
        torch.cuda.synchronize()
        for f in sorted(set([0, 1, F // 2, F - 1])):
            table = oracle.control_table(rest, (rest + deltas[f]).astype(np.float32))
            rc, tt, W, radii = oracle.build(table, okind, list(params), fo.TERM_LINEAR)
            assert rc == 0
            ref, ref_fall = oracle.deform(table, okind, radii, W, P, **okw)
            out = outs[f].cpu().numpy()
            assert parity_ratio(out, ref, P, TOL) <= 1.0, (mode, f, parity_ratio(out, ref, P, TOL))
            gated = dist2 > r2 if mode == "gate" else np.zeros(N, bool)
            assert np.array_equal(out[gated], P[gated])
            assert np.allclose(falls[f].cpu().numpy()[~gated], ref_fall[~gated], rtol=2e-6, atol=1e-7)
    batch.deform_dev(N, [d_P.data_ptr()] * F, [o.data_ptr() for o in single], d_dist2=[d_d2.data_ptr()] * F, radius2=r2, falloffrate=2.0)
    torch.cuda.synchronize()
    for f in (0, F - 1):
        a, b = outs[f].cpu().numpy(), single[f].cpu().numpy()
        assert parity_ratio(a, b, P, TOL) <= 1.0, f
    _close(engines, batch)


@pytest.mark.parametrize("F", [16, 32])
def test_prepared_sets_give_the_same_bits_and_survive_a_pipeline(hip_lib, F):
    """fd_batch_prepare_shared packs the models on the build stream; the evaluation that follows with the same
    outputs launches alone.  Three groups cooked back to back on ONE batch, as a lane of bench.py cooks them (build,
    prepare on the lane stream; evaluate on another stream that only waits for the lane), must equal the same
    groups cooked one at a time with the launch packing for itself -- bit for bit: the two scratch sets are used in
    turn and a set is not rewritten while the evaluation that reads it is still running."""
    N, M, G = 300_000, 256, 3
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    d_P = torch.from_numpy(P).to(dev)
    d_rest = torch.from_numpy(rest).to(dev)
    groups = [np.stack([synth.smooth_deltas(rest, (g * F + f) % 8) * np.float32(1.0 + 0.125 * ((g * F + f) // 8)) for f in range(F)])
              for g in range(G)]
    d_del = [torch.from_numpy(d).to(dev) for d in groups]
    lane, es = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    engines = []
    for _ in range(F):
        e = capi.Engine(); e.set_stream(lane.cuda_stream); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
        engines.append(e)
    batch = capi.Batch(engines)
    outs = [[torch.empty_like(d_P) for _ in range(F)] for _ in range(G)]
    falls = [[torch.zeros(N, device=dev) for _ in range(F)] for _ in range(G)]
    ref_out = [torch.empty_like(d_P) for _ in range(F)]
    ref_fall = [torch.zeros(N, device=dev) for _ in range(F)]
    torch.cuda.synchronize()
    built = torch.cuda.Event()
    for g in range(G):                                   # pipelined: nothing waits for an evaluation
        batch.set_points_dev([d_rest.data_ptr()] * F, [d_del[g].data_ptr() + f * M * 12 for f in range(F)], M)
        batch.build_async(lane.cuda_stream)
        batch.prepare_shared([o.data_ptr() for o in outs[g]], d_falloff=[x.data_ptr() for x in falls[g]], stream_ptr=lane.cuda_stream)
        built.record(lane)
        es.wait_event(built)
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs[g]], d_falloff=[x.data_ptr() for x in falls[g]],
                                stream_ptr=es.cuda_stream)
    torch.cuda.synchronize()
    assert [r.terminationtype for r in batch.build_result()] == [1] * F
    for g in range(G):                                   # one at a time, the launch packs for itself
        batch.set_points_dev([d_rest.data_ptr()] * F, [d_del[g].data_ptr() + f * M * 12 for f in range(F)], M)
        batch.build_async(lane.cuda_stream)
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in ref_out], d_falloff=[x.data_ptr() for x in ref_fall],
                                stream_ptr=lane.cuda_stream)
        torch.cuda.synchronize()
        for f in range(F):
            assert torch.equal(outs[g][f], ref_out[f]), (g, f)
            assert torch.equal(falls[g][f], ref_fall[f]), (g, f)
    batch.close()
    for e in engines:
        e.set_stream(None); e.close()


@pytest.mark.parametrize("kind,params", [(capi.KERNEL_GAUSSIAN_QNN, (1.0, 5.0)), (capi.KERNEL_THIN_PLATE, ())])
def test_repeated_launches_give_the_same_bits_at_full_size(hip_lib, kind, params):
    """A guard against the class of fault found while bringing up the Gaussian kinds (packed fp32 arithmetic
    software-pipelined under matrix instructions: wrong values on a few vertices per million, different ones per
    launch, no fault).  Anything of that kind shows as a difference between two launches of the same
    work: 1M vertices x 32 frames through the shared-rig launch and 1M vertices through the one-frame kernel, five
    times each, must agree bit for bit."""
    N, M, F = 1_000_000, 256, 32
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F, kind=kind, params=params)
    first = [torch.empty_like(d_P) for _ in range(F)]
    again = [torch.empty_like(d_P) for _ in range(F)]
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in first])
    torch.cuda.synchronize()
    for rep in range(4):
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in again])
        torch.cuda.synchronize()
        for f in range(F):
            assert torch.equal(first[f], again[f]), (rep, f, int((first[f] != again[f]).any(dim=1).sum()))
    one, two = torch.empty_like(d_P), torch.empty_like(d_P)
    engines[3].deform_dev(N, d_P.data_ptr(), one.data_ptr())
    engines[3].synchronize()
    for rep in range(12):
        engines[3].deform_dev(N, d_P.data_ptr(), two.data_ptr())
        engines[3].synchronize()
        assert torch.equal(one, two), (rep, int((one != two).any(dim=1).sum()))
    # and the one-frame kernels batched (grid y = frame): packed fmas beside d2 matrix instructions in the thin-plate one
    batch.deform_dev(N, [d_P.data_ptr()] * F, [o.data_ptr() for o in first])
    torch.cuda.synchronize()
    assert torch.equal(first[3], one)
    for rep in range(3):
        batch.deform_dev(N, [d_P.data_ptr()] * F, [o.data_ptr() for o in again])
        torch.cuda.synchronize()
        for f in range(F):
            assert torch.equal(first[f], again[f]), (rep, f)
    _close(engines, batch)


def test_a_member_rebuilt_on_its_own_is_not_evaluated_from_a_stale_copy(hip_lib, oracle):
    """The launch evaluates from the batch's packed copy of the weights and reuses that copy when nothing changed.
    A context rebuilt on its own (not through the batch) between two launches -- new deltas for one frame -- must be
    seen: the second launch gives that frame its NEW displacement, the others unchanged bits."""
    M, N, F = 256, 20_000, 8
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F)
    outs = [torch.empty_like(d_P) for _ in range(F)]
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    before = [o.clone() for o in outs]
    new_delta = (deltas[2] * np.float32(-0.5)).astype(np.float32)
    d_new = torch.from_numpy(new_delta).to(dev)
    single = capi.Batch([engines[2]])                      # the same context, rebuilt through a batch of one (in-place rest array)
    single.set_points_dev([keep[0].data_ptr()], [d_new.data_ptr()], M)
    single.build_async(); assert single.build_result()[0].terminationtype == 1
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    for f in range(F):
        if f != 2:
            assert torch.equal(outs[f], before[f]), f
    table = oracle.control_table(rest, (rest + new_delta).astype(np.float32))
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    assert parity_ratio(outs[2].cpu().numpy(), ref, P, TOL) <= 1.0
    assert not torch.equal(outs[2], before[2])
    single.close()
    _close(engines, batch)


def test_one_rest_rig_is_checked_by_content_not_by_address_alone(hip_lib, oracle):
    """the host decides "one rest rig" by the address the rest points were read from.  An address does not
    identify its contents: here the array is REWRITTEN between the set-ups of two contexts (same pointer, other rig).  The
    pack kernel compares the centres on the device: the frame built on other points is passed through (P_out = P_in, as
    for a failed build), the matching frames are evaluated, and the next call on the batch reports FD_E_INVALID."""
    M, N = 96, 6_000
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(100_000)[::16][:N].copy()
    rest = synth.control_points(M, "head")
    other = (rest * np.float32(1.01)).astype(np.float32)
    deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(3)]).astype(np.float32)
    d_P = torch.from_numpy(P).to(dev)
    d_rest = torch.from_numpy(rest).to(dev)
    d_del = torch.from_numpy(deltas).to(dev)
    engines = []
    for _ in range(3):
        e = capi.Engine(); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR); engines.append(e)
    pair = capi.Batch(engines[:2]); lone = capi.Batch([engines[2]]); batch = capi.Batch(engines)
    pair.set_points_dev([d_rest.data_ptr()] * 2, [d_del[k].data_ptr() for k in range(2)], M)
    pair.build_async(); assert [r.terminationtype for r in pair.build_result()] == [1, 1]
    d_rest.copy_(torch.from_numpy(other).to(dev))            # the same array, another rig
    torch.cuda.synchronize()
    lone.set_points_dev([d_rest.data_ptr()], [d_del[2].data_ptr()], M)
    lone.build_async(); assert lone.build_result()[0].terminationtype == 1
    outs = [torch.empty_like(d_P) for _ in range(3)]
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs])          # same address everywhere: accepted by the host
    torch.cuda.synchronize()
    for k in range(2):
        table = oracle.control_table(rest, (rest + deltas[k]).astype(np.float32))
        rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
        ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
        assert parity_ratio(outs[k].cpu().numpy(), ref, P, TOL) <= 1.0, k
    assert torch.equal(outs[2], d_P)                          # passed through, not evaluated with context 0's centres
    with pytest.raises(capi.FdError) as ei:
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs])
    assert ei.value.code == capi.FD_E_INVALID and "context 2" in str(ei.value)
    # a pipeline never polls between groups: it sets the next group's points first -- that call reports the word (once) instead of
    # clearing it unseen, and the one after starts clean
    d_rest.copy_(torch.from_numpy(rest).to(dev)); torch.cuda.synchronize()
    with pytest.raises(capi.FdError) as ei:
        batch.set_points_dev([d_rest.data_ptr()] * 3, [d_del[k].data_ptr() for k in range(3)], M)
    assert ei.value.code == capi.FD_E_INVALID and "previous group" in str(ei.value)
    batch.set_points_dev([d_rest.data_ptr()] * 3, [d_del[k].data_ptr() for k in range(3)], M)
    batch.build_async(); assert [r.terminationtype for r in batch.build_result()] == [1, 1, 1]
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs])
    torch.cuda.synchronize()
    assert not torch.equal(outs[2], d_P)
    for b in (pair, lone, batch):
        b.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("kind,okind,params", [(capi.KERNEL_THIN_PLATE, fo.KERNEL_THIN_PLATE, ()),
                                               (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN, (1.0, 5.0))])
@pytest.mark.parametrize("N,F", [(1, 16), (5, 32), (63, 32), (64, 16), (65, 32), (511, 16), (513, 4), (1025, 32), (130, 24), (33, 17)])
def test_shared_launch_at_the_edges_of_its_vertex_groups(hip_lib, oracle, kind, okind, params, N, F):
    """Vertex counts around the wave's 64 and the workgroup's 512, with the frame counts that take the branch-free
    epilogue (16, 32) and the general one (4): every vertex against the oracle, tangent frames and the gate on; and an
    empty mesh is a no-op."""
    M = 64
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(N * 131 + F)
    P = synth.head_mesh(4096)[rng.permutation(4096)[:N]].copy()
    rest = synth.control_points(M, "head")
    deltas = np.stack([synth.smooth_deltas(rest, f % 8) * np.float32(1.0 + 0.25 * (f // 8)) for f in range(F)]).astype(np.float32)
    d_P, d_rest, d_del = (torch.from_numpy(a).to(dev) for a in (P, rest, deltas))
    engines = []
    for _ in range(F):
        e = capi.Engine(); e.set_kernel(kind, params); e.set_term(capi.TERM_LINEAR); engines.append(e)
    batch = capi.Batch(engines)
    batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + f * M * 12 for f in range(F)], M)
    batch.build_async()
    assert [r.terminationtype for r in batch.build_result()] == [1] * F
    outs = [torch.full_like(d_P, 9.0) for _ in range(F)]
    falls = [torch.full((N,), 7.0, device=dev) for _ in range(F)]
    batch.deform_shared_dev(0, d_P.data_ptr(), [o.data_ptr() for o in outs])                 # N = 0: nothing happens
    torch.cuda.synchronize()
    assert all(bool((o == 9.0).all()) for o in outs)
    tu, tv, nn = synth.tangent_frames(P)
    d_t = [torch.from_numpy(a).to(dev) for a in (tu, tv, nn)]
    dist2 = (rng.random(N) * 0.5).astype(np.float32)
    dist2[::3] = 0.45                                                                         # beyond radius2: gated
    d_d2 = torch.from_numpy(dist2).to(dev)
    r2 = np.float32(0.36)
    for mode in ("plain", "all"):
        kw, okw = {}, {}
        if mode == "all":
            kw = dict(d_dist2=d_d2.data_ptr(), radius2=r2, falloffrate=1.5, d_tangents=[t.data_ptr() for t in d_t])
            okw = dict(dist2=dist2, radius2=r2, falloffrate=1.5, tangents=(tu, tv, nn))
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs], d_falloff=[f.data_ptr() for f in falls], **kw)
        torch.cuda.synchronize()
        for f in sorted(set([0, F // 2, F - 1])):
            table = oracle.control_table(rest, (rest + deltas[f]).astype(np.float32))
            rc, tt, W, radii = oracle.build(table, okind, list(params), fo.TERM_LINEAR)
            ref, _ = oracle.deform(table, okind, radii, W, P, **okw)
            plain = None
            if mode == "all":
                plain, _ = oracle.deform(table, okind, radii, W, P, **{k: v for k, v in okw.items() if k != "tangents"})
            assert parity_ratio(outs[f].cpu().numpy(), ref, P, TOL, scale_out=plain) <= 1.0, (mode, f)
    batch.close()
    for e in engines:
        e.close()


@pytest.mark.parametrize("F", [17, 20, 21, 23, 24, 27, 28, 31, 32])
def test_every_frame_count_of_the_32_row_launch_matches_the_oracle(hip_lib, oracle, F):
    """17..32 frames take the 32-row tiles (k_deform32_tps_shared_wide) with rows packed three per frame: two row tiles up
    to 20 frames, three above; frame slots come in fours and the slots beyond F repeat the last frame, so EVERY count takes
    the straight-line epilogue (full groups, fd_falloff everywhere) -- checked here for every frame, together with the
    general epilogue (ragged tail, gate), and that nothing beyond the F outputs is touched."""
    M, N = 256, 5 * 512 + 77
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F)
    outs = [torch.empty_like(d_P) for _ in range(F)]
    falls = [torch.full((N,), 7.0, device=dev) for _ in range(F)]
    dist2 = (np.random.default_rng(5).random(N) * 0.6).astype(np.float32)
    d_d2 = torch.from_numpy(dist2).to(dev)
    r2 = np.float32(0.49)
    refs = {}
    for gate in (False, True):
        kw = dict(d_dist2=d_d2.data_ptr(), radius2=r2, falloffrate=1.5) if gate else {}
        okw = dict(dist2=dist2, radius2=r2, falloffrate=1.5) if gate else {}
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs], d_falloff=[f.data_ptr() for f in falls], **kw)
        torch.cuda.synchronize()
        for f in range(F):
            if f not in refs:
                table = oracle.control_table(rest, (rest + deltas[f]).astype(np.float32))
                rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
                refs[f] = (table, W, radii)
            table, W, radii = refs[f]
            ref, ref_fall = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, **okw)
            out = outs[f].cpu().numpy()
            assert parity_ratio(out, ref, P, TOL) <= 1.0, (gate, f)
            gated = dist2 > r2 if gate else np.zeros(N, bool)
            assert np.array_equal(out[gated], P[gated])
            assert np.allclose(falls[f].cpu().numpy()[~gated], ref_fall[~gated], rtol=2e-6, atol=1e-7)
    _close(engines, batch)


def test_fall_off_arrays_that_are_not_16_byte_aligned_take_the_general_epilogue(hip_lib, oracle):
    """The straight-line epilogues write fd_falloff in 8- and 16-byte pieces; an array that starts on an odd float (a view
    into a larger buffer) must still come out right -- the launch falls back to the general epilogue for it."""
    M, N, F = 128, 3 * 512, 32
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(M, N, F)
    outs = [torch.empty_like(d_P) for _ in range(F)]
    big = torch.full((F, N + 8), 7.0, device=dev)
    falls = [big[f, 1:N + 1] for f in range(F)]            # 4 bytes past a 16-byte boundary
    assert all(f.data_ptr() % 16 == 4 for f in falls)
    batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs], d_falloff=[f.data_ptr() for f in falls])
    torch.cuda.synchronize()
    for f in (0, 13, 31):
        table = oracle.control_table(rest, (rest + deltas[f]).astype(np.float32))
        rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
        ref, ref_fall = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
        assert parity_ratio(outs[f].cpu().numpy(), ref, P, TOL) <= 1.0, f
        assert np.allclose(falls[f].cpu().numpy(), ref_fall, rtol=2e-6, atol=1e-7)
        assert float(big[f, 0]) == 7.0 and float(big[f, N + 1]) == 7.0      # nothing written outside the view
    _close(engines, batch)


@pytest.mark.parametrize("F,kind,params", [(20, capi.KERNEL_THIN_PLATE, ()), (32, capi.KERNEL_THIN_PLATE, ()), (12, capi.KERNEL_THIN_PLATE, ()),
                                          (24, capi.KERNEL_GAUSSIAN_QNN, (1.0, 5.0, 0.0))])
def test_the_workgroup_budget_changes_the_schedule_not_the_bits(hip_lib, F, kind, params):
    """fd_batch_set_eval_cus: fewer workgroups than CUs (CUs left to builds), one per CU, and more than CUs (oversubscribed:
    shorter shares, later workgroups start as CUs come free) walk the vertex groups in different orders and shares -- every
    vertex's arithmetic is the same, so the outputs are bit-identical."""
    N = 300_037
    dev, P, rest, deltas, d_P, keep, engines, batch = _setup(256, N, F, kind, params)
    ref = None
    for cus in (0, 97, 224, 448, 1500):
        batch.set_eval_cus(cus)
        outs = [torch.full_like(d_P, float("nan")) for _ in range(F)]
        falls = [torch.zeros(N, device=dev) for _ in range(F)]
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs], d_falloff=[f.data_ptr() for f in falls])
        torch.cuda.synchronize()
        got = torch.stack(outs)
        assert not torch.isnan(got).any().item(), cus
        if ref is None:
            ref = got
        else:
            assert torch.equal(got, ref), cus
    _close(engines, batch)
