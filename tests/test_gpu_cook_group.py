"""GPU: the call path bench.py TIMES, held to the oracle (missing #1 / next #1).

Since round 3 bench.py enqueues every group of frames through ONE foreign call, fd_batch_cook_group
(csrc/fd_capi.hip): wait until the batch's packed models are consumed, set the points, build, pack on the build stream,
order the evaluation stream behind it, launch the shared-rig evaluation.  Round 3's parity tests went through the
five-call form on two streams; here the group call itself is cooked exactly as bench.py does and every output is
checked against the oracle:

  (a) the driver's form, `python bench.py --gpus 1 --steps 20 --warmup 5`: ONE group of 20 frames, the evaluation on
      the build stream itself (eval_stream = the build stream: no cross-stream event), fd_batch_set_eval_cus(256), the
      group's four timing events recorded inside the call;
  (b) the default pipeline: 3 lanes x 32 frames, one build stream per lane, ONE evaluation stream for all lanes, 224 CUs,
      five groups per lane with different delta phases from group to group (a stale or half-overwritten model of the
      lane's previous group would show), timing events on every fourth group as bench.py records them.

Outputs are NaN-filled first; every lane's LAST TWO groups write into buffers of their own and are compared with the
oracle on the seam sample of test_gpu_bench_launch.py (first / last unit of every round, the pool units, vertices on
control points, a few thousand spread over the mesh): conftest.l2_parity_ulp <= 1 and every vertex written.
Replaces, per frame, src/SOP_FaceDeform.cpp:331-368 (model) and :404-439 (loop body)."""
import os

import numpy as np
import pytest
import torch

from conftest import l2_parity, l2_parity_ulp
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo
from test_gpu_bench_launch import _sample_indices

pytestmark = pytest.mark.gpu
TOL = 1e-5
N, M = 1_000_000, 256
N_FRAMES = 64           # bench.py's delta phases


@pytest.fixture(scope="module")
def c2():
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    P[:8] = rest[:8]                                         # vertices on centres: d2 == 0
    deltas = np.stack([synth.rig_deltas(rest, f) for f in range(N_FRAMES)])      # the fp32 difference of the two rigs: the numbers the oracle's table holds
    return {"dev": dev, "P": P, "rest": rest, "deltas": deltas, "d_P": torch.from_numpy(P).to(dev),
            "d_rest": torch.from_numpy(rest).to(dev), "d_deltas": torch.from_numpy(deltas).to(dev)}


def _report(lines):
    root = os.environ.get("GRAFT_REPO_ROOT")
    if not root:
        return
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "cook_group_parity.txt"), "a") as fh:
        fh.write("\n".join(lines) + "\n")


def _raw_events():
    import bench                     # the benchmark's own event wrapper (raw hipEvent_t handles)
    return bench.RawEvents(capi)


def _check_frames(oracle, c2, frames, outs, falls, idx, tag, lines):
    """outs[k] / falls[k]: frame frames[k] of one group; every vertex written, the sample within the bar."""
    P, rest, deltas, dev = c2["P"], c2["rest"], c2["deltas"], c2["dev"]
    Ps = np.ascontiguousarray(P[idx])
    sel = torch.from_numpy(idx).to(dev)
    worst = 0.0
    for k, f in enumerate(frames):
        assert not torch.isnan(outs[k]).any().item(), (tag, f, "vertices the launch did not write")
        table = oracle.control_table(rest, (rest + deltas[f]).astype(np.float32))
        rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
        assert tt == 1
        ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, Ps)
        out = outs[k][sel].cpu().numpy()
        ulp, raw = l2_parity_ulp(out, ref, Ps, TOL), l2_parity(out, ref, Ps)
        lines.append(f"{tag} phase {f:2d}  l2_parity_ulp {ulp:.3f}  raw {raw:.3e}")
        worst = max(worst, ulp)
        assert ulp <= 1.0, (tag, f, ulp, raw)
        assert torch.all(falls[k] == 1.0).item(), (tag, f, "fd_falloff")
    return worst


def _engines(n, stream):
    engines = []
    for _ in range(n):
        e = capi.Engine(solver=capi.SOLVER_AUTO)        # bench.py's lane_solver (--build register)
        e.set_stream(stream.cuda_stream)
        e.set_kernel(capi.KERNEL_THIN_PLATE)
        e.set_term(capi.TERM_LINEAR)
        engines.append(e)
    return engines


def test_the_drivers_single_group_of_20_frames_matches_the_oracle(hip_lib, oracle, c2):
    """`bench.py --steps 20 --warmup 5`: lane 0 cooks 5, 32, 20, 20 frames untimed, then the timed group of 20 (phases 0..19)
    through fd_batch_cook_group with eval_stream == build stream on 256 CUs."""
    dev, d_P, d_rest, d_deltas = c2["dev"], c2["d_P"], c2["d_rest"], c2["d_deltas"]
    F, B = 20, 32
    stream = torch.cuda.Stream(device=dev)
    engines = _engines(B, stream)
    outs = [torch.empty_like(d_P) for _ in range(B)]
    falls = [torch.zeros(N, device=dev, dtype=torch.float32) for _ in range(B)]
    stride = M * 12
    batches = {}

    def cook(count, frames, events=None):
        if count not in batches:
            batches[count] = capi.Batch(engines[:count])
            batches[count].set_eval_cus(256)
        b = batches[count]
        tabs = b.group_tables([d_deltas.data_ptr() + f * stride for f in frames], [o.data_ptr() for o in outs[:count]],
                              [f.data_ptr() for f in falls[:count]])
        b.cook_group(stream.cuda_stream, stream.cuda_stream, d_rest.data_ptr(), M, N, d_P.data_ptr(), tabs, events=events)
        return b

    # the untimed part of the driver's run on this lane: warm-up of 5, one full group, the ragged batch primed twice
    cook(5, list(range(5)))
    cook(B, [(32 * 3 + k) % N_FRAMES for k in range(B)])
    cook(F, [(40 + k) % N_FRAMES for k in range(F)])         # other phases than the timed group's: stale models would show
    cook(F, [(41 + k) % N_FRAMES for k in range(F)])
    torch.cuda.synchronize()
    for o in outs:
        o.fill_(float("nan"))
    for f in falls:
        f.zero_()
    torch.cuda.synchronize()
    ev = _raw_events()
    frames = list(range(F))
    b = cook(F, frames, events=ev.struct)
    torch.cuda.synchronize()
    assert [r.terminationtype for r in b.build_result()] == [1] * F
    build_ms, eval_ms = ev.ms(0, 1), ev.ms(2, 3)
    assert 0.0 < build_ms < 50.0 and 0.0 < eval_ms < 50.0
    idx = _sample_indices(N, 256)
    lines = [f"# fd_batch_cook_group, the driver's form: one group of {F} frames, evaluation on the build stream, 256 CUs; "
             f"{idx.size} sampled vertices per frame; build {build_ms * 1e3:.0f} us, evaluation launch {eval_ms * 1e3:.0f} us by the call's own events"]
    worst = _check_frames(oracle, c2, frames, outs[:F], falls[:F], idx, "driver-form", lines)
    lines.append(f"# worst l2_parity_ulp, driver's form: {worst:.3f}")
    # the slots beyond the group's 20 frames stay untouched
    for o in outs[F:]:
        assert torch.isnan(o).all().item()
    _report(lines)
    ev.close()
    for bt in batches.values():
        bt.close()
    for e in engines:
        e.set_stream(None)
        e.close()


def test_the_default_pipeline_of_three_lanes_matches_the_oracle(hip_lib, oracle, c2):
    """bench.py's default: 3 lanes x 32 frames per group, a build stream per lane, one evaluation stream, 224 CUs,
    fd_batch_cook_group per group, every fourth group timed; 15 groups (5 per lane), the last two of every lane checked."""
    dev, d_P, d_rest, d_deltas = c2["dev"], c2["d_P"], c2["d_rest"], c2["d_deltas"]
    B, n_lanes, n_groups, cus = 32, 3, 15, 224
    eval_stream = torch.cuda.Stream(device=dev)
    stride = M * 12
    lanes = []
    for _ in range(n_lanes):
        stream = torch.cuda.Stream(device=dev)
        engines = _engines(B, stream)
        batch = capi.Batch(engines)
        batch.set_eval_cus(cus)
        # three output sets: the lane's earlier groups, its last group but one, its last group
        sets = [([torch.full_like(d_P, float("nan")) for _ in range(B)], [torch.zeros(N, device=dev, dtype=torch.float32) for _ in range(B)])
                for _ in range(3)]
        lanes.append({"stream": stream, "engines": engines, "batch": batch, "sets": sets, "groups": []})
    torch.cuda.synchronize()
    per_lane = n_groups // n_lanes
    timed = {}
    for g in range(n_groups):
        ln = lanes[g % n_lanes]
        nth = g // n_lanes                                    # the lane's nth group
        which = 0 if nth < per_lane - 2 else (1 if nth == per_lane - 2 else 2)
        outs, falls = ln["sets"][which]
        frames = [(g * B + k) % N_FRAMES for k in range(B)]
        tabs = ln["batch"].group_tables([d_deltas.data_ptr() + f * stride for f in frames], [o.data_ptr() for o in outs], [f.data_ptr() for f in falls])
        ev = None
        if g % 4 == 0:
            ev = timed[g] = _raw_events()
        ln["batch"].cook_group(ln["stream"].cuda_stream, eval_stream.cuda_stream, d_rest.data_ptr(), M, N, d_P.data_ptr(), tabs,
                               events=ev.struct if ev else None)
        ln["groups"].append((g, which, frames))
        ln["keep"] = ln.get("keep", []) + [tabs]              # the pointer tables outlive the enqueued work
    torch.cuda.synchronize()
    for ln in lanes:
        assert [r.terminationtype for r in ln["batch"].build_result()] == [1] * B
    idx = _sample_indices(N, cus)
    lines = [f"# fd_batch_cook_group, the default pipeline: {n_lanes} lanes x {B} frames, one evaluation stream, {cus} CUs, {n_groups} groups, "
             f"every 4th timed; the last two groups of every lane against the oracle on {idx.size} sampled vertices per frame"]
    for g, ev in timed.items():
        lines.append(f"# group {g:2d}: build {ev.ms(0, 1) * 1e3:.0f} us, evaluation launch {ev.ms(2, 3) * 1e3:.0f} us")
        ev.close()
    worst = 0.0
    for li, ln in enumerate(lanes):
        for g, which, frames in ln["groups"][-2:]:
            outs, falls = ln["sets"][which]
            worst = max(worst, _check_frames(oracle, c2, frames, outs, falls, idx, f"lane {li} group {g:2d}", lines))
    lines.append(f"# worst l2_parity_ulp, default pipeline: {worst:.3f}")
    _report(lines)
    for ln in lanes:
        ln["batch"].close()
        for e in ln["engines"]:
            e.set_stream(None)
            e.close()


def test_a_group_on_one_stream_with_a_rig_that_cannot_be_built_passes_its_frames_through_and_says_so(hip_lib, oracle, c2):
    """Round 4: with the evaluation on the build stream, fd_batch_cook_group records no event between build, packing and
    evaluation, and polls no status inside the call (the build's event is recorded behind the evaluation: a query would see the
    previous build's completion).  A build that fails -- coincident control points, -5 -- is then handled as the header documents
    for asynchronous builds: the group's evaluation passes every frame through (P_out = P_in, what the reference's cook leaves
    behind an error, src/SOP_FaceDeform.cpp:364-368), fd_batch_build_result reports -5 for every frame, and the NEXT group on the
    batch, built on the sound rig, is correct again (oracle parity) -- nothing sticks to the batch."""
    dev, d_P, d_deltas = c2["dev"], c2["d_P"], c2["d_deltas"]
    F = 20
    n = 200_000
    stream = torch.cuda.Stream(device=dev)
    engines = _engines(F, stream)
    batch = capi.Batch(engines)
    outs = [torch.full((n, 3), float("nan"), device=dev, dtype=torch.float32) for _ in range(F)]
    falls = [torch.zeros(n, device=dev, dtype=torch.float32) for _ in range(F)]
    bad = c2["rest"].copy(); bad[40] = bad[3]
    d_bad = torch.from_numpy(bad).to(dev)
    d_Pn = d_P[:n].contiguous()
    stride = M * 12
    tabs = batch.group_tables([d_deltas.data_ptr() + f * stride for f in range(F)], [o.data_ptr() for o in outs], [f.data_ptr() for f in falls])
    torch.cuda.synchronize()
    batch.cook_group(stream.cuda_stream, stream.cuda_stream, d_bad.data_ptr(), M, n, d_Pn.data_ptr(), tabs)       # enqueued: no error yet
    torch.cuda.synchronize()
    assert [r.terminationtype for r in batch.build_result(check=False)] == [-5] * F
    for o in outs:
        assert torch.equal(o, d_Pn)                                    # passed through, every vertex written
    # the next group, sound rig, same batch and stream
    for o in outs:
        o.fill_(float("nan"))
    torch.cuda.synchronize()
    batch.cook_group(stream.cuda_stream, stream.cuda_stream, c2["d_rest"].data_ptr(), M, n, d_Pn.data_ptr(), tabs)
    torch.cuda.synchronize()
    assert [r.terminationtype for r in batch.build_result()] == [1] * F
    idx = np.unique(np.concatenate([np.arange(0, 8), np.arange(0, n, 97), [n - 1]]))
    lines = []
    sub = {"P": c2["P"][:n], "rest": c2["rest"], "deltas": c2["deltas"], "dev": dev}
    worst = _check_frames(oracle, sub, list(range(F)), outs, falls, idx, "after-a-failed-group", lines)
    assert worst <= 1.0
    batch.close()
    for e in engines:
        e.set_stream(None)
        e.close()
