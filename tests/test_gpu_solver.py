"""GPU: the two direct solvers of the dense system (include/facedeform_hip.h FD_SOLVER_*).

AUTO sends the conditionally positive definite (kernel, term) pairs through the null-space
Cholesky (facedeform_amd/csrc/fd_nullspace.hip) and everything else through the pivoted LU
(fd_build.hip); LU forces the latter.  Both solve the system of reference
src/SOP_FaceDeform.cpp:331-368 in north_star's dense form, so their weights must agree with each
other and with the fp64 oracle to rounding, and every behaviour at the boundary (termination
types, reports, fd_set_deltas, batches) must be the same whichever one ran."""
import numpy as np
import pytest
import torch

from conftest import parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu

SPD_CASES = [
    (capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR),
    (capi.KERNEL_THIN_PLATE, [1e-3], capi.TERM_LINEAR),         # smoothing on the diagonal
    (capi.KERNEL_CUBIC, [], capi.TERM_LINEAR),
    (capi.KERNEL_BIHARMONIC, [], capi.TERM_LINEAR),
    (capi.KERNEL_BIHARMONIC, [], capi.TERM_CONST),
    (capi.KERNEL_GAUSSIAN, [0.35], capi.TERM_LINEAR),
    (capi.KERNEL_GAUSSIAN, [0.35], capi.TERM_CONST),
    (capi.KERNEL_GAUSSIAN, [0.35, 1e-2], capi.TERM_ZERO),       # no projection at all: plain Cholesky
]


def _engine(kind, params, term, rest, delta, solver):
    e = capi.Engine(solver=solver)
    e.set_kernel(kind, params); e.set_term(term)
    e.set_points(rest, delta)
    return e


@pytest.mark.parametrize("M", [16, 37, 256, 300, 700])
@pytest.mark.parametrize("kind,params,term", SPD_CASES)
def test_cholesky_and_lu_agree_with_each_other_and_the_oracle(hip_lib, oracle, M, kind, params, term):
    """M = 16 is the smallest system AUTO projects; 37 and 300 leave ragged last blocks and pivot
    rows that straddle the 32-row tiles; 700 takes the ranged back-substitution."""
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 2)
    delta = (deform - rest).astype(np.float32)
    if kind == capi.KERNEL_GAUSSIAN and M > 300:
        params = [0.12] + list(params[1:])     # a fixed radius of 0.35 over 700 centres is numerically singular
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, kind, params, term)
    assert tt == 1
    got = {}
    for solver in (capi.SOLVER_AUTO, capi.SOLVER_LU):
        e = _engine(kind, params, term, rest, delta, solver)
        rep = e.build()
        T = (4, 1, 0)[term]
        assert rep.terminationtype == 1 and rep.n == M + T and rep.iterationscount == M + T
        assert 0.0 < rep.pivot_ratio <= 1.0
        got[solver], _ = e.get_weights()
        e.close()
    scale = np.abs(W).max()
    # fp64 direct solves of one system: cond * eps apart (cond up to ~1e5 for these rigs)
    assert np.abs(got[capi.SOLVER_AUTO] - got[capi.SOLVER_LU]).max() <= 2e-9 * scale
    for solver, Wg in got.items():
        assert np.abs(Wg - W).max() <= 2e-9 * scale, solver
    if kind != capi.KERNEL_GAUSSIAN or term != capi.TERM_ZERO or M > 16:
        assert not np.array_equal(got[capi.SOLVER_AUTO], got[capi.SOLVER_LU])   # two different eliminations really ran


def test_pairs_outside_the_definite_family_take_lu_whatever_is_asked(hip_lib):
    """QNN radii (non-symmetric Phi), thin-plate without the linear term, -r without any term,
    negative smoothing, tiny rigs: AUTO must not project them -- identical bits to LU."""
    rest = synth.control_points(200, "head")
    delta = synth.smooth_deltas(rest, 1).astype(np.float32)
    cases = [(capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0], capi.TERM_LINEAR, rest, delta),
             (capi.KERNEL_THIN_PLATE, [], capi.TERM_CONST, rest, delta),
             (capi.KERNEL_THIN_PLATE, [], capi.TERM_ZERO, rest, delta),
             (capi.KERNEL_CUBIC, [], capi.TERM_CONST, rest, delta),
             (capi.KERNEL_BIHARMONIC, [], capi.TERM_ZERO, rest, delta),
             (capi.KERNEL_THIN_PLATE, [-1e-4], capi.TERM_LINEAR, rest, delta),
             (capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR, rest[:12], delta[:12])]
    for kind, params, term, r, d in cases:
        W = []
        for solver in (capi.SOLVER_AUTO, capi.SOLVER_LU):
            e = _engine(kind, params, term, r, d, solver)
            assert e.build().terminationtype == 1
            W.append(e.get_weights()[0])
            e.close()
        assert np.array_equal(W[0], W[1]), (kind, term)


def test_failures_report_the_same_way(hip_lib):
    """-5 for coincident centres, -4 for a polynomial block without full column rank (all centres
    in one plane under the linear term) and for a poisoned matrix; never an exception or a hang."""
    rest = synth.control_points(64, "head")
    delta = synth.smooth_deltas(rest, 0).astype(np.float32)
    for solver in (capi.SOLVER_AUTO, capi.SOLVER_LU):
        dup = rest.copy(); dup[9] = dup[40]
        e = _engine(capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR, dup, delta, solver)
        assert e.build(check=False).terminationtype == -5
        flat = rest.copy(); flat[:, 2] = 0.25
        e.set_points(flat, delta)
        assert e.build(check=False).terminationtype == -4
        nan = rest.copy(); nan[3, 1] = np.nan
        e.set_points(nan, delta)
        assert e.build(check=False).terminationtype in (-4, -5)
        e.set_points(rest, delta)
        assert e.build().terminationtype == 1                      # and the context recovers
        e.close()


@pytest.mark.parametrize("M", [256, 2048])
def test_displacements_of_both_solvers_pass_parity(hip_lib, oracle, M):
    """End to end at the benchmark rig sizes: build with either solver, evaluate in fp32, compare
    with the oracle at 1e-5 relative per-vertex displacement."""
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 1)
    P = synth.head_mesh(1_000_000)[::250]
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
    ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    for solver in (capi.SOLVER_AUTO, capi.SOLVER_LU):
        e = _engine(capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR, rest, (deform - rest).astype(np.float32), solver)
        assert e.build().terminationtype == 1
        out, _ = e.deform(P)
        assert parity_ratio(out, ref, P, 1e-5) <= 1.0, solver
        e.close()


def test_resolve_and_batches_on_the_cholesky_path(hip_lib):
    """fd_set_deltas above the LU fast path's order limit (the Cholesky factor has none), and a
    batch of contexts against single builds: bit-identical, as on the LU path."""
    M = 2500
    rest = synth.control_points(M, "head")
    d = [synth.smooth_deltas(rest, f).astype(np.float32) for f in range(2)]
    fast, full = capi.Engine(), capi.Engine()
    for e in (fast, full):
        e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
    fast.set_points(rest, d[0]); fast.build()
    fast.set_deltas(d[1])
    assert fast.build().terminationtype == 1
    full.set_points(rest, d[1]); full.build()
    assert np.array_equal(fast.get_weights()[0], full.get_weights()[0])
    fast.close(); full.close()

    M = 700
    rest = synth.control_points(M, "head")
    dev = torch.device("cuda:0")
    d_rest = torch.from_numpy(rest).to(dev)
    es = [capi.Engine() for _ in range(5)]
    d_del = []
    for f, e in enumerate(es):
        e.set_kernel(capi.KERNEL_CUBIC); e.set_term(capi.TERM_LINEAR)
        d_del.append(torch.from_numpy(synth.smooth_deltas(rest, f).astype(np.float32)).to(dev))
    b = capi.Batch(es)
    b.set_points_dev([d_rest.data_ptr()] * 5, [t.data_ptr() for t in d_del], M)
    b.build_async(); reps = b.build_result()
    assert all(r.terminationtype == 1 for r in reps)
    for f, e in enumerate(es):
        single = capi.Engine()
        single.set_kernel(capi.KERNEL_CUBIC); single.set_term(capi.TERM_LINEAR)
        single.set_points(rest, synth.smooth_deltas(rest, f).astype(np.float32)); single.build()
        assert np.array_equal(single.get_weights()[0], e.get_weights()[0]), f
        single.close()
    b.close()
    for e in es:
        e.close()


def test_nothing_the_lu_accepts_fails_on_the_cholesky_path(hip_lib, oracle):
    """Two centres one fp32 step apart under the cubic kernel, and a fixed-radius Gaussian wider
    than the rig: the Cholesky pivot falls under the threshold (it is the square of what partial
    pivoting sees), the engine rebuilds with the LU and keeps it for the rig -- same weights as
    FD_SOLVER_LU, bit for bit, also through fd_set_deltas and in a batch."""
    M = 300
    rest = synth.control_points(M, "head")
    near = rest.copy(); near[17] = near[200] + np.float32(1e-7) * np.array([1, 0.5, -0.3], np.float32)
    cases = [(capi.KERNEL_CUBIC, [], capi.TERM_LINEAR, near), (capi.KERNEL_GAUSSIAN, [1.2], capi.TERM_ZERO, rest)]
    d0 = synth.smooth_deltas(rest, 0).astype(np.float32)
    d1 = synth.smooth_deltas(rest, 1).astype(np.float32)
    for kind, params, term, pts in cases:
        auto = _engine(kind, params, term, pts, d0, capi.SOLVER_AUTO)
        lu = _engine(kind, params, term, pts, d0, capi.SOLVER_LU)
        ra, rl = auto.build(), lu.build()
        assert ra.terminationtype == 1 and rl.terminationtype == 1
        assert np.array_equal(auto.get_weights()[0], lu.get_weights()[0]), kind
        auto.set_deltas(d1); lu.set_deltas(d1)                       # the stored factorisation is the LU's
        assert auto.build().terminationtype == 1 and lu.build().terminationtype == 1
        assert np.array_equal(auto.get_weights()[0], lu.get_weights()[0])
        auto.set_points(pts, d0)                                       # sticky: no second failed attempt
        assert auto.build().terminationtype == 1
        auto.set_term(capi.TERM_CONST if term != capi.TERM_CONST else capi.TERM_LINEAR)   # a new system: the choice is made afresh
        auto.set_term(term)
        assert auto.build().terminationtype == 1
        auto.close(); lu.close()
    # in a batch: the context that falls back is rebuilt alone; its neighbours keep their results
    dev = torch.device("cuda:0")
    es = [capi.Engine() for _ in range(3)]
    for e in es:
        e.set_kernel(capi.KERNEL_CUBIC); e.set_term(capi.TERM_LINEAR)
    pts = [rest, near, rest]
    d_pts = [torch.from_numpy(p).to(dev) for p in pts]
    d_del = [torch.from_numpy(synth.smooth_deltas(rest, f).astype(np.float32)).to(dev) for f in range(3)]
    b = capi.Batch(es)
    b.set_points_dev([t.data_ptr() for t in d_pts], [t.data_ptr() for t in d_del], M)
    b.build_async()
    assert all(r.terminationtype == 1 for r in b.build_result())
    for f, e in enumerate(es):
        ref = _engine(capi.KERNEL_CUBIC, [], capi.TERM_LINEAR, pts[f], synth.smooth_deltas(rest, f).astype(np.float32),
                      capi.SOLVER_LU if f == 1 else capi.SOLVER_AUTO)
        ref.build()
        assert np.array_equal(ref.get_weights()[0], e.get_weights()[0]), f
        ref.close()
    b.close()
    for e in es:
        e.close()


def test_every_block_shape_of_the_fused_step(hip_lib):
    """A sweep over M walks every relation between the 32-column blocks, the 192-row slabs of the
    step kernel, the 64-row waves inside them and the padding (one block only, one slab exactly,
    a slab plus one row, pivot rows straddling a block boundary ...), thin-plate + linear term and
    a Gaussian without projection; the LU is the reference."""
    sizes = list(range(16, 140, 5)) + [191, 192, 193, 196, 197, 223, 224, 225, 228, 229, 255, 256, 257, 260, 261, 287, 288, 289, 292, 293, 420, 452, 453]
    for kind, params, term in ((capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR), (capi.KERNEL_GAUSSIAN, [0.2, 1e-3], capi.TERM_ZERO)):
        for M in sizes:
            rest = synth.control_points(M, "head")
            delta = synth.smooth_deltas(rest, 1).astype(np.float32)
            W = []
            for solver in (capi.SOLVER_AUTO, capi.SOLVER_LU):
                e = _engine(kind, params, term, rest, delta, solver)
                rep = e.build()
                assert rep.terminationtype == 1, (kind, M, solver)
                W.append(e.get_weights()[0])
                e.close()
            assert np.abs(W[0] - W[1]).max() <= 1e-8 * np.abs(W[1]).max(), (kind, M)


def test_failed_async_build_is_repaired_or_reported_without_reading_the_result(hip_lib, oracle):
    """A pipeline that enqueues build + evaluations and never calls fd_build_result.
    Rig: two centres one fp32 step apart under the cubic kernel -- the Cholesky loses definiteness,
    the LU does not.  The evaluation enqueued right behind the build cannot know yet and passes the
    mesh through; once the build has executed, the next evaluation finds the posted status, rebuilds
    with the LU on the stream and is CORRECT.  With coincident centres (nothing can solve that) the
    next evaluation returns FD_E_DUPLICATE instead of passing the mesh through in silence."""
    M, N = 300, 20_000
    dev = torch.device("cuda:0")
    rest = synth.control_points(M, "head")
    near = rest.copy(); near[17] = near[200] + np.float32(1e-7) * np.array([1, 0.5, -0.3], np.float32)
    delta = synth.smooth_deltas(rest, 0).astype(np.float32)
    P = synth.head_mesh(N)
    d_P = torch.from_numpy(P).to(dev)
    out1, out2 = torch.empty_like(d_P), torch.empty_like(d_P)
    e = _engine(capi.KERNEL_CUBIC, [], capi.TERM_LINEAR, near, delta, capi.SOLVER_AUTO)
    e.build_async()
    e.deform_dev(N, d_P.data_ptr(), out1.data_ptr())          # enqueued before the status can be known
    torch.cuda.synchronize()                                    # (not an fd_* call: the engine is told nothing)
    e.deform_dev(N, d_P.data_ptr(), out2.data_ptr())          # finds the status, repairs, evaluates
    torch.cuda.synchronize()
    ref = _engine(capi.KERNEL_CUBIC, [], capi.TERM_LINEAR, near, delta, capi.SOLVER_LU)
    ref.build()
    want, _ = ref.deform(P)
    got = out2.cpu().numpy()
    assert not np.array_equal(got, P)                           # not a pass-through
    assert np.array_equal(got, want)                            # the LU's model, bit for bit
    assert e.build_result().terminationtype == 1
    e.close(); ref.close()
    # nothing solves coincident centres: reported, not passed through
    dup = rest.copy(); dup[9] = dup[2]
    e = _engine(capi.KERNEL_THIN_PLATE, [], capi.TERM_LINEAR, dup, delta, capi.SOLVER_AUTO)
    e.build_async()
    torch.cuda.synchronize()
    with pytest.raises(capi.FdError) as ei:
        e.deform_dev(N, d_P.data_ptr(), out2.data_ptr())
    assert ei.value.code == capi.FD_E_DUPLICATE
    with pytest.raises(capi.FdError):                            # sticky
        e.deform_dev(N, d_P.data_ptr(), out2.data_ptr())
    e.set_points(rest, delta)                                    # a new set-up clears it
    e.build_async()
    e.deform_dev(N, d_P.data_ptr(), out2.data_ptr())
    torch.cuda.synchronize()
    assert e.build_result().terminationtype == 1
    e.close()


def test_a_repaired_build_is_ordered_before_evaluations_on_other_streams(hip_lib):
    """the LU rebuild that the status poll enqueues runs on the CONTEXT's stream; the evaluation that
    found the failed status may launch on another one (bench.py: engines on a lane stream, evaluation on its own
    stream).  The rebuild's end is published as the context's wait event, so the other stream is ordered behind it:
    single launch (fd_deform_dev_stream) and the shared-rig launch of a batch alike give the LU's model, bit for bit.
    Rig: two centres one fp32 step apart under the cubic kernel (Cholesky fails, LU does not)."""
    M, N = 300, 50_000
    dev = torch.device("cuda:0")
    rest = synth.control_points(M, "head")
    near = rest.copy(); near[17] = near[200] + np.float32(1e-7) * np.array([1, 0.5, -0.3], np.float32)
    delta = synth.smooth_deltas(rest, 0).astype(np.float32)
    P = synth.head_mesh(N)
    d_P = torch.from_numpy(P).to(dev)
    ref = _engine(capi.KERNEL_CUBIC, [], capi.TERM_LINEAR, near, delta, capi.SOLVER_LU)
    ref.build()
    want, _ = ref.deform(P)
    ref.close()
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    # single context: build on stream A, evaluate on stream B
    e = _engine(capi.KERNEL_CUBIC, [], capi.TERM_LINEAR, near, delta, capi.SOLVER_AUTO)
    e.set_stream(sa.cuda_stream)
    out = torch.empty_like(d_P)
    e.build_async()
    torch.cuda.synchronize()                                    # the failed status is posted; the engine is told nothing
    e.deform_dev_stream(sb.cuda_stream, N, d_P.data_ptr(), out.data_ptr())      # polls, rebuilds on A, must wait for it on B
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    e.set_stream(None); e.close()
    # a batch of two on stream A, ONE evaluation call on stream B (the cubic kernel takes the per-frame launches of
    # fd_batch_deform_shared_dev; the statuses are polled, the models rebuilt on A, and B ordered behind both rebuilds)
    d_near = torch.from_numpy(near).to(dev)
    d_del = torch.from_numpy(np.stack([delta, (0.5 * delta).astype(np.float32)])).to(dev)
    wants = []
    for k in range(2):
        r = _engine(capi.KERNEL_CUBIC, [], capi.TERM_LINEAR, near, d_del[k].cpu().numpy(), capi.SOLVER_LU)
        r.build(); wants.append(r.deform(P)[0]); r.close()
    engines = []
    for k in range(2):
        x = capi.Engine(); x.set_stream(sa.cuda_stream); x.set_kernel(capi.KERNEL_CUBIC); x.set_term(capi.TERM_LINEAR)
        engines.append(x)
    b = capi.Batch(engines)
    b.set_points_dev([d_near.data_ptr()] * 2, [d_del[k].data_ptr() for k in range(2)], M)
    b.build_async(sa.cuda_stream)
    torch.cuda.synchronize()
    outs = [torch.empty_like(d_P) for _ in range(2)]
    b.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs], stream_ptr=sb.cuda_stream)
    torch.cuda.synchronize()
    assert [x.terminationtype for x in b.build_result()] == [1, 1]
    for k in range(2):
        assert np.array_equal(outs[k].cpu().numpy(), wants[k]), k
    b.close()
    for x in engines:
        x.set_stream(None); x.close()


def test_one_workgroup_build_matches_the_chain(hip_lib):
    """k_build_small (fd_config.solver = FD_SOLVER_ONE_WORKGROUP: everything after the assembly in one launch of one workgroup).
    Slower than the launch chain for a lone build and therefore not what FD_SOLVER_AUTO takes; kept selectable, so it is kept
    correct: weights against the chain's (FD_SOLVER_CHAIN) to rounding at M = 40, 256, 500, fd_set_deltas bit-identical to a
    rebuild.  (Round 2 selected it with an environment variable in a child process; the product library reads none.)"""
    res = {}
    for solver in (capi.SOLVER_CHAIN, capi.SOLVER_ONE_WORKGROUP):
        out = {}
        for M in (40, 256, 500):
            rest = synth.control_points(M, "head")
            d0 = synth.smooth_deltas(rest, 0).astype(np.float32); d1 = synth.smooth_deltas(rest, 1).astype(np.float32)
            e = capi.Engine(solver=solver); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
            e.set_points(rest, d0)
            rep = e.build()
            assert rep.terminationtype == 1 and rep.solver_used == solver, (M, rep.solver_used)
            out[f"w{M}"] = e.get_weights()[0]
            e.set_deltas(d1); assert e.build().terminationtype == 1
            wd = e.get_weights()[0]
            e.set_points(rest, d1); e.build()
            assert np.array_equal(wd, e.get_weights()[0]), M          # new deltas through the stored factor == rebuild
            out[f"d{M}"] = wd
            e.close()
        res[solver] = out
    for k in res[capi.SOLVER_CHAIN]:
        a, b = res[capi.SOLVER_CHAIN][k], res[capi.SOLVER_ONE_WORKGROUP][k]
        assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max(), k


def test_one_workgroup_solver_as_a_context_choice(hip_lib, oracle):
    """fd_config.solver = FD_SOLVER_ONE_WORKGROUP (what bench.py's frame pipeline builds with): per context, no
    environment switch.  Weights against the default solver's (rounding: the same system, another elimination order; the
    default solver's against the oracle: test_cholesky_and_lu_agree_with_each_other_and_the_oracle); a batch of such
    contexts equals the same models built one at a time bit for bit; a batch that mixes solvers takes the chain
    (FD_SOLVER_AUTO itself is the register-resident build at this size: tests/test_gpu_register_build.py)."""
    import torch
    M, F = 256, 6
    dev = torch.device("cuda", 0)
    rest = synth.control_points(M, "head")
    deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(F)]).astype(np.float32)
    d_rest = torch.from_numpy(rest).to(dev); d_del = torch.from_numpy(deltas).to(dev)
    torch.cuda.synchronize()

    def engines(solver, n):
        out = []
        for _ in range(n):
            e = capi.Engine(solver=solver); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR); out.append(e)
        return out

    ow = engines(capi.SOLVER_ONE_WORKGROUP, F)
    batch = capi.Batch(ow)
    batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + f * M * 12 for f in range(F)], M)
    batch.build_async()
    assert [r.terminationtype for r in batch.build_result()] == [1] * F
    Wb = [e.get_weights()[0] for e in ow]
    alone = engines(capi.SOLVER_ONE_WORKGROUP, 1)[0]
    auto = engines(capi.SOLVER_AUTO, 1)[0]
    for f in (0, 3, 5):
        alone.set_points(rest, deltas[f]); assert alone.build().terminationtype == 1
        assert np.array_equal(alone.get_weights()[0], Wb[f]), f                     # batch == single, same solver
        auto.set_points(rest, deltas[f]); assert auto.build().terminationtype == 1
        Wa = auto.get_weights()[0]
        assert np.abs(Wa - Wb[f]).max() <= 1e-9 * np.abs(Wa).max(), f                # same system, another elimination order
    mixed = capi.Batch([ow[0], auto])
    mixed.set_points_dev([d_rest.data_ptr()] * 2, [d_del.data_ptr(), d_del.data_ptr() + M * 12], M)
    mixed.build_async()
    assert [r.terminationtype for r in mixed.build_result()] == [1, 1]
    auto2 = engines(capi.SOLVER_CHAIN, 1)[0]
    auto2.set_points(rest, deltas[0]); auto2.build()
    assert np.array_equal(ow[0].get_weights()[0], auto2.get_weights()[0])               # the chain's bits
    for b in (mixed, batch):
        b.close()
    for e in ow + [alone, auto, auto2]:
        e.close()
