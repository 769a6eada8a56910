"""GPU: the register-resident one-launch build (csrc/fd_build_reg.hip; what FD_SOLVER_AUTO takes on the definite path up to
256 control points, and FD_SOLVER_REGISTER asks for by name): control table, kernel-matrix assembly, null-space projection,
blocked Cholesky with the diagonal blocks factorised in registers beside the trailing updates, both substitutions and the
packing by one workgroup per model -- in ONE launch where the batch leaves CUs to its builds (a pipeline), behind two short
launches over all CUs for the parallel third (assembly, Y = K V, the projection: round 4) where it has the device to itself.
Replaces alglib::rbfsetpoints + rbfbuildmodel, reference
src/SOP_FaceDeform.cpp:331-368, in the dense formulation.

Bars: weights against the oracle <= 1e-8 max|W| (observed 1e-11 .. 1e-15) and against the launch chain to rounding, for every
(kernel, term) the definite path takes and rig sizes around the tile edges; a batch equals the same models built alone bit
for bit; fd_set_deltas equals a rebuild bit for bit; coincident centres report -5, a rank-deficient polynomial block -4; the
displacement parity of a cook through it holds 1e-5."""
import numpy as np
import pytest
import torch

from conftest import parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu

CASES = [(capi.KERNEL_THIN_PLATE, fo.KERNEL_THIN_PLATE, [], 0), (capi.KERNEL_CUBIC, fo.KERNEL_CUBIC, [], 0),
         (capi.KERNEL_BIHARMONIC, fo.KERNEL_BIHARMONIC, [], 0), (capi.KERNEL_BIHARMONIC, fo.KERNEL_BIHARMONIC, [], 1),
         (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN, [0.35, 0.0], 0), (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN, [0.35, 0.0], 1),
         (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN, [0.35, 1e-3], 2)]


def _engine(kind, params, term, solver):
    e = capi.Engine(solver=solver)
    e.set_kernel(kind, params); e.set_term(term)
    return e


@pytest.mark.parametrize("M", [16, 17, 31, 32, 33, 48, 100, 129, 240, 252, 255, 256])
def test_weights_against_the_oracle_and_the_chain(hip_lib, oracle, M):
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 2)
    delta = (deform - rest).astype(np.float32)
    table = oracle.control_table(rest, deform)
    for kind, okind, params, term in CASES:
        rc, tt, W, radii = oracle.build(table, okind, params, term)
        assert tt == 1
        got = {}
        for name, solver in (("register", capi.SOLVER_REGISTER), ("auto", capi.SOLVER_AUTO), ("chain", capi.SOLVER_CHAIN)):
            e = _engine(kind, params, term, solver)
            e.set_points(rest, delta)
            rep = e.build()
            assert rep.terminationtype == 1 and rep.n == M + (4 if term == 0 else 1 if term == 1 else 0), (name, kind, term)
            got[name] = (e.get_weights()[0], rep.pivot_ratio)
            e.close()
        assert np.array_equal(got["register"][0], got["auto"][0]), (kind, term)              # AUTO is the register build here
        scale = np.abs(W).max()
        assert np.abs(got["register"][0] - W).max() <= 1e-8 * scale, (kind, term, np.abs(got["register"][0] - W).max() / scale)
        assert np.abs(got["register"][0] - got["chain"][0]).max() <= 1e-9 * scale, (kind, term)
        assert got["register"][1] == pytest.approx(got["chain"][1], rel=1e-6)              # the same pivots


def test_a_batch_equals_its_models_built_alone_and_new_deltas_equal_a_rebuild(hip_lib):
    M, F = 256, 7
    dev = torch.device("cuda", 0)
    rest = synth.control_points(M, "head")
    deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(F)]).astype(np.float32)
    d_rest = torch.from_numpy(rest).to(dev); d_del = torch.from_numpy(deltas).to(dev)
    engines = [_engine(capi.KERNEL_THIN_PLATE, [], 0, capi.SOLVER_AUTO) for _ in range(F)]
    batch = capi.Batch(engines)
    batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + f * M * 12 for f in range(F)], M)
    batch.build_async()
    assert [r.terminationtype for r in batch.build_result()] == [1] * F
    alone = _engine(capi.KERNEL_THIN_PLATE, [], 0, capi.SOLVER_AUTO)
    for f in range(F):
        alone.set_points(rest, deltas[f]); assert alone.build().terminationtype == 1
        assert np.array_equal(alone.get_weights()[0], engines[f].get_weights()[0]), f
    # fd_set_deltas: no factor is kept, the context builds again from its own copy of the rest points -- same bits as a full set-up
    alone.set_points(rest, deltas[0]); alone.build()
    alone.set_deltas(deltas[4]); assert alone.build().terminationtype == 1
    assert np.array_equal(alone.get_weights()[0], engines[4].get_weights()[0])
    engines[1].set_deltas(deltas[6]); assert engines[1].build().terminationtype == 1      # a context that was built in a batch, too
    assert np.array_equal(engines[1].get_weights()[0], engines[6].get_weights()[0])
    batch.close(); alone.close()
    for e in engines:
        e.close()


def test_failures_are_reported_as_everywhere_else(hip_lib):
    M = 64
    rest = synth.control_points(M, "head")
    delta = synth.smooth_deltas(rest, 1).astype(np.float32)
    dup = rest.copy(); dup[40] = dup[3]
    e = _engine(capi.KERNEL_THIN_PLATE, [], 0, capi.SOLVER_REGISTER)
    e.set_points(dup, delta)
    with pytest.raises(capi.FdError) as ei:
        e.build()
    assert ei.value.code == capi.FD_E_DUPLICATE
    flat = rest.copy(); flat[:, 2] = 0.25                       # every centre in one plane: [1 x y z] has rank 3
    e.set_points(flat, delta)
    with pytest.raises(capi.FdError) as ei:
        e.build()
    assert ei.value.code == capi.FD_E_SINGULAR
    e.set_points(rest, delta)                                   # and the context recovers
    assert e.build().terminationtype == 1
    e.close()


def test_a_cook_through_the_register_build_holds_the_displacement_parity(hip_lib, oracle):
    M, N = 256, 30_011
    P = synth.head_mesh(100_000)[::3][:N].copy()
    rest = synth.control_points(M, "head")
    deform = synth.deformed_rig(rest, 1)
    e = _engine(capi.KERNEL_THIN_PLATE, [], 0, capi.SOLVER_REGISTER)
    e.set_points(rest, (deform - rest).astype(np.float32))
    assert e.build().terminationtype == 1
    out, _ = e.deform(P)
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    assert parity_ratio(out, ref, P, 1e-5) <= 1.0
    e.close()


@pytest.mark.parametrize("M", [20, 100, 241, 256])
def test_the_parallel_front_end_and_the_one_workgroup_form_agree(hip_lib, oracle, M):
    """Round 4: a build that has the device to itself (single fd_build; a batch whose evaluations take every CU) assembles K, forms
    Y = K V and rotates to B in two short launches over all CUs (k_reg_front1 / k_reg_front2) and factorises in one workgroup per
    model; a batch that leaves CUs to its builds (fd_batch_set_eval_cus below the device's count: the pipeline) keeps the whole build
    in that workgroup.  Same model either way: weights to rounding, both against the oracle, for every (kernel, term) of the path
    and batches whose front-end launches take 1, 2 and 3 tiles per wave (1, 20 and 32 models)."""
    dev = torch.device("cuda", 0)
    rest = synth.control_points(M, "head")
    d_rest = torch.from_numpy(rest).to(dev)
    for F in (1, 20, 32):
        deforms = [synth.deformed_rig(rest, f % 5) for f in range(F)]
        deltas = np.stack([(d - rest).astype(np.float32) for d in deforms])
        d_del = torch.from_numpy(deltas).to(dev)
        for kind, okind, params, term in (CASES if F == 20 else CASES[:1]):
            got = {}
            for form, cus in (("front", 0), ("one-workgroup", 224)):
                engines = [_engine(kind, params, term, capi.SOLVER_AUTO) for _ in range(F)]
                batch = capi.Batch(engines)
                batch.set_eval_cus(cus)
                batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + f * M * 12 for f in range(F)], M)
                batch.build_async()
                reps = batch.build_result()
                assert [r.terminationtype for r in reps] == [1] * F, (form, kind, term)
                assert all(r.solver_used == capi.SOLVER_REGISTER for r in reps)
                got[form] = [e.get_weights()[0] for e in engines]
                batch.close()
                for e in engines:
                    e.close()
            for f in (0, F // 2, F - 1):
                rc, tt, W, radii = oracle.build(oracle.control_table(rest, deforms[f]), okind, params, term)
                scale = np.abs(W).max()
                assert np.abs(got["front"][f] - W).max() <= 1e-8 * scale, (F, kind, term, f)
                assert np.abs(got["front"][f] - got["one-workgroup"][f]).max() <= 1e-10 * scale, (F, kind, term, f)
