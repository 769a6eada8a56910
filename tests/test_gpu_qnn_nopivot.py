"""GPU: the QNN model (the SOP's default, src/SOP_FaceDeform.cpp:342-345) without the pivot search.

With q <= 1 partial pivoting never interchanges on the QNN kernel block (profiles/r02_qnn_pivot_stats.txt), so
FD_SOLVER_AUTO runs k_lu_panel_np: no search, the largest |multiplier| recorded.  Held here:
  * q = 1: the no-pivot path is the one that ran (fd_report.solver_used), its weights are BIT-IDENTICAL to the pivoted LU's
    (same arithmetic, same order, whenever that one would not interchange) and within 1e-8 of the oracle's;
  * q = 3: multipliers in the thousands (q = 2 stays below 4 on these rigs and is let through, 1e-12 of the oracle) -- the build must come back through the pivoted LU (solver_used == FD_SOLVER_LU), again
    with the oracle's weights; and the rig keeps the pivoted LU from then on;
  * fd_set_deltas through the stored no-pivot factorisation; batched builds; the asynchronous repair."""
import time

import numpy as np
import pytest
import torch

from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu


def _oracle_w(oracle, rest, deform, q, z, term):
    table = oracle.control_table(rest, deform)
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_GAUSSIAN_QNN, [q, z], term)
    return tt, np.asarray(W, np.float64)


def _engine(rest, delta, q, z, term, solver=capi.SOLVER_AUTO):
    e = capi.Engine(solver=solver)
    e.set_kernel(capi.KERNEL_GAUSSIAN_QNN, [q, z, 0.0]); e.set_term(term); e.set_points(rest, delta)
    return e


@pytest.mark.parametrize("M", [20, 64, 100, 256, 500, 1000])
@pytest.mark.parametrize("term", [capi.TERM_LINEAR, capi.TERM_ZERO])
def test_q1_runs_without_pivot_search_and_matches_the_pivoted_lu_bit_for_bit(hip_lib, oracle, M, term):
    rest = synth.control_points(M, "head"); deform = synth.deformed_rig(rest, 1)
    delta = (deform - rest).astype(np.float32)
    tt, W = _oracle_w(oracle, rest, deform, 1.0, 5.0, term)
    assert tt == 1
    e = _engine(rest, delta, 1.0, 5.0, term)
    rep = e.build()
    assert rep.terminationtype == 1 and rep.solver_used == capi.SOLVER_LU_NOPIVOT
    Wg, _ = e.get_weights()
    assert np.abs(Wg - W).max() <= 1e-8 * np.abs(W).max()
    lu = _engine(rest, delta, 1.0, 5.0, term, capi.SOLVER_LU)
    rep_lu = lu.build()
    assert rep_lu.solver_used == capi.SOLVER_LU
    Wl, _ = lu.get_weights()
    assert np.array_equal(Wg, Wl)                       # the same factors, the same substitution
    assert rep.pivot_ratio == rep_lu.pivot_ratio
    # new deltas through the stored factorisation (no interchanges to replay)
    delta2 = (synth.deformed_rig(rest, 2) - rest).astype(np.float32)
    e.set_deltas(delta2); e.build()
    lu.set_deltas(delta2); lu.build()
    assert np.array_equal(e.get_weights()[0], lu.get_weights()[0])
    _, W2 = _oracle_w(oracle, rest, (rest + delta2).astype(np.float32), 1.0, 5.0, term)
    assert np.abs(e.get_weights()[0] - W2).max() <= 1e-8 * np.abs(W2).max()
    e.close(); lu.close()


@pytest.mark.parametrize("M", [100, 256])
def test_q3_falls_back_to_the_pivoted_lu(hip_lib, oracle, M):
    rest = synth.control_points(M, "head"); deform = synth.deformed_rig(rest, 1)
    delta = (deform - rest).astype(np.float32)
    tt, W = _oracle_w(oracle, rest, deform, 3.0, 5.0, capi.TERM_LINEAR)
    assert tt == 1
    e = _engine(rest, delta, 3.0, 5.0, capi.TERM_LINEAR)
    rep = e.build()
    assert rep.terminationtype == 1 and rep.solver_used == capi.SOLVER_LU        # came back through k_lu_panel
    Wg, _ = e.get_weights()
    assert np.abs(Wg - W).max() <= 1e-8 * np.abs(W).max()
    lu = _engine(rest, delta, 3.0, 5.0, capi.TERM_LINEAR, capi.SOLVER_LU)
    lu.build()
    assert np.array_equal(Wg, lu.get_weights()[0])
    # the rig stays on the pivoted LU (no second failed attempt per cook)
    e.set_points(rest, delta)
    assert e.build().solver_used == capi.SOLVER_LU
    # ... until the kernel's parameters change
    e.set_kernel(capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0, 0.0]); e.set_points(rest, delta)
    assert e.build().solver_used == capi.SOLVER_LU_NOPIVOT
    e.close(); lu.close()


@pytest.mark.parametrize("M,q", [(64, 2.0), (256, 2.0), (64, 3.0), (100, 1.5), (256, 1.2)])
def test_between_the_two_whichever_path_runs_gives_the_oracles_weights(hip_lib, oracle, M, q):
    """Multipliers up to kMaxMultiplier = 4 are let through (bounded growth); beyond, the pivoted LU: either way 1e-8."""
    rest = synth.control_points(M, "head"); deform = synth.deformed_rig(rest, 3)
    delta = (deform - rest).astype(np.float32)
    tt, W = _oracle_w(oracle, rest, deform, q, 5.0, capi.TERM_LINEAR)
    e = _engine(rest, delta, q, 5.0, capi.TERM_LINEAR)
    rep = e.build(check=False) if tt != 1 else e.build()
    assert rep.terminationtype == tt
    if tt == 1:
        assert rep.solver_used in (capi.SOLVER_LU, capi.SOLVER_LU_NOPIVOT)
        assert np.abs(e.get_weights()[0] - W).max() <= 1e-8 * np.abs(W).max()
    e.close()


def test_batched_q1_and_the_asynchronous_repair_of_q3(hip_lib, oracle):
    M = 256
    rest = synth.control_points(M, "head")
    dev = torch.device("cuda", 0)
    deltas = [(synth.deformed_rig(rest, k) - rest).astype(np.float32) for k in range(4)]
    engines = [_engine(rest, d, 1.0, 5.0, capi.TERM_LINEAR) for d in deltas]
    batch = capi.Batch(engines)
    batch.build_async()
    reps = batch.build_result()
    assert [r.terminationtype for r in reps] == [1] * 4 and [r.solver_used for r in reps] == [capi.SOLVER_LU_NOPIVOT] * 4
    for e, d in zip(engines, deltas):
        single = _engine(rest, d, 1.0, 5.0, capi.TERM_LINEAR)
        single.build()
        assert np.array_equal(e.get_weights()[0], single.get_weights()[0])       # batched == single, bit for bit
        single.close()
    batch.close()
    for e in engines:
        e.close()
    # q = 3 in a pipeline that never collects the build's result: the first evaluation after the build has executed repairs it
    deform = (rest + deltas[1]).astype(np.float32)
    tt, W = _oracle_w(oracle, rest, deform, 3.0, 5.0, capi.TERM_LINEAR)
    table = oracle.control_table(rest, deform)
    rc, _, Wo, radii = oracle.build(table, fo.KERNEL_GAUSSIAN_QNN, [3.0, 5.0], fo.TERM_LINEAR)
    P = synth.head_mesh(20_000)
    ref, _ = oracle.deform(table, fo.KERNEL_GAUSSIAN_QNN, radii, Wo, P)
    e = _engine(rest, deltas[1], 3.0, 5.0, capi.TERM_LINEAR)
    e.set_eval_precision(capi.EVAL_FP64)        # (weights of +-200 for displacements of 0.05: this rig is beyond fp32 -- fd_report.cancellation)
    e.build_async()
    e.synchronize()
    time.sleep(0.01)
    d_P = torch.from_numpy(P).to(dev); d_out = torch.empty_like(d_P)
    e.deform_dev(P.shape[0], d_P.data_ptr(), d_out.data_ptr())
    e.synchronize()
    out = d_out.cpu().numpy()
    from conftest import parity_ratio
    assert parity_ratio(out, ref, P, 1e-5) <= 1.0
    assert e.build_result().solver_used == capi.SOLVER_LU
    e.close()
