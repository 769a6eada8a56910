"""GPU: randomised engine-vs-oracle sweep through the C ABI -- ragged M (panel widths, tile
padding, LDS chunking) and N, every kernel, every term, smoothing, fp32 and fp64 evaluation,
gate + fall-off -- on top of the fixed golden and BASELINE-size cases."""
import numpy as np
import pytest

from conftest import parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

pytestmark = pytest.mark.gpu

KINDS = [(capi.KERNEL_THIN_PLATE, fo.KERNEL_THIN_PLATE), (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN),
         (capi.KERNEL_BIHARMONIC, fo.KERNEL_BIHARMONIC), (capi.KERNEL_CUBIC, fo.KERNEL_CUBIC),
         (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN)]


def _cases():
    rng = np.random.default_rng(4242)
    out = []
    for i in range(20):
        M = int(rng.choice([5, 17, 33, 47, 49, 63, 65, 100, 129, 255, 257, 300, 385, 511, 530, 700]))
        N = int(rng.integers(1, 40_000))
        out.append((i, M, N, i % 5, int(rng.integers(3)), float(rng.choice([0.0, 0.0, 1e-3])), bool(i % 3 == 0)))
    return out


@pytest.mark.parametrize("i,M,N,k,term,lam,fp64", _cases())
def test_engine_matches_oracle(hip_lib, oracle, i, M, N, k, term, lam, fp64):
    rng = np.random.default_rng(i)
    kc, ko = KINDS[k]
    rest = synth.control_points(M, "head")
    rest = (rest + 0.1 / M ** (1 / 3) * rng.normal(size=rest.shape).astype(np.float32)).astype(np.float32)
    deform = synth.deformed_rig(rest, i % 8)
    params = {capi.KERNEL_GAUSSIAN: [1.2 / M ** (1 / 3), lam], capi.KERNEL_GAUSSIAN_QNN: [1.0, 5.0, lam]}.get(kc, [lam])
    P = synth.head_mesh(max(N, 50_000))[:: max(1, max(N, 50_000) // N)][:N].copy()
    N = P.shape[0]
    e = capi.Engine(precision=capi.EVAL_FP64 if fp64 else capi.EVAL_FP32)
    e.set_points(rest, (deform - rest).astype(np.float32)); e.set_kernel(kc, params); e.set_term(term)
    rep = e.build()
    table = oracle.control_table(rest, deform)
    rc, tt, W_ref, radii = oracle.build(table, ko, params, term)
    assert rc == 0 and tt == 1 and rep.terminationtype == 1
    W, _ = e.get_weights()
    assert np.abs(W - W_ref).max() <= 1e-7 * np.abs(W_ref).max(), np.abs(W - W_ref).max() / np.abs(W_ref).max()
    dist2 = (rng.random(N) * 0.5).astype(np.float32)
    out, fall = e.deform(P, dist2=dist2, radius2=0.36, falloffrate=1.5)
    ref, ref_fall = oracle.deform(table, ko, radii, W_ref, P, dist2=dist2, radius2=0.36, falloffrate=1.5)
    tol = 2e-7 if fp64 else (3e-5 if kc == capi.KERNEL_CUBIC else 1e-5)
    assert parity_ratio(out, ref, P, tol) <= 1.0, (M, N, k, term, parity_ratio(out, ref, P, tol))
    assert np.allclose(fall, ref_fall, rtol=2e-6, atol=1e-7)
    e.close()
