"""CPU: the oracle's dist2 producer (reference src/capture.cpp:46-99) against an independently
formulated numpy computation (tests/golden/make_golden_capture.py)."""
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def cap():
    return np.load(os.path.join(HERE, "golden", "capture_golden.npz"))


def test_distances_match_independent_formulation(oracle, cap):
    out = oracle.capture_dist2(cap["P"], cap["tris"], radius2=1e30, dofalloff=True)
    ref = cap["d2"]
    assert np.abs(out - ref).max() <= 1e-6 * max(1.0, ref.max()) and (out >= 0).all()
    assert np.all(out[:150] <= 1e-12)            # points placed on vertices, faces and edges


def test_capture_semantics(oracle, cap):
    P, tris, ref, mask = cap["P"], cap["tris"], cap["d2"], cap["mask"]
    r2 = np.float32(0.09)
    out = oracle.capture_dist2(P, tris, r2, True, mask)
    inside = mask.astype(bool)
    assert np.all(out[~inside] == 0.0)                                   # attribute default (capture.cpp:31)
    near = inside & (ref.astype(np.float32) < r2)
    far = inside & ~(ref.astype(np.float32) < r2)
    assert near.any() and far.any()
    assert np.all(out[far] == -1.0)                                      # nothing within the radius (:76,88)
    assert np.allclose(out[near], ref[near], rtol=1e-6, atol=1e-9)
    assert np.all(oracle.capture_dist2(P, tris, r2, False, mask) == 0.0)   # dofalloff off (:71-75)
    assert np.all(oracle.capture_dist2(P[:10], tris[:0], r2, True) == -1.0)   # no rig surface at all
