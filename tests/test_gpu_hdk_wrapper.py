"""GPU: the HDK-side wrapper cooks on a PAGED mock detail.  hdk/SOP_FaceDeformHip.cpp is
compiled against tests/hdk_mock/ and run by tests/hdk_mock/cook_harness.cpp: point offsets that are not point indices (a
hole every 97 offsets), attribute pages of 1024 reached through page handles, a triangle strip for the capture's edge
adjacency, the rest / animated rigs on inputs 1 and 2.  Two cooks as Houdini would issue them: everything new, then only
input 2 changed (device-resident mesh + fd_set_deltas).  The harness itself checks the operator registration
(src/SOP_FaceDeform.cpp:38-45), that holes are never written, that fd_falloff and a white Cd exist (:386-388, :401);
here both outputs are held to the oracle at 1e-5."""
import os
import subprocess

import numpy as np
import pytest

from conftest import parity_ratio
from facedeform_amd import synth
from oracle import fd_oracle as fo
from test_hdk_wrapper_compiles import build_harness

pytestmark = pytest.mark.gpu


def test_the_wrapper_cooks_a_paged_detail(hip_lib, oracle, tmp_path):
    N, M = 5_003, 96
    P = synth.head_mesh(100_000)[::19][:N].copy()
    rest = synth.control_points(M, "head")
    dea = (rest + synth.smooth_deltas(rest, 1)).astype(np.float32)
    deb = (rest + synth.smooth_deltas(rest, 4)).astype(np.float32)
    exe = build_harness(str(tmp_path / "cook_harness"))
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as fh:
        np.array([N, M, 97], np.int64).tofile(fh)
        for a in (P, rest, dea, deb):
            np.ascontiguousarray(a, np.float32).tofile(fh)
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "parms: 20" in r.stdout                       # group + the reference's 15 + this repository's 4 additions
    assert "error:" not in r.stdout
    raw = np.fromfile(fout, np.float32)
    out_a, out_b, fall = raw[:3 * N].reshape(N, 3), raw[3 * N:6 * N].reshape(N, 3), raw[6 * N:]
    assert fall.shape == (N,) and np.all(fall == 1.0)    # no rig surface, dofalloff off: every distance 0, fall-off 1
    for out, deform in ((out_a, dea), (out_b, deb)):
        table = oracle.control_table(rest, deform)
        rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
        ref, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
        assert parity_ratio(out, ref, P, 1e-5) <= 1.0
    assert not np.array_equal(out_a, out_b)
