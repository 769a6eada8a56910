"""CPU: the C-ABI library loads and exports every symbol include/*.h declares;
host-side cook logic (parm surface, clamps, error texts) behaves like the
reference's.  No compute calls: those need a GPU (tests/test_gpu_*.py)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import HAVE_GPU, ROOT
from facedeform_amd import capi
from facedeform_amd.sop import FaceDeformSOP


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "facedeform_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fd(?:sop)?_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(hip_lib):
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in include/facedeform_hip.h but not exported"
    assert sorted(capi.EXPORTS) == names


def test_abi_version(hip_lib):
    assert hip_lib.fd_abi_version() == 9   # 9: fd_shared_kernel_name; 8: fd_report carries the fp32 estimate, fd_set_eval_precision, fd_fp32_holds; 7: fd_batch_cook_group + the solver names; 6: fd_batch_set_eval_cus; 2: fdsop_geo grew the morph-space inputs; 3: solver choice, multilayer kind, fd_model_centres; 4: fd_mesh_capture + the capture inputs of fdsop_geo; 5: fd_batch_wait_consumed


def test_struct_layouts_match_header():
    assert C.sizeof(capi.FdConfig) == 32
    assert C.sizeof(capi.FdReport) == 72 and capi.FdReport.fp32_error.offset == 32   # ABI 8: the fp32 estimate behind the ABI-7 fields
    assert capi.FdsopGeo.mesh_unchanged.offset == 20 * 8 + 4   # (the two trailing int flags of ABI 3 share a slot)
    assert capi.FdsopGeo.edge_offsets.offset == 21 * 8
    assert C.sizeof(capi.FdsopGeo) == 26 * 8   # 13 mesh/rig fields + 7 morph-space fields + the flags + 5 capture fields


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-device failure mode")
def test_create_fails_loudly_without_device(hip_lib):
    with pytest.raises(capi.FdError) as ei:
        capi.Engine()
    assert ei.value.code == capi.FD_E_NO_DEVICE
    assert "no CPU path" in ei.value.text


def test_null_handles_are_rejected(hip_lib):
    assert hip_lib.fd_set_term(None, 0) == capi.FD_E_INVALID
    assert hip_lib.fd_build_async(None) == capi.FD_E_INVALID
    assert hip_lib.fd_model_bytes(None) == 0
    hip_lib.fd_destroy(None)  # no-op


# reference src/SOP_FaceDeform.cpp:99-137 (token, default)
REFERENCE_PARMS = [("group", None), ("model", 0), ("term", 0), ("qcoef", 1.0), ("zcoef", 5.0),
                   ("radius", 1.0), ("maxedges", 4), ("layers", 4), ("lambda", 0.1), ("tangent", 0),
                   ("morphspace", 0), ("doclampweight", 0), ("weightrange", (0.0, 1.0)),
                   ("dofalloff", 0), ("falloffradius", 1.0), ("falloffrate", 1.0)]


def test_parm_surface_tokens_and_defaults(hip_lib):
    toks = FaceDeformSOP.parm_tokens()
    assert toks[:16] == [t for t, _ in REFERENCE_PARMS]          # same order, nothing renamed
    assert set(toks[16:]) == {"kernel", "smoothing", "precision", "device"}
    node = FaceDeformSOP()
    for tok, default in REFERENCE_PARMS:
        if default is None:
            continue
        if isinstance(default, tuple):
            assert (node.get(tok, 0), node.get(tok, 1)) == default
        else:
            assert node.get(tok) == pytest.approx(default)
    with pytest.raises(KeyError):
        node.set("nosuchparm", 1.0)


def _tiny_inputs():
    mesh = np.zeros((4, 3), np.float32)
    rest = np.eye(4, 3, dtype=np.float32)
    return mesh, rest


def test_mismatched_rigs_error_text(hip_lib):
    node = FaceDeformSOP()
    mesh, rest = _tiny_inputs()
    res = node.cook(mesh, rest, rest[:3])
    assert res.severity == capi.FDSOP_ERROR
    assert res.errors == ["Rest and deform geometry should match."]   # reference :232
    assert np.array_equal(res.P, mesh)                                 # output is still the copy of input 0


def test_clamps_match_reference(hip_lib):
    node = FaceDeformSOP()
    for tok, v in (("qcoef", 0.0), ("zcoef", -3.0), ("radius", 0.0), ("lambda", 0.0), ("layers", 0), ("maxedges", -2)):
        node.set(tok, v)
    mesh, rest = _tiny_inputs()
    node.cook(mesh, rest, rest)          # fails later without a GPU; the clamps are applied first
    assert node.effective("qcoef") == pytest.approx(np.float32(0.1))
    assert node.effective("zcoef") == pytest.approx(np.float32(0.1))
    assert node.effective("radius") == pytest.approx(np.float32(0.01))
    assert node.effective("lambda") == pytest.approx(np.float32(0.01))
    assert node.effective("layers") == 1 and node.effective("maxedges") == 1
    node.set("radius", 2.5)
    node.cook(mesh, rest, rest)
    assert node.effective("radius") == 2.5


def test_ordinal_parms_parse_like_atoi(hip_lib):
    node = FaceDeformSOP()
    node.set("model", "1")
    node.set("term", "2")
    assert node.get_int("model") == 1 and node.get_int("term") == 2
    node.set("term", "garbage")            # atoi -> 0 -> linear
    assert node.get_int("term") == 0


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-device failure mode")
def test_cook_without_gpu_reports_engine_error(hip_lib):
    node = FaceDeformSOP()
    mesh, rest = _tiny_inputs()
    res = node.cook(mesh, rest, rest + 0.1)
    assert res.severity == capi.FDSOP_ERROR
    assert any("GPU deformation engine" in e for e in res.errors)


def test_bench_refuses_to_run_without_a_gpu():
    """bench.py measures the HIP path or nothing: on a box without a GPU it must stop with a
    message, not fall back to the oracle (only its cpu_baseline leg may touch oracle/)."""
    import subprocess, sys, os, torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "no GPU visible" in (r.stderr + r.stdout)


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` starts its N ranks itself (a child torch.distributed.run, before any
    HIP call).  Asked for more GPUs than the node shows it must exit non-zero with a message and no
    result line -- never a one-GPU figure labelled N.  Runs here (0 GPUs) and on
    a one-GPU box alike."""
    import subprocess, sys, os, torch
    have = torch.cuda.device_count()
    if have >= 8:
        pytest.skip("an 8-GPU node: nothing to refuse")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FD_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(have + 1 if have else 2),
                        "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout


def test_bench_rank_count_must_match_gpus_flag():
    """Started by a launcher with a world size other than --gpus: refuse (before touching a device)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
