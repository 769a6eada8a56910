"""Build libfacedeform_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
CSRC = os.path.join(_PKG, "csrc")
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libfacedeform_hip.so")
SOURCES = ["fd_eval.hip", "fd_eval_shared.hip", "fd_build.hip", "fd_nullspace.hip", "fd_build_reg.hip", "fd_capi.hip", "fd_morph.hip", "fd_capture.hip", "fd_sop_host.cpp"]
# per-file extras: keep the bf16 MFMA results of the evaluation kernel in VGPRs (the default puts
# them in AGPRs and pays one v_accvgpr_read per value)
EXTRA_FLAGS = {"fd_eval.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # no packed fp32 in the shared-rig kernel at all (the SLP vectoriser pairs scalar multiplies and adds into
               # v_pk_* again): packed arithmetic under its matrix instructions is what gave launch-to-launch differences
               # (DESIGN.md 4.1c); same speed, and the Gaussian instantiation loses its four spills
               "fd_eval_shared.hip": ["-fno-slp-vectorize"]}
HEADERS = [os.path.join(CSRC, "fd_internal.h"), os.path.join(CSRC, "fd_tuning.h"), os.path.join(CSRC, "fd_pack.h"), os.path.join(CSRC, "fd_eval_common.h"), os.path.join(_ROOT, "include", "facedeform_hip.h")]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(_PKG, "build")
    os.makedirs(obj_dir, exist_ok=True)
    cc = hipcc_path()
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(_ROOT, "include")]
    common += os.environ.get("FD_EXTRA_HIPCC_FLAGS", "").split()       # experiments (-DFD_...): python -m facedeform_amd._build
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmd = [cc] + common + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    tmp = LIB_PATH + ".tmp"
    link = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    subprocess.check_call(link)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
