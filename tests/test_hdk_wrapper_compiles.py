"""CPU: a compiler reads hdk/SOP_FaceDeformHip.cpp.  The wrapper -- operator registration, parm templates,
page-wise gather / scatter, the cook -- is compiled against tests/hdk_mock/ (a mock of the dozen HDK types it touches; the HDK
itself is not available) with warnings on, and the cook harness is linked against the product library.  Running the cook
needs a GPU: tests/test_gpu_hdk_wrapper.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "hdk_mock")


def _gxx():
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("g++ not available")
    return cxx


def test_the_hdk_wrapper_compiles_against_the_mock():
    r = subprocess.run([_gxx(), "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-I", MOCK,
                        "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "hdk", "SOP_FaceDeformHip.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "warning" not in r.stderr, r.stderr


def build_harness(out_path):
    """The harness (wrapper + mock + main) against facedeform_amd/lib/libfacedeform_hip.so."""
    from facedeform_amd import _build
    if not os.path.exists(_build.LIB_PATH):
        _build.build()
    lib_dir = os.path.dirname(_build.LIB_PATH)
    r = subprocess.run([_gxx(), "-std=c++17", "-O1", "-I", MOCK, "-I", os.path.join(ROOT, "include"),
                        os.path.join(MOCK, "cook_harness.cpp"), "-o", out_path, "-L", lib_dir, "-lfacedeform_hip",
                        "-Wl,-rpath," + lib_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out_path


def test_the_cook_harness_links_against_the_product_library(tmp_path):
    build_harness(str(tmp_path / "cook_harness"))
