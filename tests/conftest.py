import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAVE_GPU = _have_gpu()


def pytest_collection_modifyitems(config, items):
    if HAVE_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import fd_oracle
    fd_oracle.build()
    return fd_oracle.Oracle()


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "rbf_golden.npz"))


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built in-tree if missing (needs hipcc)."""
    from facedeform_amd import _build, capi
    if not os.path.exists(_build.LIB_PATH):
        _build.build()
    return capi.load()


def case_kind_term(name: str):
    """golden case name -> (kind, term) ints shared by oracle and engine."""
    from oracle import fd_oracle as fo
    kinds = {"thin_plate": fo.KERNEL_THIN_PLATE, "gaussian_qnn": fo.KERNEL_GAUSSIAN_QNN,
             "gaussian": fo.KERNEL_GAUSSIAN, "biharmonic": fo.KERNEL_BIHARMONIC,
             "cubic": fo.KERNEL_CUBIC}
    terms = {"linear": 0, "const": 1, "zero": 2}
    for k in ("thin_plate", "gaussian_qnn", "gaussian", "biharmonic", "cubic"):
        if name.startswith(k + "_"):
            rest = name[len(k) + 1:]
            return kinds[k], terms[rest.split("_")[0]]
    raise KeyError(name)


def parity_ratio(out, ref_out, P, tol, scale_out=None):
    """Worst per-component |out - ref| / allowed, where allowed = tol * max(|d_ref|, floor) + 1 ulp(ref).

    scale_out: take the displacement magnitude from this output instead of ref_out.  Used when
    the tangent projection is on: it can shrink a displacement a hundredfold, and the bar is
    relative to the RBF displacement, not to what is left of it after projection.

    The reference adds the fp32 displacement to the fp32 position (src/SOP_FaceDeform.cpp:438),
    so two displacements that agree to `tol` may still round P + d to neighbouring floats; the
    one-ulp term covers exactly that and nothing more.  floor = 1e-5 * max_v |d_ref|.
    <= 1 passes."""
    out = np.asarray(out, np.float64)
    ref = np.asarray(ref_out, np.float64)
    d_ref = (ref if scale_out is None else np.asarray(scale_out, np.float64)) - np.asarray(P, np.float64)
    nr = np.linalg.norm(d_ref, axis=1)
    floor = 1e-5 * (nr.max() if nr.size else 0.0)
    allowed = tol * np.maximum(nr, floor)[:, None] + np.spacing(np.abs(np.asarray(ref_out, np.float32))).astype(np.float64)
    return float((np.abs(out - ref) / allowed).max()) if out.size else 0.0
