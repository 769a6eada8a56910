"""ctypes binding of include/facedeform_hip.h.

Plumbing only: it loads the in-tree libfacedeform_hip.so and forwards calls.
There is no fallback: a missing library raises, and fd_create without a
gfx950 device fails with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

FD_OK = 0
FD_E_INVALID, FD_E_NOMEM, FD_E_DEVICE, FD_E_SINGULAR, FD_E_DUPLICATE, FD_E_NOT_BUILT, FD_E_NO_DEVICE = (
    -1, -2, -3, -4, -5, -6, -7)
KERNEL_GAUSSIAN, KERNEL_GAUSSIAN_QNN, KERNEL_THIN_PLATE, KERNEL_BIHARMONIC, KERNEL_CUBIC, KERNEL_GAUSSIAN_ML = range(6)
TERM_LINEAR, TERM_CONST, TERM_ZERO = range(3)
EVAL_FP32, EVAL_FP64 = 0, 1
ABI_VERSION = 9            # FD_ABI_VERSION of include/facedeform_hip.h this binding's structures are laid out for
SOLVER_AUTO, SOLVER_LU, SOLVER_ONE_WORKGROUP, SOLVER_REGISTER, SOLVER_CHAIN = 0, 1, 2, 3, 4
SOLVER_LU_NOPIVOT = 5          # a value of fd_report.solver_used only
OUTPUT_POSITION, OUTPUT_DISPLACEMENT = 0, 1
FDSOP_OK, FDSOP_MESSAGE, FDSOP_WARNING, FDSOP_ERROR = range(4)

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


class FdConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int), ("device", C.c_int), ("eval_precision", C.c_int),
                ("eval_variant", C.c_int), ("solver", C.c_int), ("reserved", C.c_int * 3)]


class FdReport(C.Structure):
    _fields_ = [("terminationtype", C.c_int), ("iterationscount", C.c_int), ("n", C.c_int),
                ("solver_used", C.c_int), ("pivot_ratio", C.c_double), ("t_assemble_ms", C.c_float),
                ("t_solve_ms", C.c_float), ("fp32_error", C.c_double), ("cancellation", C.c_double),
                ("delta_min", C.c_double), ("delta_max", C.c_double), ("extent", C.c_double)]


class FdsopGeo(C.Structure):
    _fields_ = [("npoints", C.c_int64), ("P", _f32p), ("tangentu", _f32p), ("tangentv", _f32p),
                ("N", _f32p), ("dist2", _f32p), ("rest_npoints", C.c_int64),
                ("deform_npoints", C.c_int64), ("rest_P", _f32p), ("deform_P", _f32p),
                ("P_out", _f32p), ("fd_falloff", _f32p), ("Cd", _f32p),
                ("nshapes", C.c_int64), ("shapes_P", C.POINTER(C.c_void_p)), ("shapes_npoints", C.POINTER(C.c_int64)),
                ("rest", _f32p), ("rest_changed", C.c_int), ("blends_changed", C.c_int),
                ("weights", C.POINTER(C.c_double)), ("weights_count", C.POINTER(C.c_int64)),
                ("rig_rest_unchanged", C.c_int), ("mesh_unchanged", C.c_int),
                ("edge_offsets", C.POINTER(C.c_int64)), ("edge_neighbours", C.POINTER(C.c_int)),
                ("rig_ntris", C.c_int64), ("rig_tris", _f32p), ("dist2_out", _f32p)]


# every symbol include/facedeform_hip.h declares
EXPORTS = [
    "fd_create", "fd_destroy", "fd_last_error", "fd_abi_version", "fd_set_stream", "fd_set_eval_precision", "fd_set_output", "fd_fp32_holds", "fd_set_points",
    "fd_set_points_dev", "fd_set_deltas", "fd_set_deltas_dev", "fd_set_kernel", "fd_set_term", "fd_build", "fd_build_async",
    "fd_build_result", "fd_deform", "fd_deform_dev", "fd_deform_dev_stream", "fd_get_weights", "fd_model_centres", "fd_model_bytes",
    "fd_export_model", "fd_import_model", "fd_synchronize", "fd_host_alloc", "fd_host_free",
    "fd_mesh_set", "fd_mesh_size", "fd_deform_mesh", "fd_mesh_capture", "fd_mesh_get_dist2",
    "fd_capture_dist2", "fd_capture_dist2_dev", "fd_capture_islands", "fd_capture_islands_dev",
    "fd_morph_create", "fd_morph_destroy", "fd_morph_last_error", "fd_morph_init", "fd_morph_init_dev",
    "fd_morph_set_rest", "fd_morph_is_initialised", "fd_morph_is_computed", "fd_morph_shape_count", "fd_morph_last_init_ms",
    "fd_morph_compute_weights_dev", "fd_morph_displace_dev", "fd_morph_apply", "fd_morph_get_weights",
    "fd_morph_get_qr",
    "fd_batch_create", "fd_batch_destroy", "fd_batch_size", "fd_batch_last_error", "fd_batch_wait_consumed", "fd_batch_prepare_shared", "fd_batch_set_eval_cus", "fd_batch_cook_group", "fd_shared_kernel_name", "fd_batch_set_shared_factor", "fd_batch_last_build_shared_factor",
    "fd_batch_set_points_dev", "fd_batch_build_async", "fd_batch_build_result", "fd_batch_deform_dev",
    "fd_batch_deform_shared_dev",
    "fdsop_create", "fdsop_destroy", "fdsop_set_float", "fdsop_set_int", "fdsop_set_string",
    "fdsop_get_float", "fdsop_get_int", "fdsop_parm_count", "fdsop_parm_token", "fdsop_cook",
    "fdsop_messages", "fdsop_effective_float", "fdsop_engine",
]

_lib = None


def lib_path() -> str:
    return os.environ.get("FACEDEFORM_HIP_LIB", _build.LIB_PATH)


def load() -> C.CDLL:
    """Load the shared library (built by facedeform_amd._build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc). The engine has no CPU fallback.")
    L = C.CDLL(path)
    vp, i32, i64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
    L.fd_create.argtypes = [C.POINTER(FdConfig)]; L.fd_create.restype = vp
    L.fd_destroy.argtypes = [vp]; L.fd_destroy.restype = None
    L.fd_last_error.argtypes = [vp]; L.fd_last_error.restype = C.c_char_p
    L.fd_abi_version.argtypes = []; L.fd_abi_version.restype = i32
    # fd_report / fdsop_geo are written whole into the caller's memory and carry no size field: a binding laid out for another ABI
    # would be overwritten past its end -- refuse the library instead
    if L.fd_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} speaks ABI {L.fd_abi_version()}, this binding is laid out for {ABI_VERSION}: rebuild "
                           "(`python -c 'import __graft_entry__ as g; g.build()'`)")
    L.fd_set_stream.argtypes = [vp, vp]; L.fd_set_stream.restype = i32
    L.fd_set_eval_precision.argtypes = [vp, i32]; L.fd_set_eval_precision.restype = i32
    L.fd_set_output.argtypes = [vp, i32]; L.fd_set_output.restype = i32
    L.fd_fp32_holds.argtypes = [C.POINTER(FdReport), C.c_double]; L.fd_fp32_holds.restype = i32
    L.fd_set_points.argtypes = [vp, vp, vp, i32]; L.fd_set_points.restype = i32
    L.fd_set_points_dev.argtypes = [vp, vp, vp, i32]; L.fd_set_points_dev.restype = i32
    L.fd_set_deltas.argtypes = [vp, vp, i32]; L.fd_set_deltas.restype = i32
    L.fd_set_deltas_dev.argtypes = [vp, vp, i32]; L.fd_set_deltas_dev.restype = i32
    L.fd_set_kernel.argtypes = [vp, i32, _f64p, i32]; L.fd_set_kernel.restype = i32
    L.fd_set_term.argtypes = [vp, i32]; L.fd_set_term.restype = i32
    L.fd_build.argtypes = [vp, C.POINTER(FdReport)]; L.fd_build.restype = i32
    L.fd_build_async.argtypes = [vp]; L.fd_build_async.restype = i32
    L.fd_build_result.argtypes = [vp, C.POINTER(FdReport)]; L.fd_build_result.restype = i32
    L.fd_deform.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float]
    L.fd_deform.restype = i32
    L.fd_deform_dev.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float]
    L.fd_deform_dev.restype = i32
    L.fd_deform_dev_stream.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float]
    L.fd_deform_dev_stream.restype = i32
    L.fd_get_weights.argtypes = [vp, _f64p, _f64p]; L.fd_get_weights.restype = i32
    L.fd_model_centres.argtypes = [vp]; L.fd_model_centres.restype = i32
    L.fd_model_bytes.argtypes = [vp]; L.fd_model_bytes.restype = sz
    L.fd_export_model.argtypes = [vp, vp, sz, i32]; L.fd_export_model.restype = i32
    L.fd_import_model.argtypes = [vp, vp, sz, i32]; L.fd_import_model.restype = i32
    L.fd_synchronize.argtypes = [vp]; L.fd_synchronize.restype = i32
    L.fd_mesh_set.argtypes = [vp, i64, vp, vp, vp, vp, vp]; L.fd_mesh_set.restype = i32
    L.fd_mesh_size.argtypes = [vp]; L.fd_mesh_size.restype = i64
    L.fd_deform_mesh.argtypes = [vp, vp, vp, C.c_float, C.c_float]; L.fd_deform_mesh.restype = i32
    L.fd_mesh_capture.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp, C.c_float, i32, vp]; L.fd_mesh_capture.restype = i32
    L.fd_mesh_get_dist2.argtypes = [vp, vp]; L.fd_mesh_get_dist2.restype = i32
    L.fd_capture_dist2.argtypes = [vp, i64, vp, vp, i32, vp, C.c_float, i32, vp]; L.fd_capture_dist2.restype = i32
    L.fd_capture_dist2_dev.argtypes = [vp, i64, vp, vp, i32, vp, C.c_float, i32, vp]; L.fd_capture_dist2_dev.restype = i32
    L.fd_capture_islands.argtypes = [vp, i64, vp, vp, vp, i32, vp, i32, vp]; L.fd_capture_islands.restype = i32
    L.fd_capture_islands_dev.argtypes = [vp, i64, vp, vp, vp, i32, vp, i32, vp]; L.fd_capture_islands_dev.restype = i32
    L.fd_morph_create.argtypes = [C.POINTER(FdConfig)]; L.fd_morph_create.restype = vp
    L.fd_morph_destroy.argtypes = [vp]; L.fd_morph_destroy.restype = None
    L.fd_morph_last_error.argtypes = [vp]; L.fd_morph_last_error.restype = C.c_char_p
    L.fd_morph_init.argtypes = [vp, i64, i32, vp, C.POINTER(vp)]; L.fd_morph_init.restype = i32
    L.fd_morph_init_dev.argtypes = [vp, i64, i32, vp, C.POINTER(vp)]; L.fd_morph_init_dev.restype = i32
    for name in ("fd_morph_set_rest", "fd_morph_is_initialised", "fd_morph_is_computed", "fd_morph_shape_count"):
        getattr(L, name).argtypes = [vp]; getattr(L, name).restype = i32
    L.fd_morph_set_rest.argtypes = [vp, vp, i32]; L.fd_morph_set_rest.restype = i32
    L.fd_morph_last_init_ms.argtypes = [vp]; L.fd_morph_last_init_ms.restype = C.c_float
    L.fd_morph_compute_weights_dev.argtypes = [vp, vp, vp]; L.fd_morph_compute_weights_dev.restype = i32
    L.fd_morph_displace_dev.argtypes = [vp, vp, vp, i32, C.c_float, vp]; L.fd_morph_displace_dev.restype = i32
    L.fd_morph_apply.argtypes = [vp, vp, vp, i32, C.c_float, _f64p]; L.fd_morph_apply.restype = i32
    L.fd_morph_get_weights.argtypes = [vp, _f64p]; L.fd_morph_get_weights.restype = i32
    L.fd_morph_get_qr.argtypes = [vp, _f64p, _f64p]; L.fd_morph_get_qr.restype = i32
    L.fd_host_alloc.argtypes = [sz]; L.fd_host_alloc.restype = vp
    L.fd_host_free.argtypes = [vp]; L.fd_host_free.restype = None
    L.fd_batch_create.argtypes = [C.POINTER(vp), i32]; L.fd_batch_create.restype = vp
    L.fd_batch_destroy.argtypes = [vp]; L.fd_batch_destroy.restype = None
    L.fd_batch_size.argtypes = [vp]; L.fd_batch_size.restype = i32
    L.fd_batch_wait_consumed.argtypes = [vp, vp]; L.fd_batch_wait_consumed.restype = i32
    L.fd_batch_prepare_shared.argtypes = [vp, vp, vp, vp]; L.fd_batch_prepare_shared.restype = i32
    L.fd_batch_set_eval_cus.argtypes = [vp, i32]; L.fd_batch_set_eval_cus.restype = i32
    L.fd_shared_kernel_name.argtypes = [i32, i32, i32]; L.fd_shared_kernel_name.restype = C.c_char_p
    L.fd_batch_set_shared_factor.argtypes = [vp, i32]; L.fd_batch_set_shared_factor.restype = i32
    L.fd_batch_last_build_shared_factor.argtypes = [vp]; L.fd_batch_last_build_shared_factor.restype = i32
    L.fd_batch_cook_group.argtypes = [vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp]; L.fd_batch_cook_group.restype = i32
    L.fd_batch_last_error.argtypes = [vp]; L.fd_batch_last_error.restype = C.c_char_p
    L.fd_batch_set_points_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i32]
    L.fd_batch_set_points_dev.restype = i32
    L.fd_batch_build_async.argtypes = [vp, vp]; L.fd_batch_build_async.restype = i32
    L.fd_batch_build_result.argtypes = [vp, C.POINTER(FdReport)]; L.fd_batch_build_result.restype = i32
    pv = C.POINTER(vp)
    L.fd_batch_deform_dev.argtypes = [vp, vp, i64, pv, pv, pv, pv, pv, pv, pv, C.c_float, C.c_float]
    L.fd_batch_deform_dev.restype = i32
    L.fd_batch_deform_shared_dev.argtypes = [vp, vp, i64, vp, pv, vp, pv, vp, vp, vp, C.c_float, C.c_float]
    L.fd_batch_deform_shared_dev.restype = i32
    L.fdsop_create.argtypes = [C.POINTER(FdConfig)]; L.fdsop_create.restype = vp
    L.fdsop_destroy.argtypes = [vp]; L.fdsop_destroy.restype = None
    L.fdsop_set_float.argtypes = [vp, C.c_char_p, i32, C.c_double]; L.fdsop_set_float.restype = i32
    L.fdsop_set_int.argtypes = [vp, C.c_char_p, i32]; L.fdsop_set_int.restype = i32
    L.fdsop_set_string.argtypes = [vp, C.c_char_p, C.c_char_p]; L.fdsop_set_string.restype = i32
    L.fdsop_get_float.argtypes = [vp, C.c_char_p, i32, _f64p]; L.fdsop_get_float.restype = i32
    L.fdsop_get_int.argtypes = [vp, C.c_char_p, C.POINTER(i32)]; L.fdsop_get_int.restype = i32
    L.fdsop_parm_count.argtypes = []; L.fdsop_parm_count.restype = i32
    L.fdsop_parm_token.argtypes = [i32]; L.fdsop_parm_token.restype = C.c_char_p
    L.fdsop_cook.argtypes = [vp, C.POINTER(FdsopGeo)]; L.fdsop_cook.restype = i32
    L.fdsop_messages.argtypes = [vp]; L.fdsop_messages.restype = C.c_char_p
    L.fdsop_effective_float.argtypes = [vp, C.c_char_p, _f64p]; L.fdsop_effective_float.restype = i32
    L.fdsop_engine.argtypes = [vp]; L.fdsop_engine.restype = vp
    _lib = L
    return L


class FdError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"facedeform_hip error {code}: {text}")
        self.code = code
        self.text = text


def _np_ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Engine:
    """Thin object wrapper over fd_ctx.  Host arrays are numpy; device pointers are ints."""

    def __init__(self, device: int = -1, precision: int = EVAL_FP32, variant: int = 0, solver: int = 0, _borrowed=None):
        self.L = load()
        self._own = _borrowed is None
        if _borrowed is not None:
            self.ctx = _borrowed
            return
        cfg = FdConfig(C.sizeof(FdConfig), device, precision, variant, solver)
        self.ctx = self.L.fd_create(C.byref(cfg))
        if not self.ctx:
            raise FdError(FD_E_NO_DEVICE, self.L.fd_last_error(None).decode())
        self.M = 0

    def close(self):
        if getattr(self, "ctx", None) and self._own:
            self.L.fd_destroy(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != FD_OK:
            raise FdError(rc, self.L.fd_last_error(self.ctx).decode())

    def last_error(self) -> str:
        return self.L.fd_last_error(self.ctx).decode()

    def set_stream(self, stream_ptr: int | None):
        self._check(self.L.fd_set_stream(self.ctx, C.c_void_p(stream_ptr or 0)))

    def set_eval_precision(self, precision: int):
        """EVAL_FP32 / EVAL_FP64 for the evaluations from here on (a built model carries both records)."""
        self._check(self.L.fd_set_eval_precision(self.ctx, precision))

    def set_output(self, what: int):
        """fd_set_output: OUTPUT_POSITION (P + d f, the reference's write-back) or OUTPUT_DISPLACEMENT (d f alone)."""
        self._check(self.L.fd_set_output(self.ctx, int(what)))

    def fp32_holds(self, report: "FdReport", tol: float = 1e-5) -> bool:
        """fd_fp32_holds: is the fp32 evaluation of the reported model expected to hold tol of every vertex's displacement?"""
        return bool(self.L.fd_fp32_holds(C.byref(report), tol))

    def set_points(self, rest, delta):
        rest = np.ascontiguousarray(rest, np.float32).reshape(-1, 3)
        delta = np.ascontiguousarray(delta, np.float32).reshape(-1, 3)
        if rest.shape != delta.shape:
            raise ValueError("rest and delta must have the same shape")
        self.M = rest.shape[0]
        self._check(self.L.fd_set_points(self.ctx, _np_ptr(rest), _np_ptr(delta), self.M))

    def set_points_dev(self, d_rest: int, d_delta: int, M: int):
        self.M = M
        self._check(self.L.fd_set_points_dev(self.ctx, C.c_void_p(d_rest), C.c_void_p(d_delta), M))

    def set_deltas(self, delta):
        """New deltas for the rest points of the last build (reuses its factorisation)."""
        delta = np.ascontiguousarray(delta, np.float32).reshape(-1, 3)
        self._check(self.L.fd_set_deltas(self.ctx, _np_ptr(delta), delta.shape[0]))

    def set_deltas_dev(self, d_delta: int, M: int):
        self._check(self.L.fd_set_deltas_dev(self.ctx, C.c_void_p(d_delta), M))

    def set_kernel(self, kind: int, params=()):
        p = np.ascontiguousarray(np.asarray(params, np.float64).reshape(-1))
        ptr = p.ctypes.data_as(_f64p) if p.size else None
        self._check(self.L.fd_set_kernel(self.ctx, kind, ptr, p.size))

    def set_term(self, term: int):
        self._check(self.L.fd_set_term(self.ctx, term))

    def build(self, check: bool = True) -> FdReport:
        rep = FdReport()
        rc = self.L.fd_build(self.ctx, C.byref(rep))
        if check:
            self._check(rc)
        rep.rc = rc
        return rep

    def build_async(self):
        self._check(self.L.fd_build_async(self.ctx))

    def build_result(self, check: bool = True) -> FdReport:
        rep = FdReport()
        rc = self.L.fd_build_result(self.ctx, C.byref(rep))
        if check:
            self._check(rc)
        rep.rc = rc
        return rep

    def deform(self, P, dist2=None, tangents=None, radius2=1.0, falloffrate=1.0, want_falloff=True):
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        N = P.shape[0]
        out = P.copy()
        d2 = None if dist2 is None else np.ascontiguousarray(dist2, np.float32)
        fall = np.zeros(N, np.float32) if want_falloff else None
        tu = tv = nr = None
        if tangents is not None:
            tu, tv, nr = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in tangents)
        self._check(self.L.fd_deform(self.ctx, N, _np_ptr(out), _np_ptr(out), _np_ptr(d2), _np_ptr(fall),
                                     _np_ptr(tu), _np_ptr(tv), _np_ptr(nr), float(radius2),
                                     float(falloffrate)))
        return out, fall

    def deform_into(self, P_in, P_out, dist2=None, falloff=None, tangents=None, radius2=1.0, falloffrate=1.0):
        """fd_deform on the caller's own arrays, no copies in the binding (float32, C-contiguous;
        page-locked arrays from host_array() take the chunked overlapped path)."""
        for a in (P_in, P_out, dist2, falloff) + tuple(tangents or ()):
            if a is not None and (a.dtype != np.float32 or not a.flags.c_contiguous):
                raise ValueError("arrays must be C-contiguous float32")
        N = P_in.shape[0]
        tu, tv, nr = tangents if tangents is not None else (None, None, None)
        self._check(self.L.fd_deform(self.ctx, N, _np_ptr(P_in), _np_ptr(P_out), _np_ptr(dist2), _np_ptr(falloff),
                                     _np_ptr(tu), _np_ptr(tv), _np_ptr(nr), float(radius2),
                                     float(falloffrate)))

    def mesh_set(self, P, dist2=None, tangents=None):
        """Upload the cook-invariant mesh arrays once (fd_mesh_set)."""
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        d2 = None if dist2 is None else np.ascontiguousarray(dist2, np.float32)
        tu = tv = nr = None
        if tangents is not None:
            tu, tv, nr = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in tangents)
        self._check(self.L.fd_mesh_set(self.ctx, P.shape[0], _np_ptr(P), _np_ptr(d2), _np_ptr(tu), _np_ptr(tv), _np_ptr(nr)))

    def mesh_capture(self, offsets, neighbours, rig, max_edges, triangles, radius2, dofalloff=True, want=True):
        """ProximityCapture on the device-resident mesh (fd_mesh_capture); returns the dist2 array if wanted."""
        offsets = np.ascontiguousarray(offsets, np.int64)
        neighbours = np.ascontiguousarray(neighbours, np.int32)
        rig = np.ascontiguousarray(rig, np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
        out = np.empty(int(self.L.fd_mesh_size(self.ctx)), np.float32) if want else None
        self._check(self.L.fd_mesh_capture(self.ctx, _np_ptr(offsets), _np_ptr(neighbours), rig.shape[0], _np_ptr(rig),
                                           int(max_edges), tri.shape[0], _np_ptr(tri), float(radius2), int(bool(dofalloff)),
                                           _np_ptr(out)))
        return out

    def deform_mesh(self, P_out, falloff=None, radius2=1.0, falloffrate=1.0):
        """Evaluate the cached mesh into caller-owned arrays (page-locked ones are written in place)."""
        for a in (P_out, falloff):
            if a is not None and (a.dtype != np.float32 or not a.flags.c_contiguous):
                raise ValueError("arrays must be C-contiguous float32")
        self._check(self.L.fd_deform_mesh(self.ctx, _np_ptr(P_out), _np_ptr(falloff), float(radius2), float(falloffrate)))

    def capture_islands(self, P, offsets, neighbours, rig, max_edges):
        """ProximityCapture::findIslands as a byte mask (host arrays; CSR adjacency of the mesh's edges)."""
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        rig = np.ascontiguousarray(rig, np.float32).reshape(-1, 3)
        offsets = np.ascontiguousarray(offsets, np.int64)
        neighbours = np.ascontiguousarray(neighbours, np.int32)
        mask = np.zeros(P.shape[0], np.uint8)
        self._check(self.L.fd_capture_islands(self.ctx, P.shape[0], _np_ptr(P), _np_ptr(offsets), _np_ptr(neighbours),
                                              rig.shape[0], _np_ptr(rig), int(max_edges), _np_ptr(mask)))
        return mask

    def capture_dist2(self, P, triangles, radius2, dofalloff=True, mask=None):
        """ProximityCapture's per-point squared distance to the rig surface (host arrays)."""
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        out = np.empty(P.shape[0], np.float32)
        self._check(self.L.fd_capture_dist2(self.ctx, P.shape[0], _np_ptr(P), _np_ptr(m), tri.shape[0], _np_ptr(tri),
                                            float(radius2), int(bool(dofalloff)), _np_ptr(out)))
        return out

    def capture_dist2_dev(self, N: int, d_P: int, d_mask: int, T: int, d_tri: int, radius2, dofalloff, d_dist2: int):
        vp = C.c_void_p
        self._check(self.L.fd_capture_dist2_dev(self.ctx, N, vp(d_P), vp(d_mask or None), T, vp(d_tri or None),
                                                float(radius2), int(bool(dofalloff)), vp(d_dist2)))

    def deform_dev(self, N: int, d_P_in: int, d_P_out: int, d_dist2: int = 0, d_falloff: int = 0,
                   d_tu: int = 0, d_tv: int = 0, d_nrm: int = 0, radius2=1.0, falloffrate=1.0):
        vp = C.c_void_p
        self._check(self.L.fd_deform_dev(self.ctx, N, vp(d_P_in), vp(d_P_out), vp(d_dist2 or None),
                                         vp(d_falloff or None), vp(d_tu or None), vp(d_tv or None),
                                         vp(d_nrm or None), float(radius2), float(falloffrate)))

    def deform_dev_stream(self, stream_ptr: int, N: int, d_P_in: int, d_P_out: int, d_dist2: int = 0,
                          d_falloff: int = 0, d_tu: int = 0, d_tv: int = 0, d_nrm: int = 0, radius2=1.0,
                          falloffrate=1.0):
        vp = C.c_void_p
        self._check(self.L.fd_deform_dev_stream(self.ctx, vp(stream_ptr), N, vp(d_P_in), vp(d_P_out),
                                                vp(d_dist2 or None), vp(d_falloff or None), vp(d_tu or None),
                                                vp(d_tv or None), vp(d_nrm or None), float(radius2),
                                                float(falloffrate)))

    def get_weights(self):
        n = int(self.L.fd_model_centres(self.ctx))     # M, or M * layers for the multilayer model
        W = np.zeros((n + 4, 3), np.float64)
        radii = np.zeros(max(n, 1), np.float64)
        self._check(self.L.fd_get_weights(self.ctx, W.ctypes.data_as(_f64p), radii.ctypes.data_as(_f64p)))
        return W, radii[:n]

    def model_bytes(self) -> int:
        return int(self.L.fd_model_bytes(self.ctx))

    def export_model(self) -> np.ndarray:
        buf = np.zeros(self.model_bytes(), np.uint8)
        self._check(self.L.fd_export_model(self.ctx, _np_ptr(buf), buf.size, 0))
        return buf

    def import_model(self, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, np.uint8)
        self._check(self.L.fd_import_model(self.ctx, _np_ptr(blob), blob.size, 0))
        self.M = int(np.frombuffer(blob[:8].tobytes(), np.int32)[1])

    def export_model_dev(self, d_buf: int, capacity: int):
        self._check(self.L.fd_export_model(self.ctx, C.c_void_p(d_buf), capacity, 1))

    def import_model_dev(self, d_buf: int, nbytes: int, M: int):
        self._check(self.L.fd_import_model(self.ctx, C.c_void_p(d_buf), nbytes, 1))
        self.M = M

    def synchronize(self):
        self._check(self.L.fd_synchronize(self.ctx))


MAX_BATCH = 32


class FdGroupEvents(C.Structure):
    """fd_group_events: raw hipEvent_t handles (ints) a caller owns; NULL members are skipped."""
    _fields_ = [("before_build", C.c_void_p), ("after_build", C.c_void_p), ("before_eval", C.c_void_p), ("after_eval", C.c_void_p)]


class _PinnedBlock:
    def __init__(self, L, ptr):
        self.L, self.ptr = L, ptr

    def __del__(self):
        try:
            self.L.fd_host_free(self.ptr)
        except Exception:
            pass


def host_array(shape, dtype=np.float32):
    """numpy array over page-locked memory from fd_host_alloc (freed with the array)."""
    L = load()
    dtype = np.dtype(dtype)
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    nbytes = int(np.prod(shape)) * dtype.itemsize
    ptr = L.fd_host_alloc(max(nbytes, 1))
    if not ptr:
        raise FdError(FD_E_NOMEM, L.fd_last_error(None).decode())
    block = _PinnedBlock(L, ptr)
    buf = (C.c_char * max(nbytes, 1)).from_address(ptr)
    buf._fd_block = block                      # keeps the allocation alive as long as any view
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class Batch:
    """fd_batch: contexts with the same M / kernel / term built by one launch chain."""

    def __init__(self, engines):
        self.L = load()
        self.engines = list(engines)
        arr = (C.c_void_p * len(self.engines))(*[e.ctx for e in self.engines])
        self.h = self.L.fd_batch_create(arr, len(self.engines))
        if not self.h:
            raise FdError(FD_E_INVALID, self.L.fd_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.fd_batch_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != FD_OK:
            raise FdError(rc, self.L.fd_batch_last_error(self.h).decode())

    def __len__(self):
        return self.L.fd_batch_size(self.h)

    def set_points_dev(self, d_rest_ptrs, d_delta_ptrs, M: int):
        """Device pointers (ints), one pair per context; read in place by the next build."""
        n = len(self.engines)
        if len(d_rest_ptrs) != n or len(d_delta_ptrs) != n:
            raise ValueError("one pointer pair per context")
        r = (C.c_void_p * n)(*d_rest_ptrs)
        d = (C.c_void_p * n)(*d_delta_ptrs)
        self._check(self.L.fd_batch_set_points_dev(self.h, r, d, M))
        for e in self.engines:
            e.M = M

    def build_async(self, stream_ptr: int | None = None):
        self._check(self.L.fd_batch_build_async(self.h, C.c_void_p(stream_ptr or 0)))

    def deform_dev(self, N: int, d_P_in, d_P_out, d_dist2=None, d_falloff=None, d_tangents=None, radius2=1.0,
                   falloffrate=1.0, stream_ptr: int | None = None):
        """One launch for all contexts: lists of device pointers (ints), one per context."""
        n = len(self.engines)

        def tab(lst):
            if lst is None:
                return None
            if len(lst) != n:
                raise ValueError("one pointer per context")
            return (C.c_void_p * n)(*[p or None for p in lst])

        tu = tv = nr = None
        if d_tangents is not None:
            tu, tv, nr = (tab(t) for t in d_tangents)
        self._check(self.L.fd_batch_deform_dev(self.h, C.c_void_p(stream_ptr or 0), N, tab(d_P_in), tab(d_P_out),
                                               tab(d_dist2), tab(d_falloff), tu, tv, nr, float(radius2),
                                               float(falloffrate)))

    def deform_shared_dev(self, N: int, d_P_in: int, d_P_out, d_dist2: int = 0, d_falloff=None, d_tangents=None,
                          radius2=1.0, falloffrate=1.0, stream_ptr: int | None = None):
        """Frames of one mesh and one rest rig: ONE input mesh (device pointer), one output per context."""
        n = len(self.engines)
        if len(d_P_out) != n or (d_falloff is not None and len(d_falloff) != n):
            raise ValueError("one output pointer per context")
        vp = C.c_void_p
        outs = (vp * n)(*d_P_out)
        falls = None if d_falloff is None else (vp * n)(*[p or None for p in d_falloff])
        tu, tv, nr = d_tangents if d_tangents is not None else (0, 0, 0)
        self._check(self.L.fd_batch_deform_shared_dev(self.h, vp(stream_ptr or 0), N, vp(d_P_in), outs, vp(d_dist2 or None),
                                                      falls, vp(tu or None), vp(tv or None), vp(nr or None),
                                                      float(radius2), float(falloffrate)))

    def prepare_shared(self, d_P_out, d_falloff=None, stream_ptr=None):
        """fd_batch_prepare_shared: pack the current models for a shared-rig evaluation into the batch's scratch on
        `stream_ptr` (typically the build stream); deform_shared_dev with the same outputs then only evaluates."""
        n = len(self.engines)
        if len(d_P_out) != n or (d_falloff is not None and len(d_falloff) != n):
            raise ValueError("one output pointer per context")
        vp = C.c_void_p
        outs = (vp * n)(*d_P_out)
        falls = None if d_falloff is None else (vp * n)(*[p or None for p in d_falloff])
        self._check(self.L.fd_batch_prepare_shared(self.h, vp(stream_ptr or 0), outs, falls))

    def set_eval_cus(self, n_cus: int):
        """fd_batch_set_eval_cus: CUs this batch's shared-rig evaluation launches may occupy (0: all)."""
        self._check(self.L.fd_batch_set_eval_cus(self.h, int(n_cus)))

    def set_shared_factor(self, on: bool = True):
        """fd_batch_set_shared_factor: one factorisation per batched build where the contexts share the rest array."""
        self._check(self.L.fd_batch_set_shared_factor(self.h, 1 if on else 0))

    def last_build_shared_factor(self) -> bool:
        return bool(self.L.fd_batch_last_build_shared_factor(self.h))

    def group_tables(self, d_delta_ptrs, d_P_out, d_falloff=None):
        """The pointer tables of fd_batch_cook_group, built once and reused by a pipeline whose arrays do not move."""
        n = len(self.engines)
        vp = C.c_void_p
        return ((vp * n)(*d_delta_ptrs), (vp * n)(*d_P_out), None if d_falloff is None else (vp * n)(*d_falloff))

    def cook_group(self, build_stream: int, eval_stream: int, d_rest: int, M: int, N: int, d_P_in: int, tables, events=None):
        """fd_batch_cook_group: one group of frames of a shot -- set-up, builds, packing and the shared-rig evaluation --
        enqueued by ONE foreign call.  tables = group_tables(...); events: an FdGroupEvents of raw hipEvent_t handles or None."""
        deltas, outs, falls = tables
        vp = C.c_void_p
        self._check(self.L.fd_batch_cook_group(self.h, vp(build_stream or 0), vp(eval_stream or 0), vp(d_rest), deltas, M, N,
                                               vp(d_P_in), outs, falls, C.byref(events) if events is not None else None))
        for e in self.engines:
            e.M = M

    def cook_group_call(self, build_stream: int, eval_stream: int, d_rest: int, M: int, N: int, d_P_in: int, tables, events=None):
        """The same call as cook_group with its arguments marshalled ONCE: returns a function of no arguments that enqueues the
        group (a pipeline whose arrays do not move calls it per group; ctypes conversions are a tenth of a short group's time)."""
        import functools
        deltas, outs, falls = tables
        vp = C.c_void_p
        fn = functools.partial(self.L.fd_batch_cook_group, self.h, vp(build_stream or 0), vp(eval_stream or 0), vp(d_rest), deltas, M, N,
                               vp(d_P_in), outs, falls, C.byref(events) if events is not None else None)
        for e in self.engines:
            e.M = M

        def call():
            rc = fn()
            if rc != FD_OK:
                self._check(rc)
        call.keep = (tables, events)          # (the tables and the event struct outlive the enqueued work)
        return call

    def wait_consumed(self, stream_ptr=None):
        """fd_batch_wait_consumed: `stream_ptr` waits until the last shared-rig evaluation has its own copy of
        the models (its pack kernel has run) -- the contexts may then be rebuilt while that evaluation runs."""
        self._check(self.L.fd_batch_wait_consumed(self.h, C.c_void_p(stream_ptr or 0)))

    def build_result(self, check: bool = True):
        n = len(self.engines)
        reps = (FdReport * n)()
        rc = self.L.fd_batch_build_result(self.h, reps)
        if check:
            self._check(rc)
        return list(reps)


class Morph:
    """fd_morph: DirectBSEdit on the device (shapes matrix + packed Householder QR, pseudo-weights,
    displacement).  Host arrays are numpy; *_dev take device pointers (ints)."""

    def __init__(self, device: int = -1):
        self.L = load()
        cfg = FdConfig(C.sizeof(FdConfig), device, EVAL_FP32, 0)
        self.h = self.L.fd_morph_create(C.byref(cfg))
        if not self.h:
            raise FdError(FD_E_NO_DEVICE, self.L.fd_morph_last_error(None).decode())
        self.N = self.S = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.fd_morph_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != FD_OK:
            raise FdError(rc, self.L.fd_morph_last_error(self.h).decode())

    def init(self, rest, shapes):
        rest = np.ascontiguousarray(rest, np.float32).reshape(-1, 3)
        shapes = [np.ascontiguousarray(s, np.float32).reshape(-1, 3) for s in shapes]
        for s in shapes:
            if s.shape != rest.shape:
                raise ValueError("every shape must have the rest pose's point count")
        arr = (C.c_void_p * max(1, len(shapes)))(*[s.ctypes.data for s in shapes])
        self._check(self.L.fd_morph_init(self.h, rest.shape[0], len(shapes), _np_ptr(rest), arr))
        self.N, self.S = rest.shape[0], len(shapes)

    def init_dev(self, N: int, d_rest: int, d_shapes):
        arr = (C.c_void_p * max(1, len(d_shapes)))(*d_shapes)
        self._check(self.L.fd_morph_init_dev(self.h, N, len(d_shapes), C.c_void_p(d_rest), arr))
        self.N, self.S = N, len(d_shapes)

    def set_rest(self, rest):
        rest = None if rest is None else np.ascontiguousarray(rest, np.float32).reshape(-1, 3)
        self._check(self.L.fd_morph_set_rest(self.h, _np_ptr(rest), 0))

    @property
    def initialised(self):
        return bool(self.L.fd_morph_is_initialised(self.h))

    @property
    def computed(self):
        return bool(self.L.fd_morph_is_computed(self.h))

    @property
    def last_init_ms(self):
        return float(self.L.fd_morph_last_init_ms(self.h))

    def compute_weights_dev(self, d_P: int, stream_ptr: int | None = None):
        self._check(self.L.fd_morph_compute_weights_dev(self.h, C.c_void_p(d_P), C.c_void_p(stream_ptr or 0)))

    def displace_dev(self, d_P: int, clamp=None, add_delta=False, falloffradius=0.0, stream_ptr: int | None = None):
        cl = None if clamp is None else np.asarray(clamp, np.float32)
        self._check(self.L.fd_morph_displace_dev(self.h, C.c_void_p(d_P), _np_ptr(cl), int(bool(add_delta)),
                                                 float(falloffradius), C.c_void_p(stream_ptr or 0)))

    def apply(self, P, clamp=None, add_delta=False, falloffradius=0.0):
        """Weights + displacement on a host array; returns (P_out, w)."""
        out = np.array(P, np.float32, copy=True).reshape(-1, 3)
        cl = None if clamp is None else np.asarray(clamp, np.float32)
        w = np.zeros(max(1, self.S), np.float64)
        self._check(self.L.fd_morph_apply(self.h, _np_ptr(out), _np_ptr(cl), int(bool(add_delta)), float(falloffradius),
                                          w.ctypes.data_as(_f64p)))
        return out, w[: self.S]

    def weights(self):
        w = np.zeros(max(1, self.S), np.float64)
        self._check(self.L.fd_morph_get_weights(self.h, w.ctypes.data_as(_f64p)))
        return w[: self.S]

    def qr(self):
        QR = np.empty((3 * self.N, self.S), np.float64, order="F")
        tau = np.zeros(max(1, self.S), np.float64)
        self._check(self.L.fd_morph_get_qr(self.h, QR.ctypes.data_as(_f64p), tau.ctypes.data_as(_f64p)))
        return QR, tau[: self.S]

