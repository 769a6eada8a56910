"""ctypes loader for the CPU oracle (oracle/fd_oracle.c).

TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED against the reference (see
fd_oracle.h).  Importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only; the product package facedeform_amd never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT_SO = os.path.join(_HERE, "libfd_oracle.so")

KERNEL_GAUSSIAN, KERNEL_GAUSSIAN_QNN, KERNEL_THIN_PLATE, KERNEL_BIHARMONIC, KERNEL_CUBIC, KERNEL_GAUSSIAN_ML = range(6)
TERM_LINEAR, TERM_CONST, TERM_ZERO = range(3)

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    """Compile oracle/libfd_oracle.so with the committed Makefile (gcc)."""
    if force or not os.path.exists(_DEFAULT_SO) or (
        os.path.getmtime(_DEFAULT_SO) < os.path.getmtime(os.path.join(_HERE, "fd_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libfd_oracle.so"])
    return _DEFAULT_SO


def build_native(out_path: str) -> str:
    """-O3 -march=native build for timing the CPU baseline on the current host."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "native", f"OUT={out_path}"])
    return out_path


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


class Oracle:
    def __init__(self, so_path: str | None = None):
        path = so_path or _DEFAULT_SO
        if not os.path.exists(path):
            build()
        self.lib = L = C.CDLL(path)
        L.fdo_control_table.argtypes = [_f32p, _f32p, C.c_int, _f64p]
        L.fdo_control_table.restype = None
        L.fdo_radii.argtypes = [_f64p, C.c_int, C.c_int, _f64p, C.c_int, _f64p]
        L.fdo_radii.restype = C.c_int
        L.fdo_build.argtypes = [_f64p, C.c_int, C.c_int, _f64p, C.c_int, C.c_int, _f64p, _f64p,
                                C.POINTER(C.c_int)]
        L.fdo_build.restype = C.c_int
        L.fdo_eval.argtypes = [_f64p, C.c_int, C.c_int, _f64p, _f64p, C.c_int64, _f64p, _f64p]
        L.fdo_eval.restype = None
        L.fdo_project_to_tangents.argtypes = [_f32p, _f32p, _f32p, _f32p]
        L.fdo_project_to_tangents.restype = None
        L.fdo_deform.argtypes = [_f64p, C.c_int, C.c_int, _f64p, _f64p, C.c_int64, _f32p, _f32p,
                                 _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_float, C.c_int]
        L.fdo_deform.restype = C.c_int
        pp = C.POINTER(_f32p)
        L.fdo_morph_shapes_matrix.argtypes = [_f32p, pp, C.c_int64, C.c_int, _f64p]
        L.fdo_morph_shapes_matrix.restype = None
        L.fdo_morph_qr.argtypes = [_f64p, C.c_int64, C.c_int, _f64p]
        L.fdo_morph_qr.restype = None
        L.fdo_morph_weights.argtypes = [_f64p, C.c_int64, C.c_int, _f32p, _f32p, _f64p]
        L.fdo_morph_weights.restype = None
        L.fdo_morph_displace.argtypes = [_f64p, C.c_int64, C.c_int, _f64p, _f32p, C.c_int, C.c_float, _f32p, _f32p]
        L.fdo_morph_displace.restype = None
        L.fdo_capture_dist2.argtypes = [_f32p, C.c_int64, C.c_void_p, _f32p, C.c_int, C.c_float, C.c_int, _f32p]
        L.fdo_capture_dist2.restype = None
        L.fdo_capture_islands.argtypes = [_f32p, C.c_int64, C.c_void_p, C.c_void_p, _f32p, C.c_int, C.c_int, C.c_void_p]
        L.fdo_capture_islands.restype = None

    # -- A2
    def control_table(self, rest, deform):
        rest = np.ascontiguousarray(rest, np.float32).reshape(-1, 3)
        deform = np.ascontiguousarray(deform, np.float32).reshape(-1, 3)
        assert rest.shape == deform.shape
        table = np.empty((rest.shape[0], 6), np.float64)
        self.lib.fdo_control_table(_ptr(rest, _f32p), _ptr(deform, _f32p), rest.shape[0],
                                   _ptr(table, _f64p))
        return table

    # -- A3-A6
    def build(self, table, kind, params=(), term=TERM_LINEAR):
        table = np.ascontiguousarray(table, np.float64)
        M = table.shape[0]
        params = np.ascontiguousarray(np.asarray(params, np.float64).reshape(-1))
        W = np.zeros((M + 4, 3), np.float64)
        radii = np.ones(max(M, 1), np.float64)
        tt = C.c_int(0)
        rc = self.lib.fdo_build(_ptr(table, _f64p), M, kind, _ptr(params, _f64p) if params.size else None,
                                params.size, term, _ptr(W, _f64p), _ptr(radii, _f64p), C.byref(tt))
        return rc, tt.value, W, radii[:M]

    # -- A4 model = 1: alglib::rbfsetalgomultilayer(model, radius, layers, lambda), reference
    #    src/SOP_FaceDeform.cpp:346-348, in the dense form include/facedeform_hip.h states for
    #    FD_KERNEL_GAUSSIAN_ML (ALGLIB itself is absent: parity unpinned against it; this restatement
    #    is pinned by tests/golden/ml_golden.npz, written with SciPy's RBFInterpolator per layer).
    #    numpy fp64 throughout: the term's polynomial by least squares first (LAPACK gelsd), then
    #    (Phi_l + lambda I) w_l = r_l per layer with r_{l+1} = r_l - Phi_l w_l.
    #    Returns tt, the expanded table (layer-major: row l*M + j = centre j), W ((M*L + 4) x 3), radii (M*L):
    #    what eval() / deform() take with kind = KERNEL_GAUSSIAN_QNN (per-record radii).
    def build_multilayer(self, table, radius, layers, lam, term=TERM_LINEAR):
        table = np.ascontiguousarray(table, np.float64)
        M = table.shape[0]
        c, f = table[:, :3], table[:, 3:6].copy()
        T = (4, 1, 0)[term]
        aff = np.zeros((4, 3))
        if T:
            Pm = np.hstack([np.ones((M, 1)), c])[:, :T]
            sol, *_ = np.linalg.lstsq(Pm, f, rcond=None)
            aff[:T] = sol
            f = f - Pm @ sol
        d2 = ((c[:, None, :] - c[None, :, :]) ** 2).sum(-1)
        if M > 1 and (d2 + np.eye(M) == 0.0).any():
            return -5, None, None, None
        W = np.zeros((M * layers + 4, 3))
        radii = np.zeros(M * layers)
        r = f
        for l in range(layers):
            R = radius / 2.0 ** l
            Phi = np.exp(-d2 / (R * R))
            try:
                w = np.linalg.solve(Phi + lam * np.eye(M), r)
            except np.linalg.LinAlgError:
                return -4, None, None, None
            W[l * M:(l + 1) * M] = w
            radii[l * M:(l + 1) * M] = R
            r = r - Phi @ w
        W[M * layers:] = aff
        if not np.isfinite(W).all():
            return -4, None, None, None
        return 1, np.tile(table, (layers, 1)), W, radii

    # -- A8
    def eval(self, table, kind, radii, W, x):
        table = np.ascontiguousarray(table, np.float64)
        x = np.ascontiguousarray(x, np.float64).reshape(-1, 3)
        radii = np.ascontiguousarray(radii, np.float64)
        W = np.ascontiguousarray(W, np.float64)
        out = np.empty_like(x)
        self.lib.fdo_eval(_ptr(table, _f64p), table.shape[0], kind, _ptr(radii, _f64p),
                          _ptr(W, _f64p), x.shape[0], _ptr(x, _f64p), _ptr(out, _f64p))
        return out

    # -- A9
    def project_to_tangents(self, u, v, n, disp):
        u, v, n = (np.ascontiguousarray(a, np.float32) for a in (u, v, n))
        d = np.array(disp, np.float32)
        self.lib.fdo_project_to_tangents(_ptr(u, _f32p), _ptr(v, _f32p), _ptr(n, _f32p), _ptr(d, _f32p))
        return d

    # -- A7-A10
    def deform(self, table, kind, radii, W, P, dist2=None, tangents=None, radius2=1.0,
               falloffrate=1.0, want_falloff=True, nthreads=1, out=None):
        table = np.ascontiguousarray(table, np.float64)
        radii = np.ascontiguousarray(radii, np.float64)
        W = np.ascontiguousarray(W, np.float64)
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        N = P.shape[0]
        P_out = np.empty_like(P) if out is None else out
        d2 = None if dist2 is None else np.ascontiguousarray(dist2, np.float32)
        fall = np.zeros(N, np.float32) if want_falloff else None
        tu = tv = nr = None
        if tangents is not None:
            tu, tv, nr = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in tangents)
        rc = self.lib.fdo_deform(_ptr(table, _f64p), table.shape[0], kind, _ptr(radii, _f64p),
                                 _ptr(W, _f64p), N, _ptr(P, _f32p), _ptr(P_out, _f32p),
                                 _ptr(d2, _f32p), _ptr(fall, _f32p), _ptr(tu, _f32p),
                                 _ptr(tv, _f32p), _ptr(nr, _f32p), float(radius2),
                                 float(falloffrate), int(nthreads))
        if rc != 0:
            raise RuntimeError(f"fdo_deform failed: {rc}")
        return P_out, fall

    # -- next row N1: morph-space reprojection (src/dbse.cpp)
    def morph_shapes_matrix(self, rest, shapes):
        """(3N, S) column-major float64 matrix of fp32 shape deltas (returned as an F-ordered array)."""
        rest = np.ascontiguousarray(rest, np.float32).reshape(-1, 3)
        shapes = [np.ascontiguousarray(s, np.float32).reshape(-1, 3) for s in shapes]
        N, S = rest.shape[0], len(shapes)
        A = np.empty((3 * N, S), np.float64, order="F")
        arr = (_f32p * S)(*[_ptr(s, _f32p) for s in shapes])
        self.lib.fdo_morph_shapes_matrix(_ptr(rest, _f32p), arr, N, S, A.ctypes.data_as(_f64p))
        return A

    def morph_qr(self, A):
        """Householder QR in Eigen's packed form; returns (QR F-ordered, tau)."""
        QR = np.array(A, np.float64, order="F", copy=True)
        tau = np.zeros(QR.shape[1], np.float64)
        self.lib.fdo_morph_qr(QR.ctypes.data_as(_f64p), QR.shape[0], QR.shape[1], _ptr(tau, _f64p))
        return QR, tau

    def morph_weights(self, QR, P, rest):
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        rest = np.ascontiguousarray(rest, np.float32).reshape(-1, 3)
        assert QR.flags.f_contiguous and QR.shape[0] == 3 * P.shape[0]
        w = np.zeros(QR.shape[1], np.float64)
        self.lib.fdo_morph_weights(QR.ctypes.data_as(_f64p), P.shape[0], QR.shape[1], _ptr(P, _f32p),
                                   _ptr(rest, _f32p), _ptr(w, _f64p))
        return w

    def morph_displace(self, shapes_matrix, w, P, rest, clamp=None, add_delta=False, falloffradius=0.0):
        P = np.array(P, np.float32, copy=True).reshape(-1, 3)
        rest = np.ascontiguousarray(rest, np.float32).reshape(-1, 3)
        w = np.ascontiguousarray(w, np.float64)
        assert shapes_matrix.flags.f_contiguous
        cl = None if clamp is None else np.asarray(clamp, np.float32)
        self.lib.fdo_morph_displace(shapes_matrix.ctypes.data_as(_f64p), P.shape[0], shapes_matrix.shape[1],
                                    _ptr(w, _f64p), None if cl is None else _ptr(cl, _f32p), int(bool(add_delta)),
                                    float(falloffradius), _ptr(rest, _f32p), _ptr(P, _f32p))
        return P

    # -- next row N2: dist2 producer (src/capture.cpp:46-99)
    def capture_dist2(self, P, triangles, radius2, dofalloff=True, mask=None):
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        out = np.empty(P.shape[0], np.float32)
        self.lib.fdo_capture_dist2(_ptr(P, _f32p), P.shape[0], None if m is None else m.ctypes.data, _ptr(tri, _f32p),
                                   tri.shape[0], float(radius2), int(bool(dofalloff)), _ptr(out, _f32p))
        return out

    def capture_islands(self, P, offsets, neighbours, rig, max_edges):
        P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
        rig = np.ascontiguousarray(rig, np.float32).reshape(-1, 3)
        offsets = np.ascontiguousarray(offsets, np.int64)
        neighbours = np.ascontiguousarray(neighbours, np.int32)
        assert offsets.shape[0] == P.shape[0] + 1 and offsets[-1] == neighbours.shape[0]
        mask = np.zeros(P.shape[0], np.uint8)
        self.lib.fdo_capture_islands(_ptr(P, _f32p), P.shape[0], offsets.ctypes.data, neighbours.ctypes.data,
                                     _ptr(rig, _f32p), rig.shape[0], int(max_edges), mask.ctypes.data)
        return mask

