"""Python handle on the C++ cook mirror (fdsop_* in include/facedeform_hip.h).

Mirrors the node a Houdini user sees: parms by token (reference
src/SOP_FaceDeform.cpp:99-137) and a cook over arrays standing in for the
mesh / rest rig / deformed rig inputs (:215-489).  All logic lives in
csrc/fd_sop_host.cpp; this file only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class CookResult:
    severity: int
    messages: list = field(default_factory=list)   # (severity_name, text)
    P: np.ndarray | None = None
    fd_falloff: np.ndarray | None = None
    Cd: np.ndarray | None = None
    weights: np.ndarray | None = None              # the detail array `weights` of the morph pass
    dist2: np.ndarray | None = None                # ProximityCapture's attribute when the cook captured on the device

    @property
    def errors(self):
        return [t for s, t in self.messages if s == "error"]

    @property
    def warnings(self):
        return [t for s, t in self.messages if s == "warning"]

    @property
    def infos(self):
        return [t for s, t in self.messages if s == "message"]


class FaceDeformSOP:
    def __init__(self, device: int = -1, precision: int = capi.EVAL_FP32, variant: int = 0):
        self.L = capi.load()
        cfg = capi.FdConfig(C.sizeof(capi.FdConfig), device, precision, variant)
        self.node = self.L.fdsop_create(C.byref(cfg))
        if not self.node:
            raise MemoryError("fdsop_create failed")

    def close(self):
        if getattr(self, "node", None):
            self.L.fdsop_destroy(self.node)
            self.node = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parm surface
    @staticmethod
    def parm_tokens():
        L = capi.load()
        return [L.fdsop_parm_token(i).decode() for i in range(L.fdsop_parm_count())]

    def set(self, token: str, value, index: int = 0):
        tok = token.encode()
        if isinstance(value, str):
            rc = self.L.fdsop_set_string(self.node, tok, value.encode())
        elif isinstance(value, (tuple, list)):
            rc = 0
            for i, v in enumerate(value):
                rc = rc or self.L.fdsop_set_float(self.node, tok, i, float(v))
        else:
            rc = self.L.fdsop_set_float(self.node, tok, index, float(value))
        if rc != 0:
            raise KeyError(f"bad parm {token!r} / value {value!r}")

    def get(self, token: str, index: int = 0) -> float:
        v = C.c_double()
        if self.L.fdsop_get_float(self.node, token.encode(), index, C.byref(v)) != 0:
            raise KeyError(token)
        return v.value

    def get_int(self, token: str) -> int:
        v = C.c_int()
        if self.L.fdsop_get_int(self.node, token.encode(), C.byref(v)) != 0:
            raise KeyError(token)
        return v.value

    def effective(self, token: str) -> float:
        """Clamped value used by the last cook (reference :249-257)."""
        v = C.c_double()
        if self.L.fdsop_effective_float(self.node, token.encode(), C.byref(v)) != 0:
            raise KeyError(token)
        return v.value

    def engine(self) -> capi.Engine | None:
        ctx = self.L.fdsop_engine(self.node)
        if not ctx:
            return None
        e = capi.Engine(_borrowed=ctx)
        return e

    # -- cook
    def cook(self, mesh_P, rest_P, deform_P, dist2=None, tangentu=None, tangentv=None, N=None,
             out_P=None, out_falloff=None, want_Cd=True, shapes=None, rest=None, rest_changed=False,
             blends_changed=False, rig_rest_unchanged=False, mesh_unchanged=False,
             edge_offsets=None, edge_neighbours=None, rig_tris=None, want_dist2=False) -> CookResult:
        """out_P / out_falloff: caller-owned result arrays (e.g. page-locked ones from
        capi.host_array, as the HDK wrapper keeps them): with every mesh array page-locked the
        evaluation runs in place over the host link.  want_Cd=False leaves the Cd fill to the
        attribute default, as the wrapper does."""
        f32 = np.float32
        P = np.ascontiguousarray(mesh_P, f32).reshape(-1, 3)
        rig_rest = np.ascontiguousarray(rest_P, f32).reshape(-1, 3)
        deform = np.ascontiguousarray(deform_P, f32).reshape(-1, 3)
        keep = [P, rig_rest, deform]

        def opt(a, cols):
            if a is None:
                return None
            arr = np.ascontiguousarray(a, f32).reshape(-1, cols) if cols > 1 else np.ascontiguousarray(a, f32)
            keep.append(arr)
            return arr

        d2, tu, tv, nn = opt(dist2, 1), opt(tangentu, 3), opt(tangentv, 3), opt(N, 3)
        P_out = np.empty_like(P) if out_P is None else out_P
        fall = np.empty(P.shape[0], f32) if out_falloff is None else out_falloff
        if P_out.dtype != f32 or not P_out.flags.c_contiguous or P_out.shape != P.shape:
            raise ValueError("out_P must be a C-contiguous float32 array shaped like the mesh")
        if fall.dtype != f32 or not fall.flags.c_contiguous or fall.shape != (P.shape[0],):
            raise ValueError("out_falloff must be a C-contiguous float32 array with one entry per point")
        Cd = np.empty_like(P) if want_Cd else None
        fp = capi._f32p

        def ptr(a):
            return None if a is None else a.ctypes.data_as(fp)

        # inputs 3..: blendshapes of the morph-space pass; `rest` = input 0's rest attribute
        shape_arrs = [np.ascontiguousarray(sh, f32).reshape(-1, 3) for sh in (shapes or [])]
        keep.extend(shape_arrs)
        ns = len(shape_arrs)
        sh_ptrs = (C.c_void_p * max(1, ns))(*[a.ctypes.data for a in shape_arrs])
        sh_counts = (C.c_int64 * max(1, ns))(*[a.shape[0] for a in shape_arrs])
        rest_attr = opt(rest, 3)
        weights = np.zeros(max(1, ns), np.float64)
        wcount = C.c_int64(0)
        # ProximityCapture's inputs: the mesh's edge adjacency (CSR) and the rig's surface triangles
        eo = en = tris = cap_out = None
        if edge_offsets is not None:
            eo = np.ascontiguousarray(edge_offsets, np.int64)
            en = np.ascontiguousarray(edge_neighbours if edge_neighbours is not None else [], np.int32)
            tris = np.ascontiguousarray(rig_tris if rig_tris is not None else np.zeros((0, 9)), f32).reshape(-1, 9)
            keep.extend([eo, en, tris])
            if want_dist2:
                cap_out = np.full(P.shape[0], np.nan, f32)
        geo = capi.FdsopGeo(P.shape[0], ptr(P), ptr(tu), ptr(tv), ptr(nn), ptr(d2), rig_rest.shape[0],
                            deform.shape[0], ptr(rig_rest), ptr(deform), ptr(P_out), ptr(fall), ptr(Cd),
                            ns, C.cast(sh_ptrs, C.POINTER(C.c_void_p)) if ns else None,
                            C.cast(sh_counts, C.POINTER(C.c_int64)) if ns else None, ptr(rest_attr),
                            int(bool(rest_changed)), int(bool(blends_changed)),
                            weights.ctypes.data_as(C.POINTER(C.c_double)), C.pointer(wcount),
                            int(bool(rig_rest_unchanged)), int(bool(mesh_unchanged)),
                            None if eo is None else eo.ctypes.data_as(C.POINTER(C.c_int64)),
                            None if en is None or en.size == 0 else en.ctypes.data_as(C.POINTER(C.c_int)),
                            -1 if tris is None else tris.shape[0],
                            None if tris is None or tris.size == 0 else tris.ctypes.data_as(fp), ptr(cap_out))
        sev = self.L.fdsop_cook(self.node, C.byref(geo))
        text = self.L.fdsop_messages(self.node).decode()
        msgs = [tuple(line.split("\t", 1)) for line in text.splitlines() if "\t" in line]
        res = CookResult(sev, msgs, P_out, fall, Cd)
        res.weights = weights[: wcount.value].copy()
        res.dist2 = cap_out
        return res
