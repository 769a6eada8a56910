"""Deterministic synthetic inputs for the deformation path.

The reference ships no assets; only point positions matter to the hot path
(topology is used solely by ProximityCapture, reference src/capture.cpp), so
meshes are Fibonacci lattices.  Everything is rounded to fp32 before use,
because that is what GA_Attribute P holds.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.pi * (1.0 + np.sqrt(5.0))


def fibonacci_sphere(n: int) -> np.ndarray:
    """Unit-sphere Fibonacci lattice, fp64, shape (n, 3)."""
    i = np.arange(n, dtype=np.float64) + 0.5
    z = 1.0 - 2.0 * i / n
    theta = _GOLDEN * i
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    return np.stack([r * np.cos(theta), r * np.sin(theta), z], axis=1)


def sphere_mesh(n: int = 10_000) -> np.ndarray:
    """C1 mesh: unit sphere, fp32."""
    return fibonacci_sphere(n).astype(np.float32)


def _head_shape(unit: np.ndarray) -> np.ndarray:
    """Ellipsoid (0.75, 1.0, 0.85) with a smooth radial bump."""
    x, y, z = unit[:, 0], unit[:, 1], unit[:, 2]
    bump = 1.0 + 0.05 * np.sin(3.0 * x) * np.cos(2.0 * y) + 0.03 * np.sin(5.0 * z)
    return unit * np.array([0.75, 1.0, 0.85]) * bump[:, None]


def head_mesh(n: int = 1_000_000) -> np.ndarray:
    """C2-C5 mesh: 'head' ellipsoid, fp32."""
    return _head_shape(fibonacci_sphere(n)).astype(np.float32)


def control_points(m: int, shape: str = "head") -> np.ndarray:
    """Rest rig: Fibonacci lattice of m well-separated points on the same surface, fp32."""
    unit = fibonacci_sphere(m)
    pts = unit if shape == "sphere" else _head_shape(unit)
    return pts.astype(np.float32)


def smooth_deltas(rest: np.ndarray, frame: int = 0) -> np.ndarray:
    """Primary delta field, phase-shifted by +0.3*frame inside each sine/cosine (C4)."""
    p = rest.astype(np.float64)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    ph = 0.3 * frame
    d = 0.05 * np.stack([np.sin(2 * x + y + ph), np.cos(3 * y + ph), np.sin(2 * z * x + ph)], axis=1)
    return d.astype(np.float32)


def noise_deltas(m: int, seed: int = 1234) -> np.ndarray:
    """Stress deltas: i.i.d. N(0, 0.05^2) (ill-conditioned weights; needs fp64 evaluation)."""
    rng = np.random.default_rng(seed)
    return (0.05 * rng.standard_normal((m, 3))).astype(np.float32)


def deformed_rig(rest: np.ndarray, frame: int = 0) -> np.ndarray:
    """Deformed rig positions (input 2 of the SOP) = rest + smooth delta, fp32."""
    return (rest + smooth_deltas(rest, frame)).astype(np.float32)


def rig_deltas(rest: np.ndarray, frame: int = 0) -> np.ndarray:
    """The control table's delta as the reference forms it from its two rig inputs (src/SOP_FaceDeform.cpp:278): the fp32
    difference deformed_rig - rest.  NOT smooth_deltas(rest, frame) bit for bit -- the deformed rig's positions are rounded to
    fp32 first -- so a test that hands the engine deltas and the oracle the two rigs uses this on both sides."""
    rest = np.asarray(rest, np.float32)
    return (deformed_rig(rest, frame) - rest).astype(np.float32)


def tangent_frames(mesh: np.ndarray, seed: int = 7):
    """Non-unit, non-orthogonal (tangentu, tangentv, N) per vertex, fp32 -- exercises
    the in-place normalisation and the non-orthogonal axes of project_to_tangents."""
    rng = np.random.default_rng(seed)
    p = mesh.astype(np.float64)
    nrm = p / np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-12)
    ref = np.where(np.abs(nrm[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    tu = np.cross(ref, nrm)
    tv = np.cross(nrm, tu) + 0.15 * tu
    scale = 0.5 + rng.random((mesh.shape[0], 3))
    return ((tu * scale[:, 0:1]).astype(np.float32), (tv * scale[:, 1:2]).astype(np.float32),
            (nrm * scale[:, 2:3]).astype(np.float32))


def parity_error(delta_test: np.ndarray, delta_ref: np.ndarray) -> np.ndarray:
    """Per-vertex parity metric of SURVEY.md section 8d:
    |d_test - d_ref| / max(|d_ref|, 1e-5 * max_v |d_ref|)."""
    dt = np.asarray(delta_test, np.float64)
    dr = np.asarray(delta_ref, np.float64)
    nr = np.linalg.norm(dr, axis=1)
    floor = 1e-5 * (nr.max() if nr.size else 0.0)
    return np.linalg.norm(dt - dr, axis=1) / np.maximum(np.maximum(nr, floor), 1e-300)
