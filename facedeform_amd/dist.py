"""Sharding of the deformation path over the GPUs of one node.

Two decompositions, both with at most one exchange step:
  * frames / blendshapes: independent units, one per GPU at a time, no collective;
  * vertex ranges of one mesh: contiguous ranges aligned to GA pages (1024 points);
    the solving rank broadcasts the solved model blob (centres + radii + weights,
    tens of KB) once -- torch.distributed broadcast, i.e. RCCL over xGMI with the
    "nccl" backend, gloo on CPU in the tests.
Pure index arithmetic + one collective; no compute here.
"""
from __future__ import annotations

GA_PAGE = 1024  # Houdini GA page size: range starts stay page-aligned


def frames_for_rank(n_frames: int, rank: int, world: int) -> list[int]:
    """Round-robin: frame f goes to rank f % world."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_frames, world))


def vertex_range(n_verts: int, rank: int, world: int, align: int = GA_PAGE) -> tuple[int, int]:
    """Contiguous [begin, end) of `rank`; starts are multiples of `align`; the union over
    ranks is exactly [0, n_verts) with no overlap; empty ranges are allowed."""
    if world <= 0 or not (0 <= rank < world) or n_verts < 0 or align <= 0:
        raise ValueError("bad arguments")
    pages = (n_verts + align - 1) // align
    lo = (pages * rank) // world * align
    hi = (pages * (rank + 1)) // world * align
    return min(lo, n_verts), min(hi, n_verts)


def broadcast_model(blob, src: int = 0, group=None):
    """Broadcast a solved-model blob (a uint8 torch tensor, host or device) from `src`.
    Every rank passes a tensor of fd_model_bytes() bytes; returns it filled."""
    import torch.distributed as dist
    dist.broadcast(blob, src=src, group=group)
    return blob
