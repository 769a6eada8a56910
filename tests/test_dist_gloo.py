"""CPU, world_size 2 over gloo: the N > 1 paths of SURVEY.md 8e.

The GPU engine cannot run here, so the per-rank evaluation is done by the oracle (the
checker standing in for the device); what is under test is the sharding arithmetic and the
single exchange step: frames round-robin with no collective, and the vertex-range split
with ONE broadcast of the solved-model blob from the solving rank."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from facedeform_amd import dist as fdist


def test_vertex_ranges_tile_the_mesh_exactly():
    for n in (0, 1, 1023, 1024, 1025, 10_000, 1_000_000, 10_000_000):
        for world in (1, 2, 3, 8):
            ranges = [fdist.vertex_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            for (a0, a1), (b0, b1) in zip(ranges[:-1], ranges[1:]):
                assert a1 == b0 and a0 <= a1
            for a0, _ in ranges:
                assert a0 % fdist.GA_PAGE == 0 or a0 == n        # page-aligned starts
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= fdist.GA_PAGE * 1 + (n % fdist.GA_PAGE)


def test_frames_round_robin_cover_every_frame_once():
    for n_frames in (0, 1, 7, 8, 64):
        for world in (1, 2, 8):
            seen = sorted(f for r in range(world) for f in fdist.frames_for_rank(n_frames, r, world))
            assert seen == list(range(n_frames))
    with pytest.raises(ValueError):
        fdist.frames_for_rank(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from facedeform_amd import synth
    from oracle import fd_oracle as fo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = fo.Oracle()
    N, M = 5000, 48
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")

    # --- vertex-range split: rank 0 solves, ONE broadcast, every rank evaluates its range
    nbytes = 8 * (M * 6 + M + (M + 4) * 3)
    blob = torch.zeros(nbytes, dtype=torch.uint8)
    if rank == 0:
        table = orc.control_table(rest, synth.deformed_rig(rest))
        rc, tt, W, radii = orc.build(table, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
        assert tt == 1
        blob = torch.from_numpy(np.concatenate([table.ravel(), radii, W.ravel()]).view(np.uint8).copy())
    fdist.broadcast_model(blob, src=0)
    flat = blob.numpy().view(np.float64)
    table = flat[: M * 6].reshape(M, 6)
    radii = flat[M * 6: M * 7]
    W = flat[M * 7:].reshape(M + 4, 3)
    lo, hi = fdist.vertex_range(N, rank, world)
    part, _ = orc.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P[lo:hi])
    np.save(os.path.join(tmpdir, f"part{rank}.npy"), part)

    # --- frames: no collective at all, each rank cooks its own frames
    mine = fdist.frames_for_rank(6, rank, world)
    outs = {}
    for f in mine:
        tf = orc.control_table(rest, synth.deformed_rig(rest, f))
        rc, tt, Wf, rf = orc.build(tf, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
        outs[f] = orc.deform(tf, fo.KERNEL_THIN_PLATE, rf, Wf, P[:500])[0]
    np.savez(os.path.join(tmpdir, f"frames{rank}.npz"), **{str(k): v for k, v in outs.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_split_mesh_and_frames(tmp_path, oracle):
    import torch.multiprocessing as mp
    from facedeform_amd import synth
    from oracle import fd_oracle as fo
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    N, M = 5000, 48
    P = synth.head_mesh(N)
    rest = synth.control_points(M, "head")
    table = oracle.control_table(rest, synth.deformed_rig(rest))
    rc, tt, W, radii = oracle.build(table, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
    whole, _ = oracle.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    parts = np.concatenate([np.load(tmp_path / f"part{r}.npy") for r in range(world)])
    assert np.array_equal(parts, whole)           # split + broadcast == single process, bit for bit
    seen = {}
    for r in range(world):
        with np.load(tmp_path / f"frames{r}.npz") as z:
            for k in z.files:
                assert int(k) not in seen
                seen[int(k)] = z[k]
    assert sorted(seen) == list(range(6))
    for f, out in seen.items():
        tf = oracle.control_table(rest, synth.deformed_rig(rest, f))
        rc, tt, Wf, rf = oracle.build(tf, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
        assert np.array_equal(out, oracle.deform(tf, fo.KERNEL_THIN_PLATE, rf, Wf, P[:500])[0])
