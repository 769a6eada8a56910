"""QNN model (the SOP's default) at M = 256: fd_build without the pivot search (FD_SOLVER_AUTO) against the pivoted LU
(FD_SOLVER_LU), and fdsop_cook with the node's defaults on a 100k-vertex mesh.  Wall clock, median.
    python tools/qnn_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facedeform_amd import capi, synth
from facedeform_amd.sop import FaceDeformSOP

for M in (64, 256, 512, 1000):
    rest = synth.control_points(M, "head"); delta = synth.smooth_deltas(rest, 0).astype(np.float32)
    for name, solver in (("no pivot search (AUTO)", capi.SOLVER_AUTO), ("pivoted LU", capi.SOLVER_LU)):
        e = capi.Engine(solver=solver); e.set_kernel(capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0, 0.0]); e.set_term(capi.TERM_LINEAR)
        ts = []
        for r in range(23):
            e.set_points(rest, delta)
            t0 = time.perf_counter(); rep = e.build(); ts.append(time.perf_counter() - t0)
        print(f"M={M:5d} QNN q=1 fd_build {name:24s}: median {np.median(ts[3:]) * 1e3:.3f} ms (solver_used {rep.solver_used})", flush=True)
        e.close()
M = 256
rest = synth.control_points(M, "head"); deform = (rest + synth.smooth_deltas(rest, 0)).astype(np.float32)
for N in (100_000,):
    P = synth.head_mesh(N)
    out = capi.host_array((N, 3), np.float32) if hasattr(capi, "host_array") else None
    sop = FaceDeformSOP()
    ts = []
    for r in range(23):
        t0 = time.perf_counter(); res = sop.cook(P, rest, deform, want_Cd=False, mesh_unchanged=r > 0); ts.append(time.perf_counter() - t0)
    print(f"fdsop_cook, the node's defaults (QNN q=1 z=5, linear term), M={M}, N={N}, mesh resident: median {np.median(ts[3:]) * 1e3:.3f} ms; messages {res.messages}", flush=True)
    sop.close()
