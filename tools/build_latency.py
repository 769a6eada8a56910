"""Single-build latency of the two direct solvers (FD_SOLVER_AUTO = null-space Cholesky for
thin-plate + linear term, FD_SOLVER_LU = pivoted LU), thin-plate kernel, linear term.
Wall clock around fd_build (hipGraph replay + the status read-back), median of `reps`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facedeform_amd import capi, synth

sizes = [int(s) for s in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 256, 512, 1024, 2048, 4096]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
print("M      solver    build ms (median)   assemble+project ms   factor+solve ms   fp64 GFLOP/s on (2/3)n^3 [LU] or (1/3)n1^3 [Cholesky]")
for M in sizes:
    rest = synth.control_points(M, "head")
    delta = synth.smooth_deltas(rest, 0).astype(np.float32)
    for name, solver in (("cholesky", capi.SOLVER_AUTO), ("lu", capi.SOLVER_LU)):
        e = capi.Engine(solver=solver)
        e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
        ts, ta, tsol = [], [], []
        for r in range(reps + 3):
            e.set_points(rest, delta)
            t0 = time.perf_counter()
            rep = e.build()
            t1 = time.perf_counter()
            if r >= 3:
                ts.append((t1 - t0) * 1e3); ta.append(rep.t_assemble_ms); tsol.append(rep.t_solve_ms)
        n = M + 4
        flops = 2.0 / 3.0 * n ** 3 if solver == capi.SOLVER_LU else 1.0 / 3.0 * (M - 4) ** 3
        t = float(np.median(ts))
        print(f"{M:5d}  {name:9s} {t:10.3f}          {float(np.median(ta)):10.3f}          {float(np.median(tsol)):10.3f}      {flops / (t * 1e-3) / 1e9:10.1f}", flush=True)
        e.close()
