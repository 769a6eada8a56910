"""Phase stamps of k_build_reg for one single build at M control points (tuning build, FD_REG_STAMPS=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facedeform_amd import capi, synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rest = synth.control_points(M, "head"); delta = synth.rig_deltas(rest, 0)
e = capi.Engine(solver=capi.SOLVER_REGISTER); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
for i in range(4):
    e.set_points(rest, delta); rep = e.build()
print("terminationtype", rep.terminationtype, "solver", rep.solver_used)
e.close()
