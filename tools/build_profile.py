"""A few single builds of one size under rocprofv3 --kernel-trace --stats (thin-plate, linear term).
usage: build_profile.py M [cholesky|lu] [builds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facedeform_amd import capi, synth

M = int(sys.argv[1]); solver = capi.SOLVER_LU if len(sys.argv) > 2 and sys.argv[2] == "lu" else capi.SOLVER_AUTO
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rest = synth.control_points(M, "head")
delta = synth.smooth_deltas(rest, 0).astype(np.float32)
e = capi.Engine(solver=solver)
e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
for r in range(n):
    e.set_points(rest, delta)
    assert e.build().terminationtype == 1
e.close()
