"""QNN build at M = 256, 40 builds: for rocprofv3 --kernel-trace --stats (per-kernel durations of the launch chain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facedeform_amd import capi, synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rest = synth.control_points(M, "head"); delta = synth.smooth_deltas(rest, 0).astype(np.float32)
e = capi.Engine(); e.set_kernel(capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0, 0.0]); e.set_term(capi.TERM_LINEAR)
for r in range(40):
    e.set_points(rest, delta); rep = e.build()
print("solver_used", rep.solver_used, "tt", rep.terminationtype)
e.close()
