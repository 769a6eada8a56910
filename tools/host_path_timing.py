"""PCIe-inclusive timing of the host-pointer boundary (fd_set_points + fd_build + fd_deform on
caller-owned host arrays), C2 sizes.  Not the bench `value` (that one keeps inputs in HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facedeform_amd import capi, synth
from facedeform_amd.sop import FaceDeformSOP
N, M = 1_000_000, 256
P = synth.head_mesh(N); rest = synth.control_points(M, "head")
e = capi.Engine(); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
ts = []
for f in range(12):
    delta = synth.smooth_deltas(rest, f)
    t0 = time.perf_counter()
    e.set_points(rest, delta); e.build(); out, fall = e.deform(P)
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print(f"host-pointer cook (H2D 12 MB + build + evaluate + D2H 16 MB, pageable numpy arrays, incl. one 12 MB host copy in the binding): median {ts[len(ts)//2]*1e3:.3f} ms -> {N/ts[len(ts)//2]/1e6:.0f} Mverts/s")
node = FaceDeformSOP(); node.set("kernel", 1)
ts = []
for f in range(8):
    t0 = time.perf_counter(); res = node.cook(P, rest, synth.deformed_rig(rest, f)); ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print(f"fdsop_cook through the C++ host mirror (same sizes, + Cd fill, fd_falloff): median {ts[len(ts)//2]*1e3:.3f} ms; severity {res.severity}; {res.infos}")
