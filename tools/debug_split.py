import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from facedeform_amd import capi, synth
P = synth.head_mesh(50_000)
rest = synth.control_points(128, "head")
deform = synth.deformed_rig(rest)
e = capi.Engine()
e.set_points(rest, (deform-rest).astype(np.float32)); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0); e.build()
w1,_ = e.deform(P); w2,_ = e.deform(P)
print("whole repeat identical:", np.array_equal(w1,w2), "nmismatch", (w1!=w2).any(axis=1).sum())
for a,b in [(0,1024),(1024,13312),(13312,13313),(13313,40000),(40000,50000),(0,50000),(1,50000),(3,1027)]:
    part,_ = e.deform(P[a:b])
    bad = (part != w1[a:b]).any(axis=1)
    print(a,b,"mismatch",bad.sum(), "first idx", np.nonzero(bad)[0][:8], "maxabs", np.abs(part-w1[a:b]).max())
