"""Parity ratio (error / tolerance, tolerance = north_star's 1e-5) of the thin-plate evaluation
variants on the hard cases: large length unit and far-away origin, small length unit, many centres."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conftest import parity_ratio
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

orc = fo.Oracle()
for M, N, scale, offset in ((256, 20000, 1.0, 0.0), (256, 20000, 100.0, 500.0), (256, 20000, 0.01, 0.0), (256, 20000, 1.0, 30.0),
                            (2048, 4000, 1.0, 0.0), (2048, 4000, 50.0, -200.0), (700, 8000, 3.0, 10.0)):
    rest = (synth.control_points(M, "head") * np.float32(scale) + np.float32(offset)).astype(np.float32)
    deform = (rest + synth.smooth_deltas(synth.control_points(M, "head"), 1) * np.float32(scale)).astype(np.float32)
    P = (synth.head_mesh(N) * np.float32(scale) + np.float32(offset)).astype(np.float32)
    table = orc.control_table(rest, deform)
    rc, tt, W, radii = orc.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    ref, _ = orc.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    row = []
    for var in (102, 200, 202):
        e = capi.Engine(variant=var)
        e.set_points(rest, (deform - rest).astype(np.float32)); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0); e.build()
        out, _ = e.deform(P)
        row.append(parity_ratio(out, ref, P, 1e-5))
        e.close()
    print(f"M={M:5d} scale {scale:6g} offset {offset:6g}: parity ratio (<= 1 passes)  VALU {row[0]:.3f}   bf16x3 {row[1]:.3f}   fp16x2 {row[2]:.3f}", flush=True)
