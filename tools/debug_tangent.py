import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo
from conftest import parity_ratio
orc = fo.Oracle()
for M, N in ((32, 10007), (256, 20000), (800, 3001)):
    P = synth.head_mesh(200_000)[:: 200_000 // N][:N].copy()
    rest = synth.control_points(M, "head"); P[:8] = rest[:8]
    deform = synth.deformed_rig(rest, 2)
    tu, tv, nn = synth.tangent_frames(P)
    table = orc.control_table(rest, deform); rc, tt, W, radii = orc.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    ref, _ = orc.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P, tangents=(tu, tv, nn))
    refn, _ = orc.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    for var in (102, 200, 2):
        for prec in (0, 1):
            e = capi.Engine(variant=var, precision=prec); e.set_points(rest, (deform-rest).astype(np.float32)); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0); e.build()
            out, _ = e.deform(P, tangents=(tu, tv, nn)); outn, _ = e.deform(P)
            d = np.abs(out.astype(np.float64)-ref).max(axis=1); w = int(np.argmax(d / np.maximum(np.linalg.norm(ref.astype(np.float64)-P,axis=1),1e-12)))
            print(f"M={M} variant {var} prec {prec}: tangent ratio@3e-5 {parity_ratio(out, ref, P, 3e-5):.3f}  plain ratio@1e-5 {parity_ratio(outn, refn, P, 1e-5):.3f}; worst vertex {w}: |proj d|={np.linalg.norm(ref[w].astype(np.float64)-P[w]):.3e} |d|={np.linalg.norm(refn[w].astype(np.float64)-P[w]):.3e}")
            e.close()
