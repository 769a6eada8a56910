"""How the parity of the thin-plate kernels behaves when the deltas are scaled up (the one-ulp
term of the tolerance shelters less and less): one-frame matrix-pipe kernel (default), all-VALU
kernel (variant 102), shared-rig kernel, fp64 evaluation.  Worst vertex of parity_ratio / raw L2.
    python tests/tools/scaled_delta_parity.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from conftest import parity_ratio, l2_parity, l2_parity_ulp
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

orc = fo.Oracle()
M, N = 256, 70_000
P = synth.head_mesh(100_000)[:: 100_000 // N][:N].copy()
rest = synth.control_points(M, "head")
dev = torch.device("cuda", 0)
d_P = torch.from_numpy(P).to(dev); d_rest = torch.from_numpy(rest).to(dev)
for frame, scale in ((0, 1.0), (19, 1.0), (19, 4.0), (19, 8.0), (19, 32.0), (5, 100.0)):
    delta = (synth.smooth_deltas(rest, frame) * np.float32(scale)).astype(np.float32)
    table = orc.control_table(rest, (rest + delta).astype(np.float32))
    _, _, W, radii = orc.build(table, fo.KERNEL_THIN_PLATE, [], 0)
    ref, _ = orc.deform(table, fo.KERNEL_THIN_PLATE, radii, W, P)
    row = []
    for name, kw in (("mfma-1frame", {}), ("valu-102", dict(variant=102)), ("fp64", dict(precision=capi.EVAL_FP64))):
        e = capi.Engine(**kw); e.set_points(rest, delta); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0); e.build()
        out, _ = e.deform(P); e.close()
        row.append(f"{name} ratio {parity_ratio(out, ref, P, 1e-5):5.2f} l2ulp {l2_parity_ulp(out, ref, P):5.2f} raw {l2_parity(out, ref, P):.1e}")
    d_del = torch.from_numpy(np.stack([delta, delta])).to(dev)
    engines = []
    for _ in range(2):
        e = capi.Engine(); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0); engines.append(e)
    b = capi.Batch(engines)
    b.set_points_dev([d_rest.data_ptr()] * 2, [d_del.data_ptr(), d_del.data_ptr() + M * 12], M)
    b.build_async(); b.build_result()
    outs = [torch.empty_like(d_P) for _ in range(2)]
    b.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in outs]); torch.cuda.synchronize()
    out = outs[1].cpu().numpy()
    row.append(f"shared ratio {parity_ratio(out, ref, P, 1e-5):5.2f} l2ulp {l2_parity_ulp(out, ref, P):5.2f} raw {l2_parity(out, ref, P):.1e}")
    b.close(); [e.close() for e in engines]
    print(f"frame {frame:2d} x{scale:5.1f}: " + " | ".join(row), flush=True)
