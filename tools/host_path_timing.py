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
# the same cook on page-locked arrays (fd_host_alloc): chunked, copies overlap evaluation
pin_in = capi.host_array((N, 3)); pin_in[:] = P
pin_out = capi.host_array((N, 3)); pin_fall = capi.host_array(N)
page_out = np.empty_like(P); page_fall = np.zeros(N, np.float32)
for label, (a_in, a_out, a_fall) in (("pageable, no binding copy", (P, page_out, page_fall)), ("page-locked", (pin_in, pin_out, pin_fall))):
    ts = []
    for f in range(12):
        delta = synth.smooth_deltas(rest, f)
        a_fall[:] = 0
        t0 = time.perf_counter()
        e.set_points(rest, delta); e.build(); e.deform_into(a_in, a_out, None, a_fall)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    print(f"host-pointer cook, {label}: median {ts[len(ts)//2]*1e3:.3f} ms -> {N/ts[len(ts)//2]/1e6:.0f} Mverts/s")
    ts = []
    for f in range(12):
        t0 = time.perf_counter(); e.deform_into(a_in, a_out, None, a_fall); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    print(f"   fd_deform alone (H2D 12 MB + evaluate + D2H 16 MB), {label}: median {ts[len(ts)//2]*1e3:.3f} ms")
node = FaceDeformSOP(); node.set("kernel", 1)
ts = []
for f in range(8):
    t0 = time.perf_counter(); res = node.cook(P, rest, synth.deformed_rig(rest, f)); ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print(f"fdsop_cook through the C++ host mirror (same sizes, + Cd fill, fd_falloff): median {ts[len(ts)//2]*1e3:.3f} ms; severity {res.severity}; {res.infos}")
ts = []
for f in range(8):
    t0 = time.perf_counter(); res = node.cook(pin_in, rest, synth.deformed_rig(rest, f), out_P=pin_out, out_falloff=pin_fall, want_Cd=False); ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print(f"fdsop_cook on page-locked mesh arrays, Cd left to the attribute default (what hdk/SOP_FaceDeformHip.cpp does): median {ts[len(ts)//2]*1e3:.3f} ms; severity {res.severity}")
ts = []
for f in range(10):
    t0 = time.perf_counter(); res = node.cook(pin_in, rest, synth.deformed_rig(rest, f), out_P=pin_out, out_falloff=pin_fall, want_Cd=False, rig_rest_unchanged=True); ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print(f"... and the rest rig unchanged from cook to cook (fd_set_deltas: factorisation reused): median {ts[len(ts)//2]*1e3:.3f} ms; severity {res.severity}")
# static mesh + animated rig, engine level: mesh on the device (fd_mesh_set), deltas only (fd_set_deltas)
e.set_points(rest, synth.smooth_deltas(rest, 0)); e.build()
e.mesh_set(P)
for label, (a_out, a_fall) in (("pageable outputs", (page_out, page_fall)), ("page-locked outputs", (pin_out, pin_fall))):
    ts = []
    for f in range(12):
        delta = synth.smooth_deltas(rest, f)
        t0 = time.perf_counter()
        e.set_deltas(delta); e.build(); e.deform_mesh(a_out, a_fall)
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    print(f"static mesh on the device + fd_set_deltas + fd_deform_mesh, {label}: median {ts[len(ts)//2]*1e3:.3f} ms -> {N/ts[len(ts)//2]/1e6:.0f} Mverts/s")
ts = []
for f in range(10):
    t0 = time.perf_counter(); res = node.cook(pin_in, rest, synth.deformed_rig(rest, f), out_P=pin_out, out_falloff=pin_fall, want_Cd=False, rig_rest_unchanged=True, mesh_unchanged=True); ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print(f"fdsop_cook, page-locked arrays, rest rig AND mesh unchanged from cook to cook (the animated-shot case): median {ts[len(ts)//2]*1e3:.3f} ms; severity {res.severity}")

# the SOP's own two models at its default parameters (model = 0: QNN q = 1, z = 5; model = 1: multilayer
# radius 1, 4 layers, lambda 0.1), page-locked arrays, everything new every cook
for label, settings in (("model = 0 (QNN radii, the SOP's default; pivoted LU + Gaussian kernel)", {"model": "0"}),
                        ("model = 1 (multilayer, 4 layers; 4 Cholesky solves + shared-distance Gaussian kernel)", {"model": "1"})):
    n2 = FaceDeformSOP()
    for k, v in settings.items():
        n2.set(k, v)
    ts = []
    for f in range(8):
        t0 = time.perf_counter(); res = n2.cook(pin_in, rest, synth.deformed_rig(rest, f), out_P=pin_out, out_falloff=pin_fall, want_Cd=False); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[2:])
    print(f"fdsop_cook, page-locked arrays, {label}: median {ts[len(ts)//2]*1e3:.3f} ms; severity {res.severity}; {res.infos}")
    n2.close()
