"""Multilayer Gaussian model (FD_KERNEL_GAUSSIAN_ML, the SOP's model = 1) at the SOP's defaults
(radius 1, 4 layers, lambda 0.1): build time and device-resident evaluation of 1M vertices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facedeform_amd import capi, synth

N = 1_000_000
dev = torch.device("cuda", 0)
d_P = torch.from_numpy(synth.head_mesh(N)).to(dev)
d_out = torch.empty_like(d_P)
d_fall = torch.empty(N, device=dev)
stream = torch.cuda.Stream(device=dev)
for M, L in ((256, 1), (256, 4), (256, 8), (1024, 4)):
    rest = synth.control_points(M, "head")
    delta = synth.smooth_deltas(rest, 0).astype(np.float32)
    e = capi.Engine(); e.set_stream(stream.cuda_stream)
    e.set_kernel(capi.KERNEL_GAUSSIAN_ML, [1.0, L, 0.1]); e.set_term(capi.TERM_LINEAR)
    tb = []
    for r in range(8):
        e.set_points(rest, delta)
        t0 = time.perf_counter(); rep = e.build(); tb.append(time.perf_counter() - t0)
    assert rep.terminationtype == 1
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(stream)
        e.deform_dev(N, d_P.data_ptr(), d_out.data_ptr(), 0, d_fall.data_ptr())
        b.record(stream)
    stream.synchronize()
    te = sorted(a.elapsed_time(b) for a, b in evs[2:])[4]
    pairs = N * M * L
    print(f"M={M:5d} layers={L}: build {sorted(tb[2:])[3]*1e3:7.3f} ms ({L} Cholesky solves of order {M}); evaluate {te*1e3:8.1f} us "
          f"= {N/te/1e3:7.0f} Mverts/s over {M*L} Gaussian records, {16*pairs/te/1e9:6.1f} TFLOP/s at 16 flops per pair "
          f"({16*pairs/te/1e9/157.3*100:4.1f}% of the fp32 peak)", flush=True)
    e.set_stream(None); e.close()
