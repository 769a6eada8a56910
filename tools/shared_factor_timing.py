"""Batched build of 32 frames of one rig: every model on its own (k_build_reg x 32) against one factorisation for the group
(fd_batch_set_shared_factor: k_build_reg + k_resolve_reg x 31).  Wall clock of fd_batch_build_async + result, and HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facedeform_amd import capi, synth
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
F = 32
dev = torch.device("cuda", 0)
rest = synth.control_points(M, "head")
deltas = np.stack([synth.rig_deltas(rest, f) for f in range(F)])
d_rest, d_del = torch.from_numpy(rest).to(dev), torch.from_numpy(deltas).to(dev)
stream = torch.cuda.Stream(device=dev)
engines = []
for _ in range(F):
    e = capi.Engine(); e.set_stream(stream.cuda_stream); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0); engines.append(e)
batch = capi.Batch(engines)
for on in (False, True, False, True):
    batch.set_shared_factor(on)
    ts, evs = [], []
    for rep in range(12):
        batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + k * M * 12 for k in range(F)], M)
        stream.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record(stream)
        batch.build_async(stream.cuda_stream)
        b.record(stream)
        stream.synchronize()
        ts.append(time.perf_counter() - t0); evs.append(a.elapsed_time(b))
        batch.build_result()
    ts, evs = sorted(ts[2:]), sorted(evs[2:])
    print(f"M = {M}, {F} frames, shared factor {'on ' if on else 'off'}: wall {ts[len(ts)//2]*1e3:.3f} ms, events {evs[len(evs)//2]*1e3:.0f} us  (took the shared path: {batch.last_build_shared_factor()})", flush=True)
