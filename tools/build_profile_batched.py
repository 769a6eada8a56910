"""A few batched builds (n contexts, one rest array, own deltas) under rocprofv3 --kernel-trace --stats (thin-plate, linear term).
usage: build_profile_batched.py M n_contexts [builds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from facedeform_amd import capi, synth

M = int(sys.argv[1]); n = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
rest = synth.control_points(M, "head")
deltas = np.stack([synth.rig_deltas(rest, f) for f in range(n)])
d_rest = torch.from_numpy(rest).to(dev); d_deltas = torch.from_numpy(deltas).to(dev)
stream = torch.cuda.Stream(device=dev)
engines = []
for _ in range(n):
    e = capi.Engine(solver=capi.SOLVER_AUTO)
    e.set_stream(stream.cuda_stream); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
    engines.append(e)
batch = capi.Batch(engines)
for r in range(reps):
    batch.set_points_dev([d_rest.data_ptr()] * n, [d_deltas.data_ptr() + f * M * 12 for f in range(n)], M)
    batch.build_async(stream.cuda_stream)
    torch.cuda.synchronize()
assert [r.terminationtype for r in batch.build_result()] == [1] * n
batch.close()
for e in engines:
    e.set_stream(None); e.close()
