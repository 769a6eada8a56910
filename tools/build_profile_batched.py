"""A few batched builds (n contexts, one rest array, own deltas) under rocprofv3 --kernel-trace --stats (thin-plate, linear term).
usage: build_profile_batched.py M n_contexts [builds] [eval_cus]
eval_cus below the device's CU count (e.g. 224): the batch leaves CUs to its builds -> the one-workgroup form of the register build
(k_build_reg<true>, what bench.py's pipeline runs); absent or 0: the parallel front end (k_reg_front1 / k_reg_front2 / k_build_reg<false>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from facedeform_amd import capi, synth

M = int(sys.argv[1]); n = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cus = int(sys.argv[4]) if len(sys.argv) > 4 else 0
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
batch.set_eval_cus(cus)
for r in range(reps):
    batch.set_points_dev([d_rest.data_ptr()] * n, [d_deltas.data_ptr() + f * M * 12 for f in range(n)], M)
    batch.build_async(stream.cuda_stream)
    torch.cuda.synchronize()
assert [r.terminationtype for r in batch.build_result()] == [1] * n
batch.close()
for e in engines:
    e.set_stream(None); e.close()
