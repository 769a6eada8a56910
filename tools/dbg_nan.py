import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from facedeform_amd import capi, synth
N, M, F = 1_000_000, 256, 32
dev = torch.device("cuda", 0)
P = synth.head_mesh(N); rest = synth.control_points(M, "head"); P[:8] = rest[:8]
deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(64)]).astype(np.float32)
d_P, d_rest, d_del = (torch.from_numpy(a).to(dev) for a in (P, rest, deltas))
stream = torch.cuda.Stream(device=dev)
engines = []
for _ in range(F):
    e = capi.Engine(); e.set_stream(stream.cuda_stream); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0); engines.append(e)
batch = capi.Batch(engines)
cus = int(sys.argv[1]) if len(sys.argv) > 1 else 224
if cus: batch.set_eval_cus(cus)
dels = [torch.empty_like(d_P) for _ in range(F)]
falls = [torch.zeros(N, device=dev) for _ in range(F)]
for rep in range(3):
  for first in (0, 32):
    batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + (first + k) * M * 12 for k in range(F)], M)
    batch.build_async(stream.cuda_stream)
    for mode in (capi.OUTPUT_POSITION, capi.OUTPUT_DISPLACEMENT):
        for e in engines: e.set_output(mode)
        for o in dels: o.fill_(float("nan"))
        batch.deform_shared_dev(N, d_P.data_ptr(), [o.data_ptr() for o in dels], d_falloff=[f.data_ptr() for f in falls], stream_ptr=stream.cuda_stream)
        torch.cuda.synchronize()
        bad = []
        for k in range(F):
            nanv = torch.isnan(dels[k]).any(dim=1)
            c = int(nanv.sum())
            if c:
                ii = torch.nonzero(nanv).flatten()[:6].tolist()
                bad.append((k, c, ii))
        print(f"rep {rep} first {first} mode {mode}: frames with NaN: {bad}", flush=True)
