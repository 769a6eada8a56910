"""Evaluation of B frames: B single launches vs one batched launch (HIP events), C2 sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from facedeform_amd import capi, synth
N, M = 1_000_000, 256
KIND, KPARAMS = {"thin_plate": (capi.KERNEL_THIN_PLATE, []), "qnn": (capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0])}[os.environ.get("FD_EVAL_KERNEL", "thin_plate")]
dev = torch.device("cuda", 0)
P = synth.head_mesh(N); rest = synth.control_points(M, "head")
d_P = torch.from_numpy(P).to(dev)
stream = torch.cuda.Stream(device=dev)
Bs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 4, 8, 16, 32]
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["single", "batched"]
for B in Bs:
    es = []
    for f in range(B):
        e = capi.Engine(); e.set_kernel(KIND, KPARAMS); e.set_term(0)
        e.set_points(rest, synth.smooth_deltas(rest, f % 8).astype(np.float32)); e.set_stream(stream.cuda_stream); es.append(e)
    b = capi.Batch(es); b.build_async(stream.cuda_stream); b.build_result()
    outs = [torch.empty_like(d_P) for _ in range(B)]
    falls = [torch.zeros(N, device=dev) for _ in range(B)]        # fd_falloff is written too, as in bench.py
    res = {}
    for mode in modes:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for a, c in evs:
            a.record(stream)
            if mode == "single":
                for e, o, fl in zip(es, outs, falls):
                    e.deform_dev(N, d_P.data_ptr(), o.data_ptr(), d_falloff=fl.data_ptr())
            else:
                b.deform_dev(N, [d_P.data_ptr()] * B, [o.data_ptr() for o in outs], d_falloff=[fl.data_ptr() for fl in falls],
                             stream_ptr=stream.cuda_stream)
            c.record(stream)
        stream.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs[2:])
        res[mode] = ts[len(ts) // 2] * 1e3 / B
    tf = (17 * M + 24) * N / 1e12
    print(f"B={B:2d}: per frame " + ", ".join(f"{res[m]:6.1f} us {m} ({tf/res[m]*1e6/157.3*100:4.1f}% fp32)" for m in modes), flush=True)
    b.close()
    for e in es:
        e.set_stream(None); e.close()
