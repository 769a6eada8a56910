"""A/B of the build variants of the 32-row shared-rig kernel INSIDE one process (same buffers, same box, interleaved):
   python tests/tools/wide_variants_timing.py [variants, e.g. 0,1,2,3,16] [rounds]      (16 = the 16-row kernel)
Launch-to-launch and process-to-process spread of this kernel is ~10 %, more than most variants differ by."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from facedeform_amd import capi, synth


def main():
    variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,16").split(",")]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    N, M, F = 1_000_000, 256, 32
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    P = synth.head_mesh(N); rest = synth.control_points(M, "head")
    d_P = torch.from_numpy(P).to(dev); d_rest = torch.from_numpy(rest).to(dev)
    deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(F)])
    d_del = torch.from_numpy(deltas).to(dev)
    engines = []
    for _ in range(F):
        e = capi.Engine(); e.set_stream(stream.cuda_stream); e.set_kernel(capi.KERNEL_THIN_PLATE, []); e.set_term(0); engines.append(e)
    batch = capi.Batch(engines)
    batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + f * M * 12 for f in range(F)], M)
    batch.build_async(stream.cuda_stream); batch.build_result()
    outs = [torch.empty_like(d_P) for _ in range(F)]
    falls = [torch.zeros(N, device=dev) for _ in range(F)]
    po, pf = [o.data_ptr() for o in outs], [f.data_ptr() for f in falls]
    times = {v: [] for v in variants}
    for r in range(rounds + 1):
        for v in variants:
            if v == 16:
                os.environ["FD_SHARED_WIDE"] = "0"
            else:
                os.environ["FD_SHARED_WIDE"] = "1"; os.environ["FD_SHARED_WIDE_VAR"] = str(v)
            batch.deform_shared_dev(N, d_P.data_ptr(), po, d_falloff=pf, stream_ptr=stream.cuda_stream)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                for _ in range(10):
                    batch.deform_shared_dev(N, d_P.data_ptr(), po, d_falloff=pf, stream_ptr=stream.cuda_stream)
                e1.record(stream)
            e1.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) * 100.0)     # us per launch (pack included)
    for v in variants:
        t = np.array(times[v])
        print(f"variant {v:2d}: median {np.median(t):7.1f} us   min {t.min():7.1f}   max {t.max():7.1f}   ({len(t)} x 10 launches)")


if __name__ == "__main__":
    main()
