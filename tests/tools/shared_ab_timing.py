"""A/B timing of the shared-rig evaluation launch INSIDE ONE PROCESS (box-to-box and process-to-process spread of this launch is
+-10 %, more than most variants differ by): the variants are environment settings a tuning build (-DFD_TUNING) reads on every
launch; they are launched in turn, round after round, and the median / minimum per variant are reported.

    python tests/tools/shared_ab_timing.py [c2|c3] [frames] [rounds] VAR=VALUE[,VAR=VALUE...] VAR=VALUE ...
    e.g.  python tests/tools/shared_ab_timing.py c2 32 40 FD_SHARED_W1=1 FD_SHARED_W1=0"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from facedeform_amd import capi, synth


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    N, M = {"c2": (1_000_000, 256), "c3": (1_000_000, 2048), "c5": (1_250_000, 512)}[cfg]
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    variants = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[4:]] or [{}]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    P = synth.head_mesh(N); rest = synth.control_points(M, "head")
    d_P = torch.from_numpy(P).to(dev); d_rest = torch.from_numpy(rest).to(dev)
    deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(F)])
    d_del = torch.from_numpy(deltas).to(dev)
    engines = []
    for _ in range(F):
        e = capi.Engine(); e.set_stream(stream.cuda_stream); e.set_term(0); engines.append(e)
        if os.environ.get("FD_AB_MODEL", "tps") == "qnn":
            e.set_kernel(capi.KERNEL_GAUSSIAN_QNN, [1.0, 5.0])
        else:
            e.set_kernel(capi.KERNEL_THIN_PLATE)
    batch = capi.Batch(engines)
    batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + f * M * 12 for f in range(F)], M)
    batch.build_async(stream.cuda_stream); batch.build_result()
    outs = [torch.empty_like(d_P) for _ in range(F)]
    falls = [torch.zeros(N, device=dev) for _ in range(F)]
    po, pf = [o.data_ptr() for o in outs], [f.data_ptr() for f in falls]
    keys = sorted({k for v in variants for k in v})

    def launch(v):
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(v)
        # pack and evaluation as two calls, so that the event pair holds the evaluation launch alone
        batch.prepare_shared(po, d_falloff=pf, stream_ptr=stream.cuda_stream)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        batch.deform_shared_dev(N, d_P.data_ptr(), po, d_falloff=pf, stream_ptr=stream.cuda_stream)
        b.record(stream)
        return a, b

    times = [[] for _ in variants]
    ref = []
    for r in range(rounds + 2):
        evs = [[launch(v) for _ in range(3)] for v in variants]
        stream.synchronize()
        if r < 2:
            if r == 1:
                ref = None
            continue
        for i, ev in enumerate(evs):
            times[i] += [a.elapsed_time(b) * 1e3 for a, b in ev]
    for v, t in zip(variants, times):
        t = np.array(t)
        print(f"{cfg} F={F:2d} {str(v):40s}: median {np.median(t):7.1f} us  min {t.min():7.1f}  p10 {np.percentile(t, 10):7.1f}  p90 {np.percentile(t, 90):7.1f}  "
              f"({t.size} launches)  HBM(alg) {(12.0 + 16.0 * F) * N / np.median(t) / 1e3:6.0f} GB/s", flush=True)
    batch.close()
    for e in engines:
        e.set_stream(None); e.close()


if __name__ == "__main__":
    main()
