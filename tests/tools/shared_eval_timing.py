"""Time the shared-rig evaluation (fd_batch_deform_shared_dev) against the per-frame batched launch
(fd_batch_deform_dev) at C2 / C3 sizes, HIP events on the launch stream, and report each frame's
parity against the oracle on a vertex sample.   python tests/tools/shared_eval_timing.py [c2|c3] [frames,...] [tps|qnn]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    N, M = {"c2": (1_000_000, 256), "c3": (1_000_000, 2048), "c5": (10_000_000, 512)}[cfg]
    frames = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "4,8,16,32".split(","))]
    model = sys.argv[3] if len(sys.argv) > 3 else "tps"
    kind, okind, params = ((capi.KERNEL_THIN_PLATE, fo.KERNEL_THIN_PLATE, []) if model == "tps" else
                           (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN, [1.0, 5.0]))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    P = synth.head_mesh(N); rest = synth.control_points(M, "head")
    d_P = torch.from_numpy(P).to(dev); d_rest = torch.from_numpy(rest).to(dev)
    orc = fo.Oracle()
    idx = np.linspace(0, N - 1, 3000).astype(np.int64)
    for F in frames:
        deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(F)])
        d_del = torch.from_numpy(deltas).to(dev)
        engines = []
        for _ in range(F):
            e = capi.Engine(); e.set_stream(stream.cuda_stream); e.set_kernel(kind, params); e.set_term(0); engines.append(e)
        batch = capi.Batch(engines)
        batch.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr() + f * M * 12 for f in range(F)], M)
        batch.build_async(stream.cuda_stream); batch.build_result()
        outs = [torch.empty_like(d_P) for _ in range(F)]
        falls = [torch.zeros(N, device=dev) for _ in range(F)]
        po, pf = [o.data_ptr() for o in outs], [f.data_ptr() for f in falls]
        res = {}
        for name, call in (("shared", lambda: batch.deform_shared_dev(N, d_P.data_ptr(), po, d_falloff=pf, stream_ptr=stream.cuda_stream)),
                           ("per-frame", lambda: batch.deform_dev(N, [d_P.data_ptr()] * F, po, d_falloff=pf, stream_ptr=stream.cuda_stream))):
            for _ in range(3):
                call()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in evs:
                a.record(stream); call(); b.record(stream)
            stream.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            worst = 0.0
            for f in range(0, F, max(1, F // 4)):
                table = orc.control_table(rest, (rest + deltas[f]).astype(np.float32))
                _, _, W, radii = orc.build(table, okind, params, 0)
                ref, _ = orc.deform(table, okind, radii, W, P[idx])
                out = outs[f].cpu().numpy()[idx]
                worst = max(worst, synth.parity_error(out.astype(np.float64) - P[idx], ref.astype(np.float64) - P[idx]).max())
            res[name] = ts[len(ts) // 2]
            us = ts[len(ts) // 2] * 1e3
            gb = (12.0 * N + F * 16.0 * N) / (us * 1e-6) / 1e9 if name == "shared" else F * 28.0 * N / (us * 1e-6) / 1e9
            print(f"{cfg} {model} N={N} M={M} F={F:2d} {name:9s}: launch {us:9.1f} us = {us / F:7.2f} us/frame  {N / (us / F) :9.0f} Mverts/s  "
                  f"HBM (algorithmic) {gb:7.0f} GB/s  parity(raw, sample) {worst:.2e}", flush=True)
        batch.close()
        for e in engines:
            e.set_stream(None); e.close()


if __name__ == "__main__":
    main()
