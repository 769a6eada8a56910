"""Time every evaluation-kernel variant on one GPU (HIP events on the launch stream) and
report the parity error of each against the oracle on a vertex sample."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

def main():
    cfgs = [("c2", 1_000_000, 256), ("c3", 1_000_000, 2048), ("c1big", 1_000_000, 32), ("c2pad", 1024 * 1024, 256), ("c2x16", 16_000_000, 256)]
    if len(sys.argv) > 2:
        cfgs = [c for c in cfgs if c[0] in sys.argv[2].split(",")]
    variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,3,11,12,13,101,102,103,111,112,113".split(","))]
    kname = os.environ.get("FD_EVAL_KERNEL", "thin_plate")
    kind_c, kind_o, kparams = {"thin_plate": (capi.KERNEL_THIN_PLATE, fo.KERNEL_THIN_PLATE, []),
                               "qnn": (capi.KERNEL_GAUSSIAN_QNN, fo.KERNEL_GAUSSIAN_QNN, [1.0, 5.0]),
                               "gaussian": (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN, [0.5]),
                               "biharmonic": (capi.KERNEL_BIHARMONIC, fo.KERNEL_BIHARMONIC, []),
                               "cubic": (capi.KERNEL_CUBIC, fo.KERNEL_CUBIC, [])}[kname]
    orc = fo.Oracle()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for name, N, M in cfgs:
        P = synth.head_mesh(N)
        rest = synth.control_points(M, "head")
        deform = synth.deformed_rig(rest)
        d_P = torch.from_numpy(P).to(dev); d_out = torch.empty_like(d_P)
        d_fall = torch.zeros(N, device=dev)
        table = orc.control_table(rest, deform)
        rc, tt, W, radii = orc.build(table, kind_o, kparams, 0)
        idx = np.linspace(0, N - 1, 4000).astype(np.int64)
        ref, _ = orc.deform(table, kind_o, radii, W, P[idx])
        for var in variants:
            e = capi.Engine(variant=var)
            e.set_stream(stream.cuda_stream)
            e.set_points(rest, (deform - rest).astype(np.float32)); e.set_kernel(kind_c, kparams); e.set_term(0)
            rep = e.build()
            for _ in range(3):
                e.deform_dev(N, d_P.data_ptr(), d_out.data_ptr(), d_falloff=d_fall.data_ptr())
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for a, b in evs:
                a.record(stream); e.deform_dev(N, d_P.data_ptr(), d_out.data_ptr(), d_falloff=d_fall.data_ptr()); b.record(stream)
            stream.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            out = d_out.cpu().numpy()[idx]
            d = out.astype(np.float64) - P[idx]; dr = ref.astype(np.float64) - P[idx]
            err = synth.parity_error(d, dr).max()
            us = ts[len(ts) // 2] * 1e3
            tf = ((17 if kname in ("thin_plate", "cubic") else 16) * M + 24) * N / (us * 1e-6) / 1e12
            print(f"{kname} {name} N={N} M={M} variant {var:4d}: median {us:8.1f} us  min {ts[0]*1e3:8.1f} us  {tf:6.1f} TFLOP/s ({tf/157.3*100:4.1f}% fp32)  "
                  f"build {rep.t_assemble_ms + rep.t_solve_ms:.3f} ms  parity(raw) {err:.2e}", flush=True)
            e.set_stream(None); e.close()

if __name__ == "__main__":
    main()
