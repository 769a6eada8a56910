"""dist2 producer (fd_capture_dist2_dev) timing: N = 1M mesh points against T rig triangles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facedeform_amd import capi, synth

def main():
    N = 1_000_000
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(N)
    d_P = torch.from_numpy(P).to(dev)
    d_d2 = torch.empty(N, device=dev)
    stream = torch.cuda.Stream(device=dev)
    e = capi.Engine(); e.set_stream(stream.cuda_stream)
    for M in (256, 512, 2048):
        rest = synth.control_points(M, "head")
        tris = []
        for i in range(M):
            d = np.linalg.norm(rest - rest[i], axis=1)
            j, k = np.argsort(d)[1:3]
            tris.append(np.concatenate([rest[i], rest[j], rest[k]]))
        tris = np.array(tris, np.float32)
        d_tri = torch.from_numpy(tris).to(dev)
        torch.cuda.synchronize()
        for r2, label in ((1e30, "no radius"), (0.05, "radius^2 = 0.05")):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
            for a, b in evs:
                a.record(stream)
                e.capture_dist2_dev(N, d_P.data_ptr(), 0, tris.shape[0], d_tri.data_ptr(), r2, True, d_d2.data_ptr())
                b.record(stream)
            stream.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs[2:])
            t = ts[len(ts) // 2]
            # roofline: the kernel is fp32-vector-bound: ~45 flop per point-triangle pair (two dot products, region
            # selection, distance) against 157.3 TFLOP/s; with a radius the wave-level cull skips pairs, so the
            # "no radius" line is the one that prices every pair
            tf = 45.0 * N * tris.shape[0] / (t * 1e-3) / 1e12
            print(f"N={N} T={tris.shape[0]}: {t*1e3:8.1f} us  ({N*tris.shape[0]/t/1e6:7.1f} G point-triangle pairs/s, {tf:6.1f} TFLOP/s = "
                  f"{tf / 157.3:5.3f} of the fp32 vector peak at 45 flop/pair)  [{label}; "
                  f"Fibonacci vertex order: neighbouring lanes are not neighbouring points]", flush=True)
    e.set_stream(None); e.close()

if __name__ == "__main__":
    main()
