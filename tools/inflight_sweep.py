"""Where does a pipelined cook's time go?  Sweep frames in flight, with and without the per-step
HIP events, and report host enqueue time beside the wall time per step (C2 by default)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facedeform_amd import capi, synth

def main():
    N, M = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 256
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["full", "build", "eval"]
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(N); rest = synth.control_points(M, "head")
    deltas = np.stack([synth.smooth_deltas(rest, f) for f in range(8)])
    d_P = torch.from_numpy(P).to(dev); d_rest = torch.from_numpy(rest).to(dev); d_deltas = torch.from_numpy(deltas).to(dev)
    for nin in (1, 2, 4, 8, 16, 32):
        lanes = []
        for _ in range(nin):
            e = capi.Engine(device=0); s = torch.cuda.Stream(device=dev)
            e.set_stream(s.cuda_stream); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
            lanes.append((e, s, torch.empty_like(d_P), torch.zeros(N, device=dev)))
        for mode in modes:
            for with_events in (False, True):
                def step(i, ev=None):
                    e, s, out, fall = lanes[i % nin]
                    if mode != "eval" or i < nin:
                        e.set_points_dev(d_rest.data_ptr(), d_deltas.data_ptr() + (i % 8) * M * 12, M)
                        if ev: ev[0].record(s)
                        e.build_async()
                        if ev: ev[1].record(s)
                    if mode != "build":
                        if ev: ev[2].record(s)
                        e.deform_dev(N, d_P.data_ptr(), out.data_ptr(), d_falloff=fall.data_ptr())
                        if ev: ev[3].record(s)
                for i in range(2 * nin): step(i)
                torch.cuda.synchronize()
                K = 200
                evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)] if with_events else [None] * K
                t0 = time.perf_counter()
                for i in range(K): step(2 * nin + i, evs[i])
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                print(f"M={M} inflight {nin:2d} {mode:5s} events={int(with_events)}: wall {1e3*(t2-t0)/K:7.4f} ms/step   host enqueue {1e3*(t1-t0)/K:7.4f} ms/step", flush=True)
        for e, s, _, _ in lanes:
            e.set_stream(None); e.close()

if __name__ == "__main__":
    main()
