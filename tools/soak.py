"""Leak / stability soak: create, use and destroy engines, batches, morph objects many times and
watch free device memory; then a long pipelined run."""
import os, sys, time, gc
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from facedeform_amd import capi, synth
from facedeform_amd.sop import FaceDeformSOP

def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20

def main():
    dev = torch.device("cuda", 0)
    P = synth.head_mesh(200_000); rest = synth.control_points(128, "head")
    shapes = [(P + np.float32(0.01 * (i + 1))).astype(np.float32) for i in range(6)]
    d_P = torch.from_numpy(P).to(dev); d_out = torch.empty_like(d_P)
    base = None
    for rnd in range(6):
        for it in range(40):
            es = [capi.Engine() for _ in range(4)]
            for k, e in enumerate(es):
                e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(0)
                e.set_points(rest, synth.smooth_deltas(rest, k).astype(np.float32))
            b = capi.Batch(es); b.build_async(); b.build_result()
            es[0].deform_dev(P.shape[0], d_P.data_ptr(), d_out.data_ptr()); es[0].synchronize()
            es[1].set_deltas(synth.smooth_deltas(rest, 5).astype(np.float32)); es[1].build()
            es[2].mesh_set(P); out = np.empty_like(P); es[2].deform_mesh(out)
            es[3].capture_dist2(P[:1000], np.zeros((3, 9), np.float32), 1.0)
            if it % 4 == 0:      # the multilayer model (record arrays grow), the LU path, an exported blob
                ml = capi.Engine(); ml.set_kernel(capi.KERNEL_GAUSSIAN_ML, [0.8, 2 + it % 5, 0.1]); ml.set_term(0)
                ml.set_points(rest, synth.smooth_deltas(rest, 1).astype(np.float32)); ml.build()
                r2 = capi.Engine(solver=capi.SOLVER_LU); r2.import_model(ml.export_model()); r2.deform(P[:5000]); r2.close(); ml.close()
            b.close()
            for e in es: e.close()
            if it % 10 == 0:
                m = capi.Morph(); m.init(P, shapes); m.apply(P); m.close()
                node = FaceDeformSOP(); node.set("kernel", 1); node.cook(P, rest, synth.deformed_rig(rest, 1)); del node
        gc.collect()
        f = free_mb()
        base = base if base is not None else f
        print(f"round {rnd}: free device memory {f:.0f} MiB (delta vs round 0: {f - base:+.0f} MiB)", flush=True)
    assert abs(free_mb() - base) < 64, "device memory is leaking"
    print("no leak", flush=True)

if __name__ == "__main__":
    main()
