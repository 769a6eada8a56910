"""Register-resident build (fd_build_reg.hip) against the oracle and the launch chain: weights, timing.
   python tests/tools/reg_build_check.py            (FD_REG_STAMPS=1 for the phase cycles of the kernel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from facedeform_amd import capi, synth
from oracle import fd_oracle as fo

fo.build(); orc = fo.Oracle()
cases = [(capi.KERNEL_THIN_PLATE, fo.KERNEL_THIN_PLATE, [], 0), (capi.KERNEL_CUBIC, fo.KERNEL_CUBIC, [], 0),
         (capi.KERNEL_BIHARMONIC, fo.KERNEL_BIHARMONIC, [], 1), (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN, [0.35, 0.0], 2),
         (capi.KERNEL_GAUSSIAN, fo.KERNEL_GAUSSIAN, [0.35, 1e-3], 0)]
for M in (16, 20, 32, 37, 100, 240, 250, 256):
    rest = synth.control_points(M, "head"); deform = synth.deformed_rig(rest, 2); delta = (deform - rest).astype(np.float32)
    for kind, okind, params, term in cases:
        table = orc.control_table(rest, deform)
        rc, tt, W, radii = orc.build(table, okind, params, term)
        res = {}
        for name, solver in (("reg", capi.SOLVER_REGISTER), ("chain", capi.SOLVER_CHAIN)):
            e = capi.Engine(solver=solver); e.set_kernel(kind, params); e.set_term(term); e.set_points(rest, delta)
            rep = e.build(); Wg, _ = e.get_weights(); res[name] = (rep.terminationtype, np.abs(Wg - W).max() / np.abs(W).max(), rep.pivot_ratio)
            e.close()
        print(f"M={M:4d} kind={kind} term={term} oracle tt={tt}  reg tt={res['reg'][0]} err={res['reg'][1]:.2e} piv={res['reg'][2]:.2e} | chain tt={res['chain'][0]} err={res['chain'][1]:.2e} piv={res['chain'][2]:.2e}", flush=True)
# timing, C2
M = 256
rest = synth.control_points(M, "head"); delta = synth.smooth_deltas(rest, 0).astype(np.float32)
dev = torch.device("cuda", 0)
d_rest = torch.from_numpy(rest).to(dev); d_del = torch.from_numpy(delta).to(dev)
for name, solver in (("reg", capi.SOLVER_REGISTER), ("chain", capi.SOLVER_CHAIN), ("one-workgroup", capi.SOLVER_ONE_WORKGROUP)):
    e = capi.Engine(solver=solver); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
    ts = []
    for i in range(12):
        e.set_points_dev(d_rest.data_ptr(), d_del.data_ptr(), M); e.synchronize()
        t0 = time.perf_counter(); e.build_async(); e.build_result(); ts.append(time.perf_counter() - t0)
    print(f"single build {name}: median {np.median(ts[2:]) * 1e3:.3f} ms", flush=True)
    e.close()
for name, solver in (("reg", capi.SOLVER_REGISTER), ("one-workgroup", capi.SOLVER_ONE_WORKGROUP), ("chain", capi.SOLVER_CHAIN)):
    F = 32
    engines = []
    for _ in range(F):
        e = capi.Engine(solver=solver); e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR); engines.append(e)
    b = capi.Batch(engines)
    ts = []
    for i in range(8):
        b.set_points_dev([d_rest.data_ptr()] * F, [d_del.data_ptr()] * F, M); torch.cuda.synchronize()
        t0 = time.perf_counter(); b.build_async(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert [r.terminationtype for r in b.build_result()] == [1] * F
    print(f"batched build of 32 {name}: median {np.median(ts[2:]) * 1e3:.3f} ms", flush=True)
    b.close()
    for e in engines: e.close()
