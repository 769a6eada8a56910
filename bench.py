#!/usr/bin/env python3
"""Benchmark of the RBF deformation hot path on MI355X.

One "step" = one cook of the hot path on one frame, inputs resident in HBM:
control table upload (device-to-device) + kernel-matrix assembly + dense fp64
solve + per-vertex evaluation with the reference's epilogue.  The model is
rebuilt every step, as the reference does every cook.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: independent frames are sharded one per GPU at a time (BASELINE config 4),
no data-path collective, weak scaling.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

# Independent frames run on separate HIP streams; the runtime maps streams onto this many
# hardware queues (default 4, which lets only ~2 streams overlap).  Must be set before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (n_verts, n_ctrl, mesh, description)
    "c1": (10_000, 32, "sphere", "C1: 10k-vert sphere, 32 control points"),
    "c2": (1_000_000, 256, "head", "C2: 1M-vert head mesh, 256 control points"),
    "c3": (1_000_000, 2048, "head", "C3: 1M-vert head mesh, 2048 control points (solve-bound)"),
    # one mesh split into vertex ranges over the ranks, model broadcast once per step
    "c5": (10_000_000, 512, "head", "C5: 10M-vert mesh, 512 control points, vertex ranges split across the GPUs"),
}
METRIC = "deformed Mverts/sec at 256 ctrl pts, 1/2/4/8 MI355X vs host-CPU ref"
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA dense peak
PEAK_FP16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA
PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X_MICROARCH.md: dense fp64 MFMA
PEAK_HBM_GBS = 8000.0      # HBM3E spec
FLOPS_PER_PAIR = 17        # thin-plate: SURVEY.md 8d
FLOPS_PER_VERTEX_AFFINE = 24
BYTES_PER_VERTEX = 24      # read P 12 + write P 12 (the conservative figure)
N_FRAMES = 64


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--variant", type=int, default=0, help="evaluation kernel variant (0 = auto)")
    ap.add_argument("--precision", choices=["fp32", "fp64"], default="fp32")
    ap.add_argument("--inflight", type=int, default=32,
                    help="frames per batched build (one engine context each; at most 32)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="groups of frames in flight per GPU (one stream + one fd_batch each)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="evaluations per HIP event pair (a pair costs ~4-5 us of stream time, spread over the run)")
    ap.add_argument("--eval-launch", choices=["shared", "batched", "single"], default="shared",
                    help="shared: the frames of a group share the mesh and the rest rig (they do, in every BASELINE "
                         "configuration): phi once per (vertex, centre), weight contraction on the matrix pipe "
                         "(fd_batch_deform_shared_dev); batched: one launch, every frame evaluated on its own "
                         "(fd_batch_deform_dev); single: one launch per frame")
    ap.add_argument("--eval-cus", type=int, default=0,
                    help="shared evaluation: CUs it occupies (one persistent workgroup each, all of a CU's LDS); the rest stay "
                         "free for the builds of the next groups, which otherwise only run in the gaps between evaluation "
                         "launches.  0 = the measured best for the build style: 192 with --build one-workgroup (its 3 x 32 persistent "
                         "workgroups want 64 CUs to themselves; 120-122k Mverts/s, 208: 110k), 224 with --build chain (113k; 256: 101k, 192: 109k)")
    ap.add_argument("--build", choices=["register", "one-workgroup", "chain"], default="register",
                    help="batched builds of the frame pipeline (config c2): one-workgroup = fd_config.solver FD_SOLVER_ONE_WORKGROUP "
                         "(one launch, one workgroup per model: 32 workgroups on 32 CUs per batch, nothing else on the device -- "
                         "the evaluation keeps the other CUs undisturbed; 120-122k Mverts/s); chain = the default solver's launch "
                         "chain (faster for a lone build; 112k in this pipeline).  The single-cook latency is measured with the "
                         "default solver either way")
    ap.add_argument("--cu-split", choices=["mask", "none"], default="none",
                    help="mask: the evaluation stream and the build streams are created with complementary CU masks "
                         "(hipExtStreamCreateWithCUMask): --eval-cus CUs for the evaluation, the others for the builds.  Measured "
                         "worse (68k against 112k Mverts/s: confined to 32 CUs a batched build takes 1.37 ms instead of 0.8 -- "
                         "unconfined, the builds also use the evaluation's CUs between its launches); kept for the record")
    ap.add_argument("--c5-group", type=int, default=32,
                    help="config c5: frames per group (one batched build, one broadcast, one evaluation launch; 8: 55k, 16: 86k, 32: 97k "
                         "Mverts/s on one GPU -- at 32 the 512-centre model no longer fits the LDS in one piece, but the 32-row kernel "
                         "it selects makes up for the staging)")
    ap.add_argument("--c5-solve", choices=["broadcast", "redundant"], default="broadcast",
                    help="config c5: rank 0 solves and broadcasts the models (default), or every rank solves them itself")
    ap.add_argument("--eval-stream", choices=["shared", "lane", "per-lane"], default="shared",
                    help="evaluations on one stream for all lanes (default), on each lane's build stream (the same: 158 k), or on an "
                         "evaluation stream per lane (worse, 147 k: two persistent launches then share the CUs and each takes twice as long)")
    ap.add_argument("--group-call", choices=["c", "python"], default="c",
                    help="shared evaluation: a group is enqueued by ONE foreign call (fd_batch_cook_group, default) or by the five "
                         "fd_batch_* calls with their pointer tables from Python; the line reports the host time per group")
    ap.add_argument("--time-every", type=int, default=4,
                    help="shared evaluation, one call per group: every n-th group carries the four HIP events that time its build and "
                         "its evaluation launch (runs of fewer than 16 groups time every group)")
    ap.add_argument("--no-shared-factor-alternative", dest="shared_factor_alternative", action="store_false",
                    help="skip the second timed pass with one factorisation per group (reported under \"alternative\", never as `value`)")
    ap.add_argument("--alt-eval-cus", type=int, default=224,
                    help="CUs of the shared-rig evaluation in the shared-factor pass (its builds hold one CU for the factorisation and "
                         "31 briefly for the other frames; measured 216 / 224 / 228 / 232 / 236: 168 / 166 / 167 / 149 / 150 k Mverts/s -- the "
                         "pass is bound by the evaluation launch, 0.19 ms per 32 frames on 224 CUs, and the CU-budget cliff is where it was)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=float, default=3.0e8,
                    help="bound on (vertex, centre) pairs in the CPU baseline sample")
    return ap.parse_args()


def cpu_baseline(cfg_name, P, rest, deform, max_pairs):
    """Time the oracle (CPU restatement, kind 'port') on this host.  Test infrastructure
    used as the timed baseline only; never part of the GPU path."""
    from oracle import fd_oracle as fo
    so = None
    try:
        so = fo.build_native(os.path.join(tempfile.mkdtemp(prefix="fdo_"), "libfd_oracle_native.so"))
    except Exception:
        fo.build()
    orc = fo.Oracle(so)
    M = rest.shape[0]
    n_sample = int(min(P.shape[0], max(1000, max_pairs // M)))
    Ps = np.ascontiguousarray(P[:n_sample])
    table = orc.control_table(rest, deform)
    t0 = time.perf_counter()
    rc, tt, W, radii = orc.build(table, fo.KERNEL_THIN_PLATE, [], fo.TERM_LINEAR)
    t_build = time.perf_counter() - t0
    assert tt == 1
    out = np.empty_like(Ps)
    t0 = time.perf_counter()
    orc.deform(table, fo.KERNEL_THIN_PLATE, radii, W, Ps, nthreads=1, out=out)
    t_eval1 = time.perf_counter() - t0
    ncores = os.cpu_count() or 1
    t0 = time.perf_counter()
    orc.deform(table, fo.KERNEL_THIN_PLATE, radii, W, Ps, nthreads=ncores, out=out)
    t_evaln = time.perf_counter() - t0
    # whole cook on the full mesh = one solve + evaluation scaled from the sample
    scale = P.shape[0] / n_sample
    cook1 = t_build + t_eval1 * scale
    cookn = t_build + t_evaln * scale
    sample = (f"{n_sample} of {P.shape[0]} vertices x {M} centres, fp64 dense restatement "
              f"(oracle/fd_oracle.c, {'-O3 -march=native' if so else '-O2 portable'}), "
              f"1 solve ({t_build * 1e3:.1f} ms) + single-thread evaluation ({t_eval1:.2f} s on the sample)")
    return {
        "value": P.shape[0] / cook1 / 1e6, "unit": "Mverts/s", "cores": 1, "kind": "port", "sample": sample,
        "all_cores": {"value": P.shape[0] / cookn / 1e6, "unit": "Mverts/s", "cores": ncores,
                      "eval_s_on_sample": t_evaln},
    }


def run_split_mesh(args, torch, dist, capi, synth, rank, world, local_rank, dev, rehearse, ranks):
    """BASELINE config 5: ONE mesh, contiguous page-aligned vertex ranges per rank.  Frames are
    cooked in groups: rank 0 assembles + solves the group's models with ONE batched build and
    exports them into one device buffer, ONE broadcast per group (RCCL over xGMI with the nccl
    backend), every rank imports the models and evaluates its own vertex range for all frames of
    the group with one launch.  Two lanes alternate, so rank 0's next build overlaps the current
    evaluations.  The mesh is resident on every rank; only ~28 KB per frame move.  Strong scaling.
    Both of SURVEY 8(e)'s alternatives are timed, one after the other: the broadcast, and every
    rank solving the (small) systems itself with no collective on the data path."""
    from facedeform_amd import dist as fdist
    n_verts, n_ctrl, mesh_kind, desc = CONFIGS["c5"]
    lo, hi = fdist.vertex_range(n_verts, rank, world)
    n_mine = hi - lo
    B = max(1, min(args.inflight, args.c5_group))     # frames per group (a 10M-vertex frame is 160 MB of outputs; two lanes)
    n_lanes = 2
    P_host = synth.head_mesh(n_verts)
    rest_host = synth.control_points(n_ctrl, mesh_kind)
    deltas_host = np.stack([synth.smooth_deltas(rest_host, f) for f in range(N_FRAMES)])
    d_P = torch.from_numpy(P_host[lo:hi]).to(dev)
    del P_host
    d_rest = torch.from_numpy(rest_host).to(dev)
    d_deltas = torch.from_numpy(deltas_host).to(dev)
    delta_stride = n_ctrl * 3 * 4
    lanes = []
    nbytes = None
    for _ in range(n_lanes):
        stream = torch.cuda.Stream(device=dev)
        engines = []
        for _ in range(B):
            e = capi.Engine(device=local_rank, variant=args.variant)
            e.set_stream(stream.cuda_stream)
            e.set_kernel(capi.KERNEL_THIN_PLATE); e.set_term(capi.TERM_LINEAR)
            e.set_points_dev(d_rest.data_ptr(), d_deltas.data_ptr(), n_ctrl)
            engines.append(e)
        batch = capi.Batch(engines)
        batch.set_eval_cus(args.eval_cus)
        batch.build_async(stream.cuda_stream); batch.build_result()
        nbytes = engines[0].model_bytes()
        lanes.append({"stream": stream, "engines": engines, "batch": batch,
                      "blob": torch.zeros((B, nbytes), dtype=torch.uint8, device=dev),
                      "blob_host": torch.zeros((B, nbytes), dtype=torch.uint8).pin_memory() if rehearse else None,
                      "out": [torch.empty_like(d_P) for _ in range(B)],
                      "fall": [torch.zeros(max(n_mine, 1), device=dev, dtype=torch.float32) for _ in range(B)]})
    n_groups = (args.steps + B - 1) // B

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def one_mode(redundant):
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(n_groups)]

        def group(g, first, e=None):
            """Frames first .. first + B - 1 (always a full group: the last one may cook spare frames)."""
            ln = lanes[g % n_lanes]
            stream, engines, batch, blob = ln["stream"], ln["engines"], ln["batch"], ln["blob"]
            frames = [(first + k) % N_FRAMES for k in range(B)]
            with torch.cuda.stream(stream):
                if e: e[0].record(stream)
                if rank == 0 or redundant:
                    batch.set_points_dev([d_rest.data_ptr()] * B, [d_deltas.data_ptr() + f * delta_stride for f in frames], n_ctrl)
                    batch.build_async(stream.cuda_stream)
                    if world > 1 and not redundant:
                        for k, eng in enumerate(engines):
                            eng.export_model_dev(blob[k].data_ptr(), nbytes)      # waits for the build status
                if e: e[1].record(stream)
                if world > 1 and not redundant:
                    if rehearse:                                            # gloo: through host memory
                        if rank == 0:
                            ln["blob_host"].copy_(blob, non_blocking=False)
                        dist.broadcast(ln["blob_host"], src=0)
                        if rank != 0:
                            blob.copy_(ln["blob_host"], non_blocking=False)
                    else:
                        dist.broadcast(blob, src=0)
                    if rank != 0:
                        for k, eng in enumerate(engines):
                            eng.import_model_dev(blob[k].data_ptr(), nbytes, n_ctrl)
                if e: e[2].record(stream)
                if n_mine > 0 and args.eval_launch == "shared" and B > 1:
                    # the frames of a group share the mesh and the rest rig (imported models carry the rig's identity):
                    # phi once per (vertex, centre) for all of them, on this rank's vertex range
                    batch.deform_shared_dev(n_mine, d_P.data_ptr(), [o.data_ptr() for o in ln["out"]],
                                            d_falloff=[f.data_ptr() for f in ln["fall"]], stream_ptr=stream.cuda_stream)
                elif n_mine > 0:
                    batch.deform_dev(n_mine, [d_P.data_ptr()] * B, [o.data_ptr() for o in ln["out"]],
                                     d_falloff=[f.data_ptr() for f in ln["fall"]], stream_ptr=stream.cuda_stream)
                if e: e[3].record(stream)

        if redundant and rank != 0:
            for ln in lanes:                       # contexts that only imported so far get control points of their own
                for eng in ln["engines"]:
                    eng.set_points_dev(d_rest.data_ptr(), d_deltas.data_ptr(), n_ctrl)
        for g in range((args.warmup + B - 1) // B + n_lanes):
            group(g, g * B)
        sync_all()
        t0 = time.perf_counter()
        for g in range(n_groups):
            group(g, g * B, ev[g])
        sync_all()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return {"elapsed": float(t.item()),
                "build_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in ev])),
                "bcast_ms": float(np.mean([e[1].elapsed_time(e[2]) for e in ev])),
                "eval_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))}

    order = ["broadcast", "redundant"] if args.c5_solve == "broadcast" else ["redundant", "broadcast"]
    res = {m: one_mode(m == "redundant") for m in order}
    main_r, alt_r = res[order[0]], res[order[1]]
    redundant = order[0] == "redundant"
    frames_done = n_groups * B                    # >= args.steps; the surplus is work done, not credited
    if rank == 0:
        flops = (FLOPS_PER_PAIR * n_ctrl + FLOPS_PER_VERTEX_AFFINE) * n_mine * B
        tf = flops / (main_r["eval_ms"] * 1e-3) / 1e12
        mv = lambda r: args.steps * n_verts / r["elapsed"] / 1e6
        print(json.dumps({
            "metric": "deformed Mverts/sec, one 10M-vert mesh at 512 ctrl pts split across the GPUs",
            "value": mv(main_r), "unit": "Mverts/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_r["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc + ", thin-plate kernel, linear term, " +
                                   ("the frames of a group share mesh and rest rig: phi once per (vertex, centre), weight contraction on the "
                                    "fp16 matrix pipe (fp16 x 2 split, fp32 accumulate), " if args.eval_launch == "shared" and B > 1 else
                                    "d2 on fp16x2-split MFMA with fp32 accumulation, ") + "models rebuilt " +
                                   ("on every rank every step " if redundant else "on rank 0 every step ") +
                                   f"({B} per batched build)" + ("" if redundant else f", one broadcast per {B} frames"),
                       "n_verts": n_verts,
                       "n_ctrl": n_ctrl, "verts_on_rank0": n_mine, "model_blob_bytes": nbytes,
                       "frames_per_group": B, "frames_cooked": frames_done,
                       "parallelism": (f"vertex ranges over {world} GPU(s), every rank solves the models itself, {n_lanes} lanes" if redundant else
                                       f"vertex ranges over {world} GPU(s), 1 broadcast per {B} frames, {n_lanes} lanes")},
            "ranks": ranks,
            "roofline": c5_roofline(args, B, n_ctrl, n_mine, main_r["eval_ms"], flops, tf),
            "phases_ms": {"build_group_rank0": main_r["build_ms"], "broadcast_and_import": main_r["bcast_ms"],
                          "evaluate_group": main_r["eval_ms"]},
            # the other way of getting the model to every rank, same run, same sizes
            "alternative": {"c5_solve": order[1], "value": mv(alt_r), "ms_per_step": alt_r["elapsed"] / args.steps * 1e3,
                            "phases_ms": {"build_group": alt_r["build_ms"], "broadcast_and_import": alt_r["bcast_ms"],
                                          "evaluate_group": alt_r["eval_ms"]}},
        }), flush=True)
    torch.cuda.synchronize()
    for ln in lanes:
        ln["batch"].close()
        for e in ln["engines"]:
            e.set_stream(None)
            e.close()
    if world > 1:
        dist.destroy_process_group()


def c5_roofline(args, B, n_ctrl, n_mine, eval_ms, flops_frames, tf_frames):
    """Roofline block of the C5 line: the shared-rig launch on this rank's vertex range (bound by the roofline model at
    its algorithmic intensity, as in the C2 line), or the one-frame kernels when asked for."""
    if args.eval_launch == "shared" and B > 1:
        secs = eval_ms * 1e-3
        fl = ((8 + 6 * B) * n_ctrl + FLOPS_PER_VERTEX_AFFINE * B) * n_mine
        by = (12 + 16 * B) * n_mine
        ridge = PEAK_FP16_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        hbm = {"achieved": by / secs / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": by / secs / 1e9 / PEAK_HBM_GBS, "bytes_per_launch": by}
        mfma = {"achieved": fl / secs / 1e12, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": fl / secs / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                "flops_per_launch": fl}
        first = hbm if fl / by < ridge else mfma
        from facedeform_amd import capi
        return {"bound": "hbm" if fl / by < ridge else "mfma", "kernel": capi.load().fd_shared_kernel_name(n_ctrl, B, capi.KERNEL_THIN_PLATE).decode(),
                "achieved": first["achieved"],
                "peak": first["peak"], "unit": first["unit"], "frac": first["frac"], "traffic": None, "avg_launch_ms": eval_ms,
                "frames_per_launch": B, "intensity_flop_per_byte": fl / by, "ridge_flop_per_byte": ridge, "hbm": hbm, "mfma": mfma}
    return {"bound": "valu_fp32", "kernel": "k_deform32_tps_mfma_batch" if B > 1 else "k_deform32_tps_mfma",
            "achieved": tf_frames, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf_frames / PEAK_FP32_TFLOPS, "traffic": None,
            "flops_per_launch": flops_frames, "avg_launch_ms": eval_ms, "frames_per_launch": B}


def masked_streams(torch, dev, n_build_streams, eval_cus, total_cus=256):
    """One evaluation stream and n build streams on complementary CU masks (hipExtStreamCreateWithCUMask), wrapped as
    torch ExternalStreams.  The build CUs are spread evenly whichever way the mask's bits map onto the XCDs (bit i of
    block i // 32 with i % 8 == block: four per XCD under an XCD-major as under an XCD-interleaved numbering, 32 in all;
    more or fewer are taken in the same pattern).  Returns None where the runtime lacks the call."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        create = hip.hipExtStreamCreateWithCUMask
    except (OSError, AttributeError):
        return None
    create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    create.restype = ctypes.c_int
    n_build = total_cus - eval_cus
    if n_build <= 0 or n_build >= total_cus:
        return None
    order = sorted(range(total_cus), key=lambda i: ((i % 8 - i // 32) % 8, i // 8 % 4, i))     # the pattern above first
    build_bits = set(order[:n_build])
    words = total_cus // 32

    def mask(bits):
        arr = (ctypes.c_uint32 * words)()
        for i in bits:
            arr[i // 32] |= 1 << (i % 32)
        return arr

    def make(bits):
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = create(ctypes.byref(h), words, mask(bits))
        if rc != 0 or not h.value:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
        return torch.cuda.ExternalStream(h.value, device=dev)

    try:
        ev = make(set(range(total_cus)) - build_bits)
        builds = [make(build_bits) for _ in range(n_build_streams)]
    except RuntimeError:
        return None
    return ev, builds


class RawEvents:
    """Four HIP events of a group, as raw handles: recorded inside fd_batch_cook_group (build start / end on the build
    stream, evaluation launch start / end on the evaluation stream), read with hipEventElapsedTime."""
    _hip = None

    def __init__(self, capi, after_build=True, before_build=True):
        import ctypes
        if RawEvents._hip is None:
            RawEvents._hip = ctypes.CDLL("libamdhip64.so")
            RawEvents._hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
            RawEvents._hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
            RawEvents._hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
            RawEvents._hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.h = []
        for _ in range(4):
            e = ctypes.c_void_p()
            if RawEvents._hip.hipEventCreate(ctypes.byref(e)) != 0:
                raise RuntimeError("hipEventCreate failed")
            self.h.append(e)
        # after_build=False: no event between the builds and the packing kernel (NULL members are skipped by the library): the
        # "build" of such a group is then timed to the event in front of the evaluation and includes the packing kernel
        # before_build=False: no event in front of the group's first kernel either (after a device-wide wait the first packets of a
        # queue are its slowest: 20-50 us of host time for the group's first two records and launch against 9 warm) -- such a group
        # times its evaluation launch only; its build is timed on an untimed repeat
        self.build_end = 1 if after_build else 2
        self.has_build = before_build
        self.struct = capi.FdGroupEvents(*[(e.value if ((q != 1 or after_build) and (q != 0 or before_build)) else None) for q, e in enumerate(self.h)])

    def warm(self, stream_ptr):
        """One record of every event before the timed region: the runtime sets an event's signal up on its first record, which
        would otherwise happen inside the timed group's one foreign call."""
        for e in self.h:
            RawEvents._hip.hipEventRecord(e, stream_ptr)

    def ms(self, a, b):
        import ctypes
        out = ctypes.c_float()
        rc = RawEvents._hip.hipEventElapsedTime(ctypes.byref(out), self.h[a], self.h[b])
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return float(out.value)

    def close(self):
        for e in self.h:
            RawEvents._hip.hipEventDestroy(e)


def host_cook_ms(capi, synth, P_host, rest_host, device):
    """the cook end to end INCLUDING the host link, as the Houdini node lives it -- fdsop_cook (the cookMySop
    mirror: parm read, control table, build, evaluation, write-back) on page-locked mesh arrays as hdk/SOP_FaceDeformHip.cpp keeps
    them, thin-plate kernel, the node's other parms at their defaults.  Two figures: everything new every cook (mesh upload 12 MB,
    model rebuilt, 16 MB back), and the animated-shot case (mesh and rest rig unchanged: deltas only, results back).  Not `value`."""
    from facedeform_amd.sop import FaceDeformSOP
    n = P_host.shape[0]
    pin_in = capi.host_array((n, 3)); pin_in[:] = P_host
    pin_out = capi.host_array((n, 3)); pin_fall = capi.host_array(n)
    node = FaceDeformSOP(device=device)
    node.set("kernel", 1)                       # thin-plate (the wrapper's added ordinal; BASELINE's kernel)
    out = {}
    for key, kw in (("rebuild", {}), ("mesh_and_rest_rig_unchanged", {"rig_rest_unchanged": True, "mesh_unchanged": True})):
        ts = []
        for f in range(12):
            deform = synth.deformed_rig(rest_host, f)
            t0 = time.perf_counter()
            res = node.cook(pin_in, rest_host, deform, out_P=pin_out, out_falloff=pin_fall, want_Cd=False, **kw)
            ts.append(time.perf_counter() - t0)
            if res.severity >= 3:              # FDSOP_ERROR
                raise SystemExit(f"fdsop_cook failed: {res.errors}")
        ts = sorted(ts[2:])
        out[key] = ts[len(ts) // 2] * 1e3
    node.close()
    return out


def shared_rows(frames, kernel):
    """Rows of the weight operand the shared-rig launch runs for `frames` frames (mirrors launch_deform_shared in
    csrc/fd_eval_shared.hip).  17..32 frames (both 32-row kernels): row 3 f + c of a stack of 32-row tiles for nslot = 4 ceil(F / 4)
    frame slots, i.e. 32 ceil(3 nslot / 32) rows -- 64 at 20 frames, 96 from 21 on.  Up to 16 frames (16-row tiles): 13..16 frames
    one tile per component, fewer as 4 frames x (x, y, z, pad) per tile."""
    if kernel in ("k_deform32_shared_w1", "k_deform32_tps_shared_wide"):
        nslot = 4 * ((frames + 3) // 4)
        return 32 * ((3 * nslot + 31) // 32)
    return 16 * (3 * ((frames + 15) // 16) if frames > 12 else (frames + 3) // 4)


def rank_report(torch, dist, rank, world, local_rank, rehearse):
    """Who ran: the process group's size as torch.distributed reports it and every rank's device, so
    that a reader of the JSON line sees N ranks on N GPUs rather than taking n_gpus on trust."""
    props = torch.cuda.get_device_properties(local_rank)
    mine = {"rank": rank, "local_device": local_rank, "name": props.name, "uuid": str(getattr(props, "uuid", "")),
            "pid": os.getpid()}
    everyone = [mine]
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    return {"world_size": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() if world > 1 else None),
            "collective_library": ("gloo (FD_BENCH_REHEARSE)" if rehearse else "RCCL (torch.distributed nccl backend)") if world > 1 else None,
            "launcher": os.environ.get("FD_BENCH_LAUNCHER", "external" if "RANK" in os.environ else "none"),
            "devices": everyone}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves,
    as a CHILD process (`python -m torch.distributed.run`, one rank per GPU), before this process
    has made any HIP call -- a process that has initialised the GPU must never be replaced by
    another program.  Rank 0's JSON line and the launcher's return code are relayed."""
    import socket
    import subprocess
    import torch          # importing torch and counting devices do not initialise HIP
    rehearse = os.environ.get("FD_BENCH_REHEARSE") == "1"
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not rehearse:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible on this node; "
                         "refusing to report a multi-GPU figure from fewer devices")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env["FD_BENCH_LAUNCHER"] = "self"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            lines.append(ln)
        else:
            print(ln, file=sys.stderr, flush=True)
    rc = proc.wait()
    for ln in lines:
        print(ln, flush=True)
    if rc == 0 and not lines:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        rc = 1
    raise SystemExit(rc)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    if args.eval_cus <= 0:
        # one-workgroup builds (r2) want 64 CUs to themselves: 192; the register-resident build holds 32 CUs per batch: 224.
        # A run of a single group (steps <= frames per group: the driver's `--steps 20`) has nothing to build beside its one
        # evaluation: the whole device.
        args.eval_cus = 192 if (args.build == "one-workgroup" and args.config == "c2") else 224
        if args.config in ("c2", "c1", "c3") and args.steps <= min(args.inflight, 32):
            args.eval_cus = 256
    import torch
    import torch.distributed as dist
    from facedeform_amd import capi, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (the engine has no CPU path)")
    # FD_BENCH_REHEARSE=1: several ranks share GPU 0 and talk over gloo -- only for rehearsing the
    # multi-rank launch on a one-GPU box (RCCL refuses two ranks on one device)
    rehearse = os.environ.get("FD_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ranks = rank_report(torch, dist, rank, world, local_rank, rehearse)
    n_verts, n_ctrl, mesh_kind, desc = CONFIGS[args.config]
    if args.config == "c5":
        return run_split_mesh(args, torch, dist, capi, synth, rank, world, local_rank, dev, rehearse, ranks)
    P_host = synth.sphere_mesh(n_verts) if mesh_kind == "sphere" else synth.head_mesh(n_verts)
    rest_host = synth.control_points(n_ctrl, mesh_kind)
    deltas_host = np.stack([synth.smooth_deltas(rest_host, f) for f in range(N_FRAMES)])

    d_P = torch.from_numpy(P_host).to(dev)
    d_rest = torch.from_numpy(rest_host).to(dev)
    d_deltas = torch.from_numpy(deltas_host).to(dev)

    precision = capi.EVAL_FP64 if args.precision == "fp64" else capi.EVAL_FP32
    # Independent frames are cooked in groups: a lane holds `inflight` engine contexts (one per
    # frame), one stream and one fd_batch.  Per group the lane enqueues ONE batched build (the
    # assemble + LU + back-substitution launch chain with one workgroup column per frame) on its
    # stream; the evaluations of all lanes go, one per frame, to a single evaluation stream that
    # waits for the group's build.  So the build chains (a few CUs each) run ahead and overlap
    # the evaluations (the whole device), while evaluations never compete with each other.
    # Why batches: a lone order-260 build keeps one CU busy through ~25 dependent launches, and
    # the device was measured to overlap only two or three such chains however many HIP streams
    # they come from (tools/inflight_sweep.py; DESIGN.md "Batched build").
    # Streams of our own: torch's default stream has handle 0, which fd_set_stream reads as
    # "use the context's stream", and HIP events only time the stream they are recorded on.
    B = max(1, min(args.inflight, capi.MAX_BATCH))
    n_lanes = max(1, args.lanes)
    eval_stream = torch.cuda.Stream(device=dev)
    lane_eval_streams = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
    build_streams = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
    cu_split = "none"
    if args.cu_split == "mask" and args.eval_launch == "shared" and B > 1 and 0 < args.eval_cus < 256 and not rehearse:
        ms = masked_streams(torch, dev, n_lanes, args.eval_cus)
        if ms is not None:
            eval_stream, build_streams = ms
            cu_split = "mask"
            if args.eval_stream == "per-lane":
                args.eval_stream = "shared"            # (one masked evaluation stream)
    # the pipeline's models: one workgroup per model where that applies (M <= 512; config c3 takes the chain either way)
    lane_solver = (capi.SOLVER_ONE_WORKGROUP if (args.build == "one-workgroup" and B > 1) else
                   capi.SOLVER_CHAIN if args.build == "chain" else capi.SOLVER_AUTO)      # AUTO: the register-resident one-launch build up to 256 control points
    lanes = []
    for li in range(n_lanes):
        stream = build_streams[li]
        engines = []
        for _ in range(B):
            eng = capi.Engine(device=local_rank, precision=precision, variant=args.variant, solver=lane_solver)
            eng.set_stream(stream.cuda_stream)
            eng.set_kernel(capi.KERNEL_THIN_PLATE)
            eng.set_term(capi.TERM_LINEAR)
            engines.append(eng)
        lane_batch = capi.Batch(engines)
        lane_batch.set_eval_cus(args.eval_cus)       # CU budget of the shared-rig evaluation, per batch (fd_batch_set_eval_cus)
        lanes.append({"engines": engines, "stream": stream, "batches": {B: lane_batch},
                      "evals_done": torch.cuda.Event(), "built": torch.cuda.Event(),
                      "out": [torch.empty_like(d_P) for _ in range(B)],
                      "fall": [torch.zeros(n_verts, device=dev, dtype=torch.float32) for _ in range(B)]})
    torch.cuda.synchronize()

    delta_stride = n_ctrl * 3 * 4
    ev_runs, ev_run_ends = {}, {}      # first step of a timed run -> its length; last step -> first step
    shared_eval = args.eval_launch == "shared" and B > 1
    batched_eval = (args.eval_launch == "batched" and B > 1) or shared_eval

    host_s = [0.0, 0]                                  # host time spent enqueueing timed groups, groups

    def group(g, first, count, ev=None):
        th0 = time.perf_counter()
        _group(g, first, count, ev)
        if ev is not None and ev != "tables-only":
            host_s[0] += time.perf_counter() - th0
            host_s[1] += 1

    def _group(g, first, count, ev=None):
        """Cook steps first .. first+count-1 (count <= B frames) on lane g % n_lanes."""
        ln = lanes[g % n_lanes]
        ln["used"] = max(ln.get("used", 0), count)
        stream = ln["stream"]
        if count not in ln["batches"]:
            ln["batches"][count] = capi.Batch(ln["engines"][:count])
            ln["batches"][count].set_eval_cus(args.eval_cus)
        batch = ln["batches"][count]
        frames = [((first + k) * world + rank) % N_FRAMES for k in range(count)]
        # (a run of ONE group -- the driver's `--steps 20` -- has nothing to overlap: its evaluation goes on the build stream,
        #  behind the packing kernel without a cross-stream event between them)
        es = stream if (args.eval_stream == "lane" or args.steps <= B) else (eval_stream if args.eval_stream == "shared" else lane_eval_streams[g % n_lanes])
        if shared_eval and args.group_call == "c":
            # ONE foreign call per group (fd_batch_cook_group): wait_consumed, set-up, builds, packing, the evaluation.
            key = (count, frames[0])
            tabs = ln.setdefault("tables", {})
            if key not in tabs:
                # (the pointer tables of a group -- static: the same arrays every time the group comes round -- are built on first use;
                #  `prime_only` builds those of the timed groups before the timed region starts)
                tabs[key] = batch.group_tables([d_deltas.data_ptr() + f * delta_stride for f in frames],
                                               [o.data_ptr() for o in ln["out"][:count]], [f.data_ptr() for f in ln["fall"][:count]])
            calls = ln.setdefault("calls", {})
            if ev == "tables-only":
                # ... and the call itself with its arguments marshalled (ctypes conversions: ~10 us of a 350 us group)
                evs = events[first].struct if first in events else None
                calls[(key, first)] = batch.cook_group_call(stream.cuda_stream, es.cuda_stream, d_rest.data_ptr(), n_ctrl, n_verts, d_P.data_ptr(),
                                                            tabs[key], events=evs)
                return
            if ev is not None and (key, first) in calls:
                calls[(key, first)]()            # the timed groups: prepared above, events included
            else:
                # timed groups: four raw HIP events recorded INSIDE the call, around the builds and around the evaluation launch
                batch.cook_group(stream.cuda_stream, es.cuda_stream, d_rest.data_ptr(), n_ctrl, n_verts, d_P.data_ptr(), tabs[key],
                                 events=ev[first].struct if (ev and first in ev) else None)
            ln["last_shared"] = batch
            # (no event of ours behind the evaluation: the lane's next build is ordered by fd_batch_wait_consumed inside the call,
            #  its next evaluation by the evaluation stream itself -- every event record or wait is a barrier packet on that stream,
            #  and four of them per group kept it idle for 40 us between two 171 us launches)
            return
        # The lane's previous group must be done with its models before they are overwritten.  The shared-rig
        # launch copies what it reads of them in its first small kernel (fd_batch_wait_consumed): the next group's
        # assemble + solve then runs while the previous one is still being evaluated.  The other launch styles
        # read the models throughout: wait for the evaluation.
        if shared_eval and ln.get("last_shared") is not None:
            ln["last_shared"].wait_consumed(stream.cuda_stream)
        else:
            stream.wait_event(ln["evals_done"])
        batch.set_points_dev([d_rest.data_ptr()] * count,
                             [d_deltas.data_ptr() + f * delta_stride for f in frames], n_ctrl)
        if ev:
            ev[first][0].record(stream)
        batch.build_async(stream.cuda_stream)
        if ev:
            ev[first][1].record(stream)
        if shared_eval:
            # the weights become fp16 tiles right here, on the build stream (fd_batch_prepare_shared): the evaluation
            # stream then runs evaluation launches back to back
            batch.prepare_shared([o.data_ptr() for o in ln["out"][:count]], d_falloff=[f.data_ptr() for f in ln["fall"][:count]],
                                 stream_ptr=stream.cuda_stream)
        if batched_eval:
            # ONE launch evaluates the group's frames (grid y = frame).  The evaluation stream is
            # made to wait for the build here, so that the event pair holds the launch alone.
            ln["built"].record(stream)
            es.wait_event(ln["built"])
            if ev:
                ev[first][2].record(es)
            if shared_eval:
                batch.deform_shared_dev(n_verts, d_P.data_ptr(), [o.data_ptr() for o in ln["out"][:count]],
                                        d_falloff=[f.data_ptr() for f in ln["fall"][:count]], stream_ptr=es.cuda_stream)
                ln["last_shared"] = batch
            else:
                batch.deform_dev(n_verts, [d_P.data_ptr()] * count, [o.data_ptr() for o in ln["out"][:count]],
                                 d_falloff=[f.data_ptr() for f in ln["fall"][:count]], stream_ptr=es.cuda_stream)
            if ev:
                ev[first][3].record(es)
            ln["evals_done"].record(es)
            return
        for k in range(count):
            # an event pair brackets a run of consecutive evaluations (ev_runs: first step -> length)
            if ev is not None and (first + k) in ev_runs:
                ev[first + k][2].record(es)
            # fd_deform_dev_stream makes es wait for the batch that builds this model
            ln["engines"][k].deform_dev_stream(es.cuda_stream, n_verts, d_P.data_ptr(),
                                               ln["out"][k].data_ptr(), d_falloff=ln["fall"][k].data_ptr())
            if ev is not None and (first + k) in ev_run_ends:
                ev[ev_run_ends[first + k]][3].record(es)
        ln["evals_done"].record(es)

    def run_steps(nsteps, ev=None, g0=0):
        g, i = g0, 0
        while i < nsteps:
            count = min(B, nsteps - i)
            group(g, i, count, ev)
            g += 1
            i += count
        return g

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    solver_seen = []

    def check_builds():
        for ln in lanes:
            for eng in ln["engines"][: ln.get("used", 0)]:
                rep = eng.build_result()
                if rep.terminationtype != 1:
                    raise SystemExit(f"build failed: terminationtype {rep.terminationtype}")
                solver_seen.append(rep.solver_used)

    g_next = run_steps(args.warmup)
    if args.warmup < B * n_lanes:                      # touch every lane and capture its graph once
        g_next = run_steps(B * n_lanes, g0=g_next)
    if args.steps % B:                                 # the ragged last group has its own batch object (primed twice: a batch
        for _ in range(2):                             # alternates between two packed sets, each allocated on first use)
            run_steps(args.steps % B, g0=args.steps // B)
    check_builds()

    c_groups = shared_eval and args.group_call == "c"
    stride = 1
    if c_groups:
        # one set of four events per TIMED group, recorded inside the C call; every `--time-every`-th group is timed (all of them
        # when there are few): the event packets are not free on the evaluation stream (see _group)
        n_groups = (args.steps + B - 1) // B
        stride = 1 if n_groups < 16 else max(1, args.time_every)
        # a run of ONE group (the driver's `--steps 20`) cooks on one stream, where every event record between two kernels is a
        # barrier packet the queue idles ~4 us for: no event between its builds and its packing kernel
        one_group = args.steps <= B
        events = {i: RawEvents(capi, after_build=not one_group, before_build=not one_group) for i in range(0, args.steps, B * stride)}
        run_steps(args.steps, "tables-only", g0=0)         # argument tables of the timed groups: static pointers, built once
        for e in events.values():
            e.warm(lanes[0]["stream"].cuda_stream)
    else:
        events = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    # One HIP event pair per run of `event_every` consecutive evaluations of a group (the pair's
    # own cost, ~4 us of launch hand-over, is spread over the run), never starting on the first
    # evaluation of a group: that one also waits for the group's build.
    run_len = max(1, args.event_every)
    i = 0
    while i < args.steps:
        g_end = min((i // B + 1) * B, args.steps)
        start = i + 1 if (i % B == 0 and g_end - i > 1) else i
        while start < g_end:
            n = min(run_len, g_end - start)
            ev_runs[start] = n
            ev_run_ends[start + n - 1] = start
            start += n
        i = g_end
    ev_idx = sorted(ev_runs)
    import gc
    sync_all()
    gc.disable()                                       # (as timeit does: no collection pause inside the timed region)
    t0 = time.perf_counter()
    run_steps(args.steps, events, g0=0)
    sync_all()
    elapsed = time.perf_counter() - t0
    gc.enable()
    build_events = events
    if c_groups and args.steps <= B:
        # the one group again, untimed, with an event on either side of its builds: `phases_ms.build_batch` (the timed group
        # carries no event in front of its first kernel and none between its builds and its packing kernel)
        for ln in lanes:
            ln["calls"] = {}
        build_events = {0: RawEvents(capi)}
        host_timed = list(host_s)                       # (the repeat is not a timed group: `host.us_per_group` stays the timed call's)
        run_steps(args.steps, build_events, g0=0)
        sync_all()
        host_s[0], host_s[1] = host_timed
    check_builds()

    # SURVEY 8e's alternative for frames that share a rest rig: ONE factorisation per group (fd_batch_set_shared_factor), every
    # frame's right-hand sides through it.  The same steps, the same pipeline, timed again with the switch on and reported
    # under "alternative" -- `value` stays the rebuild-per-frame figure (the reference rebuilds its model every cook).
    alternative = None
    if c_groups and args.shared_factor_alternative and args.build == "register" and n_ctrl <= 256:
        for ln in lanes:
            for bt in ln["batches"].values():
                bt.set_shared_factor(True)
                bt.set_eval_cus(args.alt_eval_cus if args.steps > B else args.eval_cus)
        run_steps(min(args.steps, 2 * B * n_lanes), g0=0)          # every lane once or twice with the switch on
        if args.steps % B:
            run_steps(args.steps % B, g0=args.steps // B)
        sync_all()
        t0 = time.perf_counter()
        run_steps(args.steps, None, g0=0)
        sync_all()
        alt_elapsed = time.perf_counter() - t0
        check_builds()
        flags = [(len(bt), bt.last_build_shared_factor()) for ln in lanes for bt in ln["batches"].values()]
        if os.environ.get("FD_BENCH_DEBUG"):
            print("[shared-factor pass: (batch size, took the shared path)]", flags, file=sys.stderr, flush=True)
        took = any(t for _, t in flags) and all(t for n, t in flags if n == min(B, args.steps) or n == B)
        ta = torch.tensor([alt_elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(ta, op=dist.ReduceOp.MAX)
        alt_elapsed = float(ta.item())
        alternative = {"build": "one factorisation per group of frames that share the rest rig (fd_batch_set_shared_factor: k_build_reg for the "
                                "group's first frame, k_resolve_reg -- one workgroup per frame -- for the others); weights equal the per-frame "
                                "builds' to 1e-12 (tests/test_gpu_shared_factor.py)",
                       "value": world * args.steps * n_verts / alt_elapsed / 1e6, "unit": "Mverts/s", "ms_per_step": alt_elapsed / args.steps * 1e3,
                       "evaluation_cus": args.alt_eval_cus if args.steps > B else args.eval_cus, "took_the_shared_path": bool(took)}
        for ln in lanes:
            for bt in ln["batches"].values():
                bt.set_shared_factor(False)
                bt.set_eval_cus(args.eval_cus)

    # one cook at a time, host-synchronised: the latency a single interactive cook sees
    # (unbatched fd_set_points_dev + fd_build_async + fd_deform_dev on one context)
    lat = []
    ln0 = lanes[0]
    solo = capi.Engine(device=local_rank, precision=precision, variant=args.variant)       # default solver: what a lone cook takes
    solo.set_stream(ln0["stream"].cuda_stream)
    solo.set_kernel(capi.KERNEL_THIN_PLATE)
    solo.set_term(capi.TERM_LINEAR)
    for i in range(10):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng = solo
        eng.set_points_dev(d_rest.data_ptr(), d_deltas.data_ptr() + (i % N_FRAMES) * delta_stride, n_ctrl)
        eng.build_async()
        eng.deform_dev(n_verts, d_P.data_ptr(), ln0["out"][0].data_ptr(), d_falloff=ln0["fall"][0].data_ptr())
        eng.build_result()
        eng.synchronize()
        lat.append(time.perf_counter() - t1)
    latency_ms = float(np.median(lat)) * 1e3
    # the solve alone, one model at a time (fd_build: assemble + factorise + substitute + pack, host-synchronised)
    bl = []
    for i in range(10):
        eng = solo
        eng.set_points_dev(d_rest.data_ptr(), d_deltas.data_ptr() + (i % N_FRAMES) * delta_stride, n_ctrl)
        eng.synchronize()
        t1 = time.perf_counter()
        eng.build_async()
        eng.build_result()
        bl.append(time.perf_counter() - t1)
    single_build_ms = float(np.median(bl)) * 1e3

    t = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # a batched build is timed once per group (events on the group's first step)
    group_firsts = sorted(events) if c_groups else list(range(0, args.steps, B))
    def ev_ms(i, a, b):
        return events[i].ms(a, b) if c_groups else events[i][a].elapsed_time(events[i][b])
    build_group_ms = float(np.mean([(build_events[i].ms(0, build_events[i].build_end) if c_groups else ev_ms(i, 0, 1)) for i in group_firsts]))
    build_ms = build_group_ms / min(B, args.steps)
    if batched_eval:
        # one event pair per evaluation launch; a launch covers the frames of its group
        g_counts = [min(B, args.steps - i) for i in group_firsts]
        g_ms = [ev_ms(i, 2, 3) for i in group_firsts]
        eval_ms = float(np.sum(g_ms) / np.sum(g_counts))                 # per frame
        full = [m for m, c in zip(g_ms, g_counts) if c == max(g_counts)]
        frames_per_launch = int(max(g_counts))
        launch_ms = float(np.mean(full))
    else:
        eval_ms = float(np.sum([events[i][2].elapsed_time(events[i][3]) for i in ev_idx]) / sum(ev_runs[i] for i in ev_idx))
        frames_per_launch = 1
        launch_ms = eval_ms

    if rank == 0 and os.environ.get("FD_BENCH_GAPS") and args.eval_stream == "shared" and not batched_eval:
        # idle time on the evaluation stream between consecutive evaluations (diagnostic)
        pairs = [(a, b) for a, b in zip(ev_idx[:-1], ev_idx[1:]) if a // B == b // B and a + ev_runs[a] == b]
        gaps = np.array([events[a][3].elapsed_time(events[b][2]) * 1e3 for a, b in pairs])
        print(f"[gaps us] idle between consecutive timed runs within a group: mean {gaps.mean():.1f} median {np.median(gaps):.1f} "
              f"max {gaps.max():.1f}", file=sys.stderr, flush=True)
    if rank == 0 and os.environ.get("FD_BENCH_GAPS") and not c_groups:
        base = events[0][0]
        first = range(0, args.steps, B)
        for i in list(first[:6]) + list(first[40:52]):
            print(f"[timeline ms] group {i // B}: build {base.elapsed_time(events[i][0]):8.3f} -> "
                  f"{base.elapsed_time(events[i][1]):8.3f}   evaluation {base.elapsed_time(events[i][2]):8.3f} -> "
                  f"{base.elapsed_time(events[i][3]):8.3f}", file=sys.stderr, flush=True)
    lane_solver_used = max(set(solver_seen), key=solver_seen.count) if solver_seen else lane_solver
    host_cook = host_cook_ms(capi, synth, P_host, rest_host, local_rank) if (rank == 0 and world == 1 and not rehearse) else None
    if rank == 0:
        total_verts = world * args.steps * n_verts
        flops = (FLOPS_PER_PAIR * n_ctrl + FLOPS_PER_VERTEX_AFFINE) * n_verts
        achieved_tflops = flops / (eval_ms * 1e-3) / 1e12
        achieved_gbs = BYTES_PER_VERTEX * n_verts / (eval_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (profiles/traffic_<config>.json, one entry per
        # kernel): what a launch moves whatever its frame count (the mesh, the model) + what it moves per frame.
        tj = {}
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.config}.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))["kernels"]
            except Exception:
                tj = {}

        def measured_traffic(kernel):
            """HBM bytes of one launch from the COMMITTED counter passes -- not a measurement of this run, and only quoted when
            this run's launches carry as many frames as the profiled ones did (no extrapolation to another frame count)."""
            k = tj.get(f"{kernel}@{frames_per_launch}") or tj.get(kernel)       # (a kernel profiled at several frame counts: one entry each)
            if not k or k.get("frames_per_launch", 1) != frames_per_launch:
                return None
            return k.get("hbm_bytes_per_launch", k["hbm_bytes_fixed_per_launch"] + k["hbm_bytes_per_frame"] * frames_per_launch)

        def committed_pmc(kernel):
            k = tj.get(f"{kernel}@{frames_per_launch}") or tj.get(kernel)
            if not k:
                return None
            return {"source": f"profiles/traffic_{args.config}.json", "round": k.get("round"), "frames_measured": k.get("frames_per_launch"),
                    "hbm_bytes_per_launch_measured": k.get("hbm_bytes_per_launch"),
                    "hbm_bytes_fixed_per_launch": k.get("hbm_bytes_fixed_per_launch"), "hbm_bytes_per_frame": k.get("hbm_bytes_per_frame"),
                    "mfma_busy_frac": k.get("mfma_busy_frac"), "wait_inst_frac": k.get("wait_inst_frac"),
                    "note": "rocprofv3 --pmc passes of an earlier run of this command (profiles/), not counters of this run"}

        # the kernel that takes most GPU time in the committed kernel trace of this command, and where it stands against its roof
        dominant = None
        dpath = os.path.join(ROOT, "profiles", f"dominant_{args.config}.json")
        if os.path.exists(dpath):
            try:
                dominant = json.load(open(dpath))
            except Exception:
                dominant = None
        mfma_eval = precision == capi.EVAL_FP32 and n_ctrl >= 49
        use_shared = shared_eval and mfma_eval and n_ctrl >= 32
        if use_shared:
            # Frames of a launch share the mesh and the rest rig: phi is formed once per (vertex, centre)
            # and contracted with the 3 F weight columns on the matrix pipe.  Algorithmic work of the
            # launch: per pair 8 flop once (d2: 5, d2 log d2: 3) + 6 flop per frame;
            # per vertex 24 flop of polynomial per frame; bytes: P read once, P + fd_falloff written per frame.
            Fl = frames_per_launch
            flops_launch = ((8 + 6 * Fl) * n_ctrl + FLOPS_PER_VERTEX_AFFINE * Fl) * n_verts
            bytes_launch = (12 + 16 * Fl) * n_verts
            # the kernel the library launches for this (M, frames): 17..32 frames take 32-row tiles -- k_deform32_shared_w1 where the
            # model is resident in LDS (C2), k_deform32_tps_shared_wide where it is staged in chunks (C3, C5)
            kern = capi.load().fd_shared_kernel_name(n_ctrl, Fl, capi.KERNEL_THIN_PLATE).decode()
            # executed on the matrix pipe: 3 split products over the rows of the output tiles + the d2 tiles + the polynomial tile
            rows = shared_rows(Fl, kern)
            mfma_exec = (3 * 2 * rows * n_ctrl + 2 * 16 * n_ctrl + 2 * 16 * rows) * n_verts        # flop, fp16 MFMA
            secs = launch_ms * 1e-3
            mfma_alg = {"achieved": flops_launch / secs / 1e12, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops_launch / secs / 1e12 / PEAK_FP16_MFMA_TFLOPS, "flops_per_launch": flops_launch,
                        "executed": {"achieved": mfma_exec / secs / 1e12, "frac": mfma_exec / secs / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                                     "note": "what the pipe runs: fp16 x 2 split = three products"},
                        }
            hbm_alg = {"achieved": bytes_launch / secs / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": bytes_launch / secs / 1e9 / PEAK_HBM_GBS, "bytes_per_launch": bytes_launch}
            # Which roof binds is the roofline model's answer at the launch's ALGORITHMIC intensity: below the
            # ridge of the fp16 matrix pipe over HBM (312 flop/B) the outputs bound it (C2 x 32 frames: 98 flop/B,
            # 512 MB written per launch), above it the matrix pipe does (C3: 780 flop/B).  Both are on the line.
            intensity = flops_launch / bytes_launch
            ridge = PEAK_FP16_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            first, second, key = (hbm_alg, mfma_alg, "mfma") if intensity < ridge else (mfma_alg, hbm_alg, "hbm")
            roof = {"bound": "hbm" if intensity < ridge else "mfma", "kernel": kern,
                    "achieved": first["achieved"], "peak": first["peak"], "unit": first["unit"], "frac": first["frac"],
                    "traffic": measured_traffic(kern), "committed_pmc": committed_pmc(kern), "dominant_by_gpu_time": dominant,
                    "avg_launch_ms": launch_ms, "frames_per_launch": Fl,
                    "intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge,
                    "bytes_per_launch": bytes_launch, "flops_per_launch": flops_launch, key: second,
                    ("hbm" if key == "mfma" else "mfma"): first}
        else:
            kern = (("k_deform32_tps_mfma_batch" if frames_per_launch > 1 else "k_deform32_tps_mfma")
                    if mfma_eval else ("k_deform32" if precision == capi.EVAL_FP32 else "k_deform64"))
            roof = {
                # the one-frame kernels are compute-bound at this M (intensity ~182 flop/B vs ridge ~20) and
                # their binding pipe is the fp32 VECTOR unit (logarithm + weight contraction), not the
                # matrix pipe, which only forms the squared distances: PMC in profiles/ (VALU active
                # ~90-100 % of busy cycles, MFMA pipe busy 13-17 %).  Roof: 157.3 TFLOP/s fp32 vector.
                "bound": "valu_fp32", "kernel": kern,
                "achieved": achieved_tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / PEAK_FP32_TFLOPS, "traffic": measured_traffic(kern), "committed_pmc": committed_pmc(kern),
                "dominant_by_gpu_time": dominant,
                # a launch evaluates frames_per_launch frames (algorithmic flops per frame x frames)
                "flops_per_launch": flops * frames_per_launch, "avg_launch_ms": launch_ms,
                "frames_per_launch": frames_per_launch,
                "hbm": {"achieved": achieved_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": achieved_gbs / PEAK_HBM_GBS,
                        "bytes_per_launch": BYTES_PER_VERTEX * n_verts * frames_per_launch},
            }
        line = {
            "metric": METRIC,
            "value": total_verts / elapsed / 1e6,
            "unit": "Mverts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if precision == capi.EVAL_FP32 else "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{desc}, thin-plate kernel, linear term, {args.precision} evaluation"
                            + (" (squared distances on the matrix pipe from fp16 x 2 split operands, 22 bits; fp32 logarithm"
                               + ("; the frames of a group share the mesh and the rest rig, so d2 log d2 is formed once per "
                                  "(vertex, centre) and contracted with every frame's weights on the matrix pipe, fp16 x 2 split "
                                  "operands, fp32 accumulation)" if use_shared else "; fp32 weights and accumulation)") if mfma_eval else "")
                            + ", every frame's model assembled and solved on its own in fp64, every step (as the reference rebuilds "
                              "its model every cook), one frame per step",
                "n_verts": n_verts, "n_ctrl": n_ctrl,
                "frames_per_batched_build": B, "frames_per_evaluation_launch": frames_per_launch,
                "evaluation": args.eval_launch if B > 1 else "single",
                "evaluation_cus": args.eval_cus or 256,
                "evaluation_stream": "the group's build stream (a single group: nothing to overlap)" if args.steps <= B else args.eval_stream,
                "timed_groups": f"HIP event pairs around the build and the evaluation launch of every {stride}-th group" if c_groups else "every group",
                "cu_split": cu_split,
                # what the lanes' builds actually ran (fd_report.solver_used of a lane's last build), not what was asked for
                "pipeline_build": {capi.SOLVER_REGISTER: "one launch, one workgroup per model, matrix in registers (fd_build_reg.hip)",
                                   capi.SOLVER_ONE_WORKGROUP: "one workgroup per model, matrix in L2 (FD_SOLVER_ONE_WORKGROUP)",
                                   capi.SOLVER_CHAIN: "launch chain: null-space Cholesky, one launch per 32 columns (fd_nullspace.hip)",
                                   capi.SOLVER_LU: "launch chain: pivoted LU (fd_build.hip)",
                                   capi.SOLVER_LU_NOPIVOT: "launch chain: LU without pivot search (fd_build.hip)"}.get(lane_solver_used, f"solver {lane_solver_used}"),
                "lanes_per_gpu": n_lanes,
                "parallelism": f"independent frames: {world} GPU(s) x {n_lanes} lanes x {B} frames per batched "
                               "build and per evaluation launch (one build stream per lane, one evaluation "
                               "stream), no collective",
            },
            "roofline": roof,
            # the dense solve of ONE model against the fp64 matrix peak ((1/3) n1^3 for the Cholesky of the
            # projected block, n1 = M - 4; batched, 32 models share a launch chain: build_batch / 32 per model)
            "roofline_solve": {"bound": "mfma_fp64", "flops_per_model": (n_ctrl - 4) ** 3 / 3.0, "peak": PEAK_FP64_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "single_build_ms": single_build_ms,
                               "achieved_single": (n_ctrl - 4) ** 3 / 3.0 / (single_build_ms * 1e-3) / 1e12,
                               "frac_single": (n_ctrl - 4) ** 3 / 3.0 / (single_build_ms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                               "achieved_batched": (n_ctrl - 4) ** 3 / 3.0 / (build_ms * 1e-3) / 1e12,
                               "frac_batched": (n_ctrl - 4) ** 3 / 3.0 / (build_ms * 1e-3) / 1e12 / PEAK_FP64_MFMA_TFLOPS,
                               "note": "a chain of dependent launches on a small matrix: latency, not flops, bounds it (DESIGN.md 4.2b/4.2c)"},
            "alternative": alternative,
            "ranks": ranks,
            # host side of the pipeline: wall time the rank's Python thread spends enqueueing one group (fd_batch_cook_group: one
            # foreign call; --group-call python: five calls with pointer tables built per group)
            "host": {"group_call": args.group_call if shared_eval else "python", "us_per_group": host_s[0] / max(1, host_s[1]) * 1e6,
                     "groups_timed": host_s[1]},
            "phases_ms": {"build_per_frame_batched": build_ms, "build_batch": build_group_ms,
                          # (a run of one group has no event between its builds and its packing kernel: see RawEvents)
                          "build_batch_measured_on": ("an untimed repeat of the group, with events around its builds" if (c_groups and args.steps <= B)
                                                      else "the timed groups"),
                          "evaluate": eval_ms, "single_cook_latency": latency_ms, "single_build": single_build_ms,
                          # PCIe-inclusive, through the cook mirror on page-locked arrays; never `value`
                          "end_to_end_host_cook": host_cook["rebuild"] if host_cook else None,
                          "end_to_end_host_cook_static_mesh_and_rig": host_cook["mesh_and_rest_rig_unchanged"] if host_cook else None},
            "end_to_end_host_cook_mverts_s": (n_verts / (host_cook["rebuild"] * 1e-3) / 1e6) if host_cook else None,
            "eval_only_mverts_s": n_verts / (eval_ms * 1e-3) / 1e6,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, P_host, rest_host,
                                                (rest_host + deltas_host[0]).astype(np.float32), args.cpu_pairs)
        print(json.dumps(line), flush=True)

    torch.cuda.synchronize()
    solo.set_stream(None)
    solo.close()
    for ln in lanes:
        for batch in ln["batches"].values():
            batch.close()
        for eng in ln["engines"]:
            eng.set_stream(None)
            eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
