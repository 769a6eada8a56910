"""gpurun_out/profiles_r04/ (tools/collect_profiles_r04.sh) -> the round-4 files committed under profiles/."""
import csv
import glob
import json
import os
import shutil

SRC = "gpurun_out/profiles_r04"
DST = "profiles"
PEAK_HBM = 8000.0e9


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def clean(path):
    return "".join(l for l in open(path) if "amdgpu" not in l and "simple_timer" not in l)


def counters(dirs, match):
    tot = {}
    names = set()
    for d in dirs:
        files = glob.glob(f"{SRC}/{d}/**/*counter_collection.csv", recursive=True)
        if not files:
            continue
        for r in csv.DictReader(open(files[0])):
            if match in r["Kernel_Name"]:
                tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"].split("(")[0][:80])
    return {k: sum(v) / len(v) for k, v in tot.items()}, {k: len(v) for k, v in tot.items()}, names


def kernel_stats(path):
    rows = list(csv.DictReader(open(path)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    return rows, total


def main():
    for a, b in (("bench_c2_line.json", "r04_bench_c2_line.json"), ("bench_c2_20_line.json", "r04_bench_c2_20steps_line.json"),
                 ("bench_c2_profiled.json", "r04_bench_c2_profiled_line.json"), ("bench_c2_independent_line.json", "r04_bench_c2_independent_line.json"),
                 ("bench_c3_line.json", "r04_bench_c3_line.json"), ("bench_c5_line.json", "r04_bench_c5_line.json")):
        if os.path.exists(f"{SRC}/{a}") and os.path.getsize(f"{SRC}/{a}"):
            json.dump(last_json(f"{SRC}/{a}"), open(f"{DST}/{b}", "w"), indent=1)
    bs = glob.glob(f"{SRC}/bench/**/*kernel_stats.csv", recursive=True)[0]
    shutil.copy(bs, f"{DST}/r04_bench_c2_kernel_stats.csv")
    shutil.copy(glob.glob(f"{SRC}/shared/**/*kernel_stats.csv", recursive=True)[0], f"{DST}/r04_shared_c2_kernel_stats.csv")
    shutil.copy(glob.glob(f"{SRC}/reg/**/*kernel_stats.csv", recursive=True)[0], f"{DST}/r04_build_reg_kernel_stats.csv")
    for sub, name in (("regsplit", "r04_build_front_single_kernel_stats.csv"), ("regsplit20", "r04_build_front_batch20_kernel_stats.csv")):
        g = glob.glob(f"{SRC}/{sub}/**/*kernel_stats.csv", recursive=True)
        if g:
            shutil.copy(g[0], f"{DST}/{name}")
    s20 = glob.glob(f"{SRC}/shared20/**/*kernel_stats.csv", recursive=True)
    if s20:
        shutil.copy(s20[0], f"{DST}/r04_shared_c2_20frames_kernel_stats.csv")
    for a, b in (("shared_timing_c2.txt", "r04_shared_timing.txt"), ("solver_latency.txt", "r04_solver_latency.txt"), ("qnn_latency.txt", "r04_qnn_latency.txt"),
                 ("ubench_f64.txt", "r04_ubench_f64.txt"), ("reg_build_check.txt", "r04_reg_build_check.txt"),
                 ("shared_factor_timing.txt", "r04_shared_factor_timing.txt"), ("host_path.txt", "r04_host_path.txt")):
        if os.path.exists(f"{SRC}/{a}"):
            open(f"{DST}/{b}", "w").write(clean(f"{SRC}/{a}"))
    with open(f"{DST}/r04_bench_c2_variants.txt", "w") as f:
        f.write("bench.py at C2, one line per variant of the pipeline (same box, back to back): value Mverts/s | ms/step | build per frame batched | evaluate per frame | single build | single cook\n")
        for a, what in (("bench_c2_line.json", "default (register build, 224 evaluation CUs)"), ("bench_c2_cus192.json", "--eval-cus 192"), ("bench_c2_cus256.json", "--eval-cus 256"),
                        ("bench_c2_chain.json", "--build chain"), ("bench_c2_onewg.json", "--build one-workgroup (round 2's pipeline)"),
                        ("bench_c2_20_line.json", "--steps 20 --warmup 5 (the driver's form)"), ("bench_c2_independent_line.json", "--eval-launch batched (independent rigs)")):
            if os.path.exists(f"{SRC}/{a}") and os.path.getsize(f"{SRC}/{a}"):
                d = last_json(f"{SRC}/{a}")
                ph = d["phases_ms"]
                alt = d.get("alternative") or {}
                f.write(f"{what:52s} {d['value']:10.0f} (shared factor: {alt.get('value', 0):.0f}) | {d['ms_per_step']:.5f} | {ph['build_per_frame_batched']:.5f} | {ph['evaluate']:.5f} | {ph['single_build']:.4f} | {ph['single_cook_latency']:.4f}\n")

    # ---- the evaluation kernel: PMC
    m, n, names = counters(("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mfma"), "k_deform32_shared_w1")
    kname = "k_deform32_shared_w1"
    fetch, write = m["FETCH_SIZE"] * 1024 * 2, m["WRITE_SIZE"] * 1024
    cycles = m["GRBM_GUI_ACTIVE"] / 8
    with open(f"{DST}/r04_pmc_shared_c2.txt", "w") as f:
        f.write(f"{kname}, C2 (N=1e6, M=256), 32 thin-plate frames per launch; rocprofv3 --pmc, one counter group per pass\n")
        f.write(f"(tools/collect_profiles_r04.sh: FETCH_SIZE | WRITE_SIZE | SQ_* activity | SQ_*MFMA/LDS), mean over {n['FETCH_SIZE']} launches\n\n")
        for k in sorted(m):
            f.write(f"{k:28s} {m[k]:16.1f}\n")
        f.write(f"\nHBM read  = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) = {fetch / 1e6:.2f} MB  (algorithmic: P 12.00 MB + model tiles)\n")
        f.write(f"HBM write = WRITE_SIZE KiB x 1024 = {write / 1e6:.2f} MB  (algorithmic: 32 x (12 + 4) MB = 512.00 MB)\n")
        f.write(f"traffic / algorithmic = {(fetch + write) / (12e6 + 32 * 16e6):.4f}\n")
        f.write(f"\nGRBM_GUI_ACTIVE / 8 XCDs = {cycles:.0f} cycles per launch (the counter is summed over the XCDs)\n")
        f.write(f"matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs) = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024):.3f}\n")
        f.write(f"MFMA instructions per launch = {m['SQ_INSTS_MFMA']:.0f} (wave level)\n")
        f.write(f"VALU instructions = {m['SQ_INSTS_VALU']:.0f}; LDS instructions = {m['SQ_INSTS_LDS']:.0f}; LDS bank conflicts = {m['SQ_LDS_BANK_CONFLICT']:.0f}\n")
        f.write(f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.3f};  SQ_WAIT_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}\n")
    tj = json.load(open(f"{DST}/traffic_c2.json"))
    k = tj["kernels"].setdefault(kname, {})
    k.update({"round": 4, "frames_per_launch": 32, "FETCH_SIZE_KiB_raw": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"], "hbm_bytes_per_launch": fetch + write,
              "hbm_bytes_fixed_per_launch": fetch, "hbm_bytes_per_frame": write / 32,
              "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024),
              "valu_active_frac": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
              "wait_inst_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]})

    # ---- the same kernel at the driver's 20 frames per launch: bytes and pipe activity (its own entry: bench.py quotes traffic only
    #      for the frame count that was measured)
    m20, n20, _ = counters(("pmc_fetch20", "pmc_write20", "pmc_mfma20"), "k_deform32_shared_w1")
    if "FETCH_SIZE" in m20 and "WRITE_SIZE" in m20:
        f20, w20 = m20["FETCH_SIZE"] * 1024 * 2, m20["WRITE_SIZE"] * 1024
        c20 = m20.get("GRBM_GUI_ACTIVE", 0.0) / 8
        tj["kernels"][kname + "@20"] = {"round": 4, "frames_per_launch": 20, "FETCH_SIZE_KiB_raw": m20["FETCH_SIZE"], "WRITE_SIZE_KiB": m20["WRITE_SIZE"],
                                        "hbm_bytes_per_launch": f20 + w20, "hbm_bytes_fixed_per_launch": f20, "hbm_bytes_per_frame": w20 / 20,
                                        "mfma_busy_frac": (m20["SQ_VALU_MFMA_BUSY_CYCLES"] / (c20 * 1024)) if c20 and "SQ_VALU_MFMA_BUSY_CYCLES" in m20 else None,
                                        "wait_inst_frac": (m20["SQ_WAIT_INST_ANY"] / m20["SQ_WAVE_CYCLES"]) if m20.get("SQ_WAVE_CYCLES") else None}
        with open(f"{DST}/r04_pmc_shared_c2_20frames.txt", "w") as f:
            f.write(f"{kname}, C2 (N=1e6, M=256), 20 thin-plate frames per launch (the driver's `--steps 20`); rocprofv3 --pmc, one counter group per pass, mean over {n20['FETCH_SIZE']} launches\n\n")
            for kk in sorted(m20):
                f.write(f"{kk:28s} {m20[kk]:16.1f}\n")
            f.write(f"\nHBM read = FETCH_SIZE KiB x 1024 x 2 = {f20 / 1e6:.2f} MB, write = {w20 / 1e6:.2f} MB; algorithmic 12 + 20 x 16 = 332.00 MB; traffic / algorithmic = {(f20 + w20) / 332e6:.4f}\n")
            if c20:
                f.write(f"matrix pipe busy = {m20['SQ_VALU_MFMA_BUSY_CYCLES'] / (c20 * 1024):.3f}; MFMA instructions {m20['SQ_INSTS_MFMA']:.0f}\n")

    # ---- the register-resident build: PMC
    r, rn, _ = counters(("pmc_reg_mfma", "pmc_reg_sq", "pmc_reg_fetch", "pmc_reg_write"), "k_build_reg")
    rows, _ = kernel_stats(f"{DST}/r04_build_reg_kernel_stats.csv")
    reg_row = [x for x in rows if "k_build_reg" in x["Name"]][0]
    reg_us = float(reg_row["AverageNs"]) / 1e3
    rcycles = r["GRBM_GUI_ACTIVE"] / 8
    n1 = 252
    flops = n1 ** 3 / 3.0
    with open(f"{DST}/r04_pmc_build_reg.txt", "w") as f:
        f.write("k_build_reg (fd_build_reg.hip), one model of M = 256 control points, thin-plate + linear term: ONE workgroup of 512 threads on one CU\n")
        f.write(f"rocprofv3 --kernel-trace --stats: average {reg_us:.1f} us over {reg_row['Calls']} launches (tools/build_profile_batched.py 256 1 40 224: the one-workgroup form, k_build_reg<true>)\n")
        f.write(f"rocprofv3 --pmc, one counter group per pass, mean over {rn.get('SQ_INSTS_MFMA', 0)} launches\n\n")
        for kk in sorted(r):
            f.write(f"{kk:28s} {r[kk]:16.1f}\n")
        f.write(f"\nGRBM_GUI_ACTIVE / 8 XCDs = {rcycles:.0f} cycles per launch\n")
        f.write(f"fp64 matrix instructions (16x16x4) per launch = {r['SQ_INSTS_MFMA']:.0f}; at 64 cycles each on 4 SIMDs: {r['SQ_INSTS_MFMA'] * 64 / 4:.0f} cycles if perfectly spread\n")
        f.write(f"matrix pipe busy on the ONE CU the kernel occupies = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 4 SIMDs) = {r['SQ_VALU_MFMA_BUSY_CYCLES'] / (rcycles * 4):.3f}\n")
        f.write(f"algorithmic flop of the solve, (1/3) n1^3 with n1 = {n1}: {flops / 1e6:.2f} Mflop -> {flops / (reg_us * 1e-6) / 1e12:.4f} TFLOP/s = "
                f"{flops / (reg_us * 1e-6) / 1e12 / 78.6:.5f} of the chip's 78.6 TFLOP/s fp64 matrix peak; against ONE CU's share (78.6 / 256 = 0.307 TFLOP/s): "
                f"{flops / (reg_us * 1e-6) / 1e12 / (78.6 / 256):.3f}\n")
        f.write(f"HBM: read {r['FETCH_SIZE'] * 1024 * 2 / 1e3:.1f} KB (x2 corrected), written {r['WRITE_SIZE'] * 1024 / 1e3:.1f} KB per launch "
                f"(algorithmic: 6 KB of control points in; 272 KB staging of K out and back in L2, 50 KB of weights, records and tiles out)\n")
        f.write(f"LDS instructions {r['SQ_INSTS_LDS']:.0f}, bank conflict cycles {r['SQ_LDS_BANK_CONFLICT']:.0f}; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {r['SQ_WAIT_INST_ANY'] / r['SQ_WAVE_CYCLES']:.3f}\n")
    tj["kernels"]["k_build_reg"] = {"round": 4, "frames_per_launch": 1, "avg_launch_us": reg_us, "mfma_insts": r["SQ_INSTS_MFMA"],
                                    "mfma_busy_frac_one_cu": r["SQ_VALU_MFMA_BUSY_CYCLES"] / (rcycles * 4),
                                    "hbm_bytes_per_launch": r["FETCH_SIZE"] * 1024 * 2 + r["WRITE_SIZE"] * 1024,
                                    "hbm_bytes_fixed_per_launch": r["FETCH_SIZE"] * 1024 * 2 + r["WRITE_SIZE"] * 1024, "hbm_bytes_per_frame": 0.0}
    json.dump(tj, open(f"{DST}/traffic_c2.json", "w"), indent=1)

    # ---- which kernel takes the GPU time of the default bench command, and where it stands against its roof
    import re
    rows, total = kernel_stats(f"{DST}/r04_bench_c2_kernel_stats.csv")
    rows.sort(key=lambda x: -float(x["TotalDurationNs"]))

    def kname_of(x):
        mm = re.search(r"(k_\w+|__amd_\w+)", x["Name"])
        return mm.group(1) if mm else x["Name"][:60]

    def roof_of(x):
        avg = float(x["AverageNs"]) * 1e-9
        nm = kname_of(x)
        if "tps_shared" in nm or "shared_w1" in nm:
            alg = (12 + 16 * 32) * 1.0e6
            return {"bound": "hbm", "achieved": alg / avg / 1e9, "unit": "GB/s", "peak": PEAK_HBM / 1e9, "frac_of_roof": alg / avg / PEAK_HBM,
                    "note": "32 frames per launch: (12 + 16 x 32) MB algorithmic"}
        if "k_build_reg" in nm:
            fl = 32 * 252 ** 3 / 3.0
            return {"bound": "mfma_fp64", "achieved": fl / avg / 1e12, "unit": "TFLOP/s", "peak": 78.6, "frac_of_roof": fl / avg / 1e12 / 78.6,
                    "note": "32 models per launch on 32 CUs ((1/3) n1^3 each); a latency chain on one CU per model, 12.5 % of the chip: against those 32 CUs' share "
                            f"{fl / avg / 1e12 / (78.6 / 8):.3f}"}
        return None
    top = rows[0]
    dom = {"source": "profiles/r04_bench_c2_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline`)",
           "note": "shares are of SUMMED kernel durations; the build's 32 workgroups (32 CUs) and the evaluation (224 CUs) run side by side",
           "kernel": kname_of(top), "share_of_gpu_time": float(top["TotalDurationNs"]) / total, "calls": int(top["Calls"]),
           "avg_launch_ms": float(top["AverageNs"]) / 1e6, "roof": roof_of(top),
           "next": [{"kernel": kname_of(x), "share_of_gpu_time": float(x["TotalDurationNs"]) / total, "avg_us": float(x["AverageNs"]) / 1e3, "roof": roof_of(x)} for x in rows[1:4]]}
    json.dump(dom, open(f"{DST}/dominant_c2.json", "w"), indent=1)
    print(open(f"{DST}/r04_pmc_shared_c2.txt").read())
    print(open(f"{DST}/r04_pmc_build_reg.txt").read())
    print(open(f"{DST}/r04_bench_c2_variants.txt").read())
    print(json.dumps(dom, indent=1))


if __name__ == "__main__":
    main()
