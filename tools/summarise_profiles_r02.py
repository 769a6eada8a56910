"""gpurun_out/profiles_r02/ (tools/collect_profiles_r02.sh) -> the round-2 files committed under profiles/."""
import csv
import glob
import json
import os
import shutil

SRC = "gpurun_out/profiles_r02"
DST = "profiles"
KERNEL = "tps_shared"


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def clean(path):
    return "".join(l for l in open(path) if "amdgpu" not in l)


def main():
    for a, b in (("bench_c2_line.json", "r02_bench_c2_line.json"), ("bench_c2_profiled.json", "r02_bench_c2_profiled_line.json"),
                 ("bench_c2_independent_line.json", "r02_bench_c2_independent_line.json"), ("bench_c3_line.json", "r02_bench_c3_line.json"),
                 ("bench_c5_line.json", "r02_bench_c5_line.json")):
        if os.path.exists(f"{SRC}/{a}"):
            json.dump(last_json(f"{SRC}/{a}"), open(f"{DST}/{b}", "w"), indent=1)
    shutil.copy(glob.glob(f"{SRC}/bench/**/*kernel_stats.csv", recursive=True)[0], f"{DST}/r02_bench_c2_kernel_stats.csv")
    shutil.copy(glob.glob(f"{SRC}/shared/**/*kernel_stats.csv", recursive=True)[0], f"{DST}/r02_shared_c2_kernel_stats.csv")
    for a, b in (("scaled_delta_parity.txt", "r02_scaled_delta_parity.txt"), ("shared_timing_c2.txt", "r02_shared_timing.txt"),
                 ("solver_latency.txt", "r02_solver_latency.txt"), ("hbm_write_rate.txt", "r02_hbm_write_rate.txt"),
                 ("ubench_mfma16.txt", "r02_ubench_mfma16.txt"), ("wide_variants.txt", "r02_wide_variants.txt")):
        if os.path.exists(f"{SRC}/{a}"):
            open(f"{DST}/{b}", "w").write(clean(f"{SRC}/{a}"))
    tot = {}
    kname = "k_deform32_tps_shared"
    for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_mfma"):
        for r in csv.DictReader(open(glob.glob(f"{SRC}/{d}/**/*counter_collection.csv", recursive=True)[0])):
            if KERNEL in r["Kernel_Name"]:
                tot.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if "shared_wide" in r["Kernel_Name"]:
                    kname = "k_deform32_tps_shared_wide"
    m = {k: sum(v) / len(v) for k, v in tot.items()}
    fetch, write = m["FETCH_SIZE"] * 1024 * 2, m["WRITE_SIZE"] * 1024
    cycles = m["GRBM_GUI_ACTIVE"] / 8
    with open(f"{DST}/r02_pmc_shared_c2.txt", "w") as f:
        f.write(f"{kname}, C2 (N=1e6, M=256), 32 thin-plate frames per launch; rocprofv3 --pmc, one counter group per pass\n")
        f.write(f"(tools/collect_profiles_r02.sh: FETCH_SIZE | WRITE_SIZE | SQ_* activity | SQ_*MFMA/LDS), mean over {len(tot['FETCH_SIZE'])} launches\n\n")
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
    k = tj["kernels"].setdefault(kname, dict(tj["kernels"]["k_deform32_tps_shared"]))
    k.update({"FETCH_SIZE_KiB_raw": m["FETCH_SIZE"], "WRITE_SIZE_KiB": m["WRITE_SIZE"], "hbm_bytes_per_launch": fetch + write,
              "hbm_bytes_fixed_per_launch": fetch, "hbm_bytes_per_frame": write / 32,
              "mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cycles * 1024),
              "valu_active_frac": m["SQ_ACTIVE_INST_VALU"] * 4 / m["SQ_WAVE_CYCLES"] / 4 if m.get("SQ_WAVE_CYCLES") else None,
              "wait_inst_frac": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]})
    json.dump(tj, open(f"{DST}/traffic_c2.json", "w"), indent=1)
    print(open(f"{DST}/r02_pmc_shared_c2.txt").read())
    d = last_json(f"{SRC}/bench_c2_line.json")
    print("bench:", round(d["value"]), d["ms_per_step"], d["phases_ms"], d["roofline"]["bound"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])


if __name__ == "__main__":
    main()
