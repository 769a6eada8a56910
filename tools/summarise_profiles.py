"""Turn gpurun_out/profiles_r01/ (tools/collect_profiles.sh) into the files committed under
profiles/: kernel-stats CSVs, the bench JSON line, the PMC summary and traffic_c2.json."""
import collections, csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles_r01")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def counters(sub, match):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
                agg[m.group(1) if m else r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v)


def main():
    for sub, name in (("bench", "bench_c2"), ("bench1", "bench_c2_single"), ("evalonly", "evalonly_c2")):
        f = os.path.join(SRC, sub, f"{name}_kernel_stats.csv")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(DST, f"{TAG}_{name}_kernel_stats.csv"))
    for name in ("bench_c2", "bench_c2_single"):
        log = os.path.join(SRC, f"{name}.log")
        if os.path.exists(log):
            lines = [l for l in open(log) if l.startswith("{")]
            if lines:
                open(os.path.join(DST, f"{TAG}_{name}_line.json"), "w").write(lines[-1])

    out = []
    kern = "k_deform32_tps_mfma_batch"
    frames = 32                       # tools/collect_profiles.sh evaluates 32 frames per launch
    fetch = counters("pmc_fetch", kern)
    write = counters("pmc_write", kern)
    sq = counters("pmc_sq", kern)
    mf = counters("pmc_mfma", kern)
    kname = next(iter(fetch), next(iter(sq), "?"))
    out.append(f"rocprofv3 --pmc <counters> --kernel-trace (one counter group per run), evaluation kernel {kname} at C2 "
               f"(N=1e6, M=256), {frames} frames per launch as in bench.py (tests/tools/batch_eval_timing.py 32 batched); "
               f"counter values are per launch")
    traffic = {}
    if fetch and write:
        fk = mean(next(iter(fetch.values()))["FETCH_SIZE"])
        wk = mean(next(iter(write.values()))["WRITE_SIZE"])
        hbm = fk * 1024 * 2.0 + wk * 1024
        out.append(f"FETCH_SIZE {fk:.1f} KiB raw; x2 (gfx950 correction, MI355X_MICROARCH.md) = {fk * 2048 / 1e6:.2f} MB")
        out.append(f"WRITE_SIZE {wk:.1f} KiB = {wk * 1024 / 1e6:.2f} MB")
        out.append(f"HBM bytes per launch {hbm / 1e6:.2f} MB = {hbm / frames / 1e6:.2f} MB per frame vs algorithmic 28.0 MB per frame "
                   f"(P in 12 + P out 12 + fd_falloff 4): every frame reads its input once and writes its outputs once")
        traffic = {
            "kernel": kname, "config": "C2: N=1e6, M=256, dist2=NULL, fd_falloff written, 32 frames per launch (same input mesh, own outputs)",
            "FETCH_SIZE_KiB_raw": fk, "fetch_correction": 2.0,
            "fetch_correction_note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of a coalesced "
                                     "streaming read; x2 gives P (12.0 MB) + model, which calibrates the factor",
            "WRITE_SIZE_KiB": wk, "hbm_bytes_per_launch": hbm, "frames_per_launch": frames,
            "hbm_bytes_per_frame": hbm / frames,
            "algorithmic_bytes_per_frame": {"P_in": 12e6, "P_out": 12e6, "fd_falloff": 4e6, "model_tiles": 17 * 768},
        }
        json.dump(traffic, open(os.path.join(DST, "traffic_c2.json"), "w"), indent=1)
    for agg in (sq, mf):
        for k, d in agg.items():
            for c, v in sorted(d.items()):
                out.append(f"{c} {mean(v):.1f}   (n={len(v)})")
    if sq:
        d = next(iter(sq.values()))
        waves, valu, act = mean(d["SQ_WAVES"]), mean(d["SQ_INSTS_VALU"]), mean(d["SQ_ACTIVE_INST_VALU"])
        busy = mean(d["SQ_BUSY_CYCLES"]) / 32.0           # summed over 8 XCD x 4 SE
        out.append(f"derived: {valu / waves:.0f} VALU instructions per wave; {4 * act / valu:.2f} VALU-active cycles per VALU instruction")
        out.append(f"derived: VALU active {4 * act / 1024 / busy * 100:.0f}% of the kernel's {busy:.0f} busy cycles per SIMD "
                   f"(SQ_ACTIVE_INST_VALU counts quad-cycles summed over 1024 SIMDs)")
    if mf:
        d = next(iter(mf.values()))
        if "SQ_INSTS_MFMA" in d and sq:
            busy = mean(next(iter(sq.values()))["SQ_BUSY_CYCLES"]) / 32.0
            out.append(f"derived: MFMA pipe busy {mean(d['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024 / busy * 100:.1f}% "
                       f"({mean(d['SQ_VALU_MFMA_BUSY_CYCLES']) / max(1.0, mean(d['SQ_INSTS_MFMA'])):.0f} cycles per matrix instruction)")
    bd = counters("pmc_build", "k_lu_trail")
    for k, d in bd.items():
        out.append(f"C3 build, {k} per launch: " + ", ".join(f"{c}={mean(v):.0f}" for c, v in sorted(d.items())))
    open(os.path.join(DST, f"{TAG}_pmc_eval_c2.txt"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
