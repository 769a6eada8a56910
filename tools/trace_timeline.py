"""Interleaved HIP-API / kernel timeline of one group of frames from a rocprofv3 trace of bench.py:
    rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python tools/trace_timeline.py DIR/t [frames_in_group] [which]
Times in microseconds relative to the start of the group's build kernel (the `which`-th k_build_reg launch with that many models,
default -1 = the last; a one-group run of bench.py repeats its group once, untimed, with events around the builds: its TIMED group
is -2)."""
import csv
import re
import sys


def main():
    base = sys.argv[1]
    frames = sys.argv[2] if len(sys.argv) > 2 else "20"
    K = list(csv.DictReader(open(base + "_kernel_trace.csv")))
    A = list(csv.DictReader(open(base + "_hip_api_trace.csv")))
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    builds = [k for k in K if "k_build_reg" in k["Kernel_Name"] and k["Grid_Size_Z"] == frames]
    builds.sort(key=lambda k: int(k["Start_Timestamp"]))
    t0 = int(builds[which]["Start_Timestamp"])
    lo, hi = t0 - 300_000, t0 + 520_000
    ev = []
    for a in A:
        s = int(a["Start_Timestamp"])
        if lo <= s <= hi and a["Function"] not in ("hipGetDevice", "hipGetLastError"):
            ev.append((s, int(a["End_Timestamp"]), "API " + a["Function"]))
    for k in K:
        s = int(k["Start_Timestamp"])
        if lo <= s <= hi:
            n = re.search(r"(k_\w+|__amd_\w+)", k["Kernel_Name"])
            ev.append((s, int(k["End_Timestamp"]), "   GPU " + (n.group(1) if n else k["Kernel_Name"][:30])))
    ev.sort()
    print("start us  duration us")
    for s, e, n in ev:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")


if __name__ == "__main__":
    main()
