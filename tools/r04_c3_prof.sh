#!/bin/bash
set -u
mkdir -p gpurun_out/r04
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04/c3stats -o c3 -- python3 $root/bench.py --config c3 --no-cpu-baseline --steps 384 --warmup 96 > $root/gpurun_out/r04/c3stats.json 2> $root/gpurun_out/r04/c3stats.err
python3 - $root/gpurun_out/r04/c3stats <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 1: print("  ", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $root/gpurun_out/r04/c3pmc -o c3 -- python3 $root/bench.py --config c3 --no-cpu-baseline --steps 192 --warmup 96 > /dev/null 2>&1
python3 - $root/gpurun_out/r04/c3pmc <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-40:]
    tot[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in tot.items():
    if "GRBM_GUI_ACTIVE" in d and len(d["GRBM_GUI_ACTIVE"]) > 20:
        m = {c: sum(v) / len(v) for c, v in d.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        print(k, "launches", len(d["GRBM_GUI_ACTIVE"]), "cycles", round(cyc), "mfma", round(m.get("SQ_INSTS_MFMA", 0)), "pipe busy", round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024), 3), "wait_inst", round(m.get("SQ_WAIT_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)), 3))
PY
