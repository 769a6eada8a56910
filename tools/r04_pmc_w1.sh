#!/bin/bash
# counter passes of the shared-rig launch (C2 x 32 frames), one counter group per pass (no tracing domains beside --kernel-trace)
set -u
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU_TRANS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $root/gpurun_out/r04/pmc_$i -o pmc --output-format csv -- python3 $root/tests/tools/shared_eval_timing.py c2 32 tps > $root/gpurun_out/r04/pmc_$i.log 2>&1 || { tail -5 $root/gpurun_out/r04/pmc_$i.log; exit 1; }
done
cd $root
python3 - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04/pmc_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "shared_w1" in r["Kernel_Name"] or "shared_wide" in r["Kernel_Name"]:
            if "pack" in r["Kernel_Name"]: continue
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d in per.values():
        for k, v in d.items(): tot[k].append(v)
with open("gpurun_out/r04/pmc_w1_summary.txt", "w") as out:
    for k in sorted(tot):
        line = f"{k:32s} {sum(tot[k]) / len(tot[k]):16.1f}   ({len(tot[k])} launches)"
        print(line); out.write(line + "\n")
PY
