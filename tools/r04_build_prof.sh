#!/bin/bash
# the build kernels: phase stamps of k_build_reg (tuning build), kernel durations of the per-frame and the shared-factor builds
set -u
mkdir -p gpurun_out/r04
root=$GRAFT_REPO_ROOT
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
FD_REG_STAMPS=1 timeout -k 10 300 python tests/tools/reg_build_check.py 2>&1 | grep -A40 "stamps" | head -60 > gpurun_out/r04/reg_stamps.txt
tail -30 gpurun_out/r04/reg_stamps.txt
python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04/prof_sf -o sf -- python3 $root/tools/shared_factor_timing.py 256 > $root/gpurun_out/r04/prof_sf.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r04/prof_sf/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["Percentage"])
PY
