#!/bin/bash
set -u
mkdir -p gpurun_out/r04
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for what in "build_profile.py 256 cholesky 40" "build_profile_batched.py 256 20 40" "build_profile_batched.py 256 32 40"; do
  tag=$(echo $what | tr ' ./' '___')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04/prof_$tag -o p -- python3 $root/tools/$what > $root/gpurun_out/r04/prof_$tag.log 2>&1 || { tail -5 $root/gpurun_out/r04/prof_$tag.log; exit 1; }
  echo "== $what"
  python3 - $root/gpurun_out/r04/prof_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3: print("  ", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"])
PY
done
