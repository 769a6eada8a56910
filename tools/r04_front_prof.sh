#!/bin/bash
set -u
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_register_build.py tests/test_gpu_solver.py tests/test_gpu_shared_factor.py tests/test_gpu_edges.py tests/test_gpu_deltas.py -x -q 2>&1 | tail -2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for what in "build_profile.py 256 cholesky 40" "build_profile_batched.py 256 20 40"; do
  tag=$(echo $what | tr ' ./' '___')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04/prof_$tag -o p -- python3 $root/tools/$what > $root/gpurun_out/r04/prof_$tag.log 2>&1 || { tail -5 $root/gpurun_out/r04/prof_$tag.log; exit 1; }
  echo "== $what"
  python3 - $root/gpurun_out/r04/prof_$tag <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 3: print("  ", r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done
cd $root
python tools/build_latency.py 256 11 2>&1 | tail -2
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-shared-factor-alternative 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step']*20,4), round(d['host']['us_per_group']), round(d['phases_ms']['single_build'],4))"; done
