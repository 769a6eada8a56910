#!/bin/bash
set -u
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_register_build.py tests/test_gpu_solver.py tests/test_gpu_fp32_contract.py tests/test_gpu_cook_group.py tests/test_gpu_shared.py tests/test_gpu_shared_factor.py tests/test_gpu_batch.py tests/test_gpu_deltas.py tests/test_gpu_parity.py tests/test_gpu_edges.py -x -q > gpurun_out/r04/pack_tests.txt 2>&1 || { tail -30 gpurun_out/r04/pack_tests.txt; exit 1; }
tail -3 gpurun_out/r04/pack_tests.txt
timeout -k 10 200 python tools/build_latency.py 64,256 11 2>&1 | tail -5
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-shared-factor-alternative > gpurun_out/r04/pack_b20_$i.json 2> gpurun_out/r04/pack_b20_$i.err || { tail -5 gpurun_out/r04/pack_b20_$i.err; exit 1; }
python - $i <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04/pack_b20_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("driver form:", round(d["value"]), round(d["ms_per_step"] * 20, 4), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items() if not k.startswith("end")}, round(d["host"]["us_per_group"]), round(d["roofline"]["frac"], 3))
PY
done
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04/prof_pk -o p -- python3 $root/tools/build_profile_batched.py 256 20 40 > $root/gpurun_out/r04/prof_pk.log 2>&1
python3 - $root/gpurun_out/r04/prof_pk <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.3: print("  ", r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"])
PY
