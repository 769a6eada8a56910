#!/bin/bash
# the split register build (k_reg_front1 / k_reg_front2 / k_build_reg<false>): correctness, then latency against round 3's one-workgroup form
set -u
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_register_build.py tests/test_gpu_solver.py tests/test_gpu_shared_factor.py tests/test_gpu_deltas.py tests/test_gpu_batch.py -x -q > gpurun_out/r04/split_tests.txt 2>&1 || { tail -30 gpurun_out/r04/split_tests.txt; exit 1; }
tail -3 gpurun_out/r04/split_tests.txt
timeout -k 10 300 python tests/tools/reg_build_check.py > gpurun_out/r04/split_reg_check.txt 2>&1 || { tail -20 gpurun_out/r04/split_reg_check.txt; exit 1; }
tail -8 gpurun_out/r04/split_reg_check.txt
timeout -k 10 200 python tools/build_latency.py 64,256 11 > gpurun_out/r04/split_latency.txt 2>&1; cat gpurun_out/r04/split_latency.txt
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-shared-factor-alternative > gpurun_out/r04/split_b20_$i.json 2> gpurun_out/r04/split_b20_$i.err || { tail -5 gpurun_out/r04/split_b20_$i.err; exit 1; }
python - $i <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04/split_b20_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("driver form:", round(d["value"]), d["ms_per_step"] * 20, {k: round(v, 4) for k, v in d["phases_ms"].items()}, d["host"])
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r04/split_bdef.json 2> gpurun_out/r04/split_bdef.err || { tail -5 gpurun_out/r04/split_bdef.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/split_bdef.json").read().strip().splitlines()[-1])
print("default:", round(d["value"]), "alt", d.get("alternative") and round(d["alternative"]["value"]), {k: round(v, 4) for k, v in d["phases_ms"].items()}, d["host"])
PY
