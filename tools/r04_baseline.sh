#!/bin/bash
# round-4 baseline of the tree as round 3 left it + the new call-path tests (run through gpurun)
set -u
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_cook_group.py tests/test_gpu_shared.py -x -q -m gpu > gpurun_out/r04/t_cook.log 2>&1 || { tail -30 gpurun_out/r04/t_cook.log; exit 1; }
tail -3 gpurun_out/r04/t_cook.log
timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 20,32 tps > gpurun_out/r04/shared_timing_base.txt 2>&1 || exit 1
cat gpurun_out/r04/shared_timing_base.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/bench20_base.json 2> gpurun_out/r04/bench20_base.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_base.json 2> gpurun_out/r04/bench_base.err || exit 1
python - <<'PY'
import json
for f in ("bench20_base","bench_base"):
    d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), d["ms_per_step"], d["phases_ms"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
PY
