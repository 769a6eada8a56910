#!/bin/bash
set -u
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_shared.py tests/test_gpu_bench_launch.py tests/test_gpu_cook_group.py tests/test_gpu_configs.py tests/test_gpu_batch.py -x -q -m gpu > gpurun_out/r04/t_w1c.log 2>&1; tail -5 gpurun_out/r04/t_w1c.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench20_w1.json 2> gpurun_out/r04/bench20_w1.err || { tail -5 gpurun_out/r04/bench20_w1.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_w1.json 2> gpurun_out/r04/bench_w1.err || { tail -5 gpurun_out/r04/bench_w1.err; exit 1; }
timeout -k 10 300 python bench.py --no-cpu-baseline --config c3 --steps 2000 --warmup 200 > gpurun_out/r04/bench_c3_w1.json 2> gpurun_out/r04/bench_c3_w1.err || { tail -5 gpurun_out/r04/bench_c3_w1.err; exit 1; }
python - <<'PY'
import json
for f in ("bench20_w1","bench_w1","bench_c3_w1"):
    d=json.loads(open(f"gpurun_out/r04/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]), d["ms_per_step"], d["phases_ms"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], round(d["roofline"]["frac"],3), d["config"]["pipeline_build"][:40], d["roofline"]["mfma"]["executed"])
PY
