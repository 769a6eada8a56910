#!/bin/bash
set -u
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_gpu_cook_group.py tests/test_gpu_shared.py tests/test_gpu_shared_factor.py tests/test_gpu_batch.py tests/test_gpu_register_build.py tests/test_gpu_deltas.py -x -q > gpurun_out/r04/lean_tests.txt 2>&1 || { tail -30 gpurun_out/r04/lean_tests.txt; exit 1; }
tail -3 gpurun_out/r04/lean_tests.txt
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-shared-factor-alternative > gpurun_out/r04/lean_b20_$i.json 2> gpurun_out/r04/lean_b20_$i.err || { tail -5 gpurun_out/r04/lean_b20_$i.err; exit 1; }
python - $i <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r04/lean_b20_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("driver form:", round(d["value"]), round(d["ms_per_step"] * 20, 4), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items()}, d["host"], d["roofline"]["frac"])
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r04/lean_bdef.json 2> gpurun_out/r04/lean_bdef.err || { tail -5 gpurun_out/r04/lean_bdef.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04/lean_bdef.json").read().strip().splitlines()[-1])
print("default:", round(d["value"]), "alt", d.get("alternative") and round(d["alternative"]["value"]), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["phases_ms"].items()}, d["host"])
PY
bash tools/r04_timeline.sh > gpurun_out/r04/lean_timeline.log 2>&1; head -45 gpurun_out/r04/timeline_20.txt
