#!/bin/bash
set -u
mkdir -p gpurun_out/r04
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
for i in 1 2 3; do
FD_COOK_TIMING=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-shared-factor-alternative 2> gpurun_out/r04/cook_timing_$i.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step']*20,4), round(d['host']['us_per_group']))"
grep "cook_group host" gpurun_out/r04/cook_timing_$i.err | tail -3
done
