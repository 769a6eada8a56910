#!/bin/bash
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_round2.txt
for v in "-DFD_TUNING" "-DFD_TUNING -DFD_W1_WAVES=8" ""; do
  echo "=== variant [$v]" | tee -a $out
  FD_EXTRA_HIPCC_FLAGS="$v" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
  if [ -n "$v" ]; then
    FD_SHARED_STAMPS=1 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 tps 2>&1 | grep -v amdgpu.ids | grep -v per-frame | tail -14 | tee -a $out
  fi
  timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32,20 tps 2>&1 | grep "shared " | tee -a $out || exit 1
done
FD_SHARED_W1=0 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32,20 tps 2>&1 | grep "shared " | tee -a $out
timeout -k 10 900 python -m pytest tests/test_gpu_shared.py tests/test_gpu_bench_launch.py tests/test_gpu_cook_group.py -x -q -m gpu > gpurun_out/r04/t_w1b.log 2>&1; tail -5 gpurun_out/r04/t_w1b.log
