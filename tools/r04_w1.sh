#!/bin/bash
# round 4: the one-tile kernel (k_deform32_shared_w1) against round 3's two-tile kernel, same box, same process order
set -u
mkdir -p gpurun_out/r04
for w in 1 0 1 0; do
  echo "== FD_SHARED_W1=$w" >> gpurun_out/r04/w1_timing.txt
  FD_SHARED_W1=$w timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 20,24,32 tps 2>&1 | grep shared >> gpurun_out/r04/w1_timing.txt || exit 1
done
FD_SHARED_W1=1 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 qnn 2>&1 | grep shared >> gpurun_out/r04/w1_timing.txt
FD_SHARED_W1=0 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 qnn 2>&1 | grep shared >> gpurun_out/r04/w1_timing.txt
FD_SHARED_W1=1 timeout -k 10 300 python tests/tools/shared_eval_timing.py c3 32 tps 2>&1 | grep shared >> gpurun_out/r04/w1_timing.txt
FD_SHARED_W1=0 timeout -k 10 300 python tests/tools/shared_eval_timing.py c3 32 tps 2>&1 | grep shared >> gpurun_out/r04/w1_timing.txt
cat gpurun_out/r04/w1_timing.txt
timeout -k 10 900 python -m pytest tests/test_gpu_shared.py tests/test_gpu_bench_launch.py tests/test_gpu_cook_group.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04/t_w1.log 2>&1; tail -15 gpurun_out/r04/t_w1.log
