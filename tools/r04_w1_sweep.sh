#!/bin/bash
# compile-time variants of k_deform32_shared_w1 on one box (each: rebuild, then the launch alone at C2 x 32 and x 20 frames)
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_sweep.txt
for v in "$@"; do
  echo "=== variant [$v]" | tee -a $out
  FD_EXTRA_HIPCC_FLAGS="$v" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
  timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32,20 tps 2>&1 | grep shared | tee -a $out || exit 1
done
