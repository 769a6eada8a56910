#!/bin/bash
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_abl.txt
for a in 0 1 2 3 0; do
  FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING -DFD_W1_ABL=$a" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
  echo "=== ablation $a (1: no phi vector work, 2: one product of three, 3: no LDS weight reads)" | tee -a $out
  timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 32 20 FD_SHARED_W1=1 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
  FD_SHARED_STAMPS=1 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 tps 2>&1 | grep "wave  [048]:" | tail -3 | tee -a $out
done
