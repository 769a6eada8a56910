#!/bin/bash
# where k_deform32_shared_w1's cycles are: tuning build (-DFD_TUNING), stamps, launches without stores / without the K loop
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_diag.txt
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
echo "== stamps" | tee -a $out
FD_SHARED_STAMPS=1 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 tps 2>&1 | grep -v amdgpu.ids | tail -16 | tee -a $out
for d in 0 1 2 3; do
  echo "== FD_SHARED_DBG=$d (1: no stores, 2: no K loop)" | tee -a $out
  FD_SHARED_DBG=$d timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 tps 2>&1 | grep "shared " | tee -a $out
done
for v in "-DFD_W1_PRIO=1" "-DFD_W1_PRIO=2" "-DFD_W1_STAGGER=1" "-DFD_W1_STAGGER=2"; do
  echo "=== variant [$v]" | tee -a $out
  FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING $v" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
  timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32,20 tps 2>&1 | grep "shared " | tee -a $out || exit 1
done
