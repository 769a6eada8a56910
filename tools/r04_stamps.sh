#!/bin/bash
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_stamps.txt
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING ${EXTRA:-}" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
for d in "$@"; do
  echo "=== FD_SHARED_DBG=$d EXTRA=${EXTRA:-}" | tee -a $out
  FD_SHARED_DBG=$d FD_SHARED_STAMPS=1 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 tps 2>&1 | grep -v amdgpu.ids | grep -v per-frame | tail -14 | tee -a $out
done
