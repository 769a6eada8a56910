#!/bin/bash
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_ab.txt
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING ${EXTRA:-}" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
echo "=== $(date) EXTRA=${EXTRA:-} F=${F:-32}" | tee -a $out
timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 ${F:-32} ${ROUNDS:-30} "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
