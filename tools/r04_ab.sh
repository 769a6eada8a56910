#!/bin/bash
# A/B inside one process (tests/tools/shared_ab_timing.py) on a tuning build
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_ab.txt
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING ${EXTRA:-}" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
echo "=== $(date) EXTRA=${EXTRA:-}" | tee -a $out
timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 32 30 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 20 30 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
FD_SHARED_STAMPS=1 timeout -k 10 300 python tests/tools/shared_eval_timing.py c2 32 tps 2>&1 | grep -v amdgpu.ids | grep -v per-frame | tail -14 | tee -a $out
