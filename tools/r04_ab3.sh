#!/bin/bash
set -u
mkdir -p gpurun_out/r04
out=gpurun_out/r04/w1_ab.txt
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
echo "=== $(date) qnn / c3 / c5-range" | tee -a $out
FD_AB_MODEL=qnn timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 32 15 FD_SHARED_W1=1 FD_SHARED_W1=0 2>&1 | grep -v amdgpu.ids | sed 's/^/qnn /' | tee -a $out || exit 1
FD_AB_MODEL=qnn timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 20 15 FD_SHARED_W1=1 FD_SHARED_W1=0 2>&1 | grep -v amdgpu.ids | sed 's/^/qnn /' | tee -a $out || exit 1
timeout -k 10 600 python tests/tools/shared_ab_timing.py c3 32 8 FD_SHARED_W1=1 FD_SHARED_W1=0 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
timeout -k 10 600 python tests/tools/shared_ab_timing.py c5 32 10 FD_SHARED_W1=1 FD_SHARED_W1=0 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 24 15 FD_SHARED_W1=1 FD_SHARED_W1=0 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
timeout -k 10 600 python tests/tools/shared_ab_timing.py c2 28 15 FD_SHARED_W1=1 FD_SHARED_W1=0 2>&1 | grep -v amdgpu.ids | tee -a $out || exit 1
