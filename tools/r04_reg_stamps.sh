#!/bin/bash
set -u
mkdir -p gpurun_out/r04
FD_EXTRA_HIPCC_FLAGS="-DFD_TUNING ${EXTRA:-}" python -c "import facedeform_amd._build as b; b.build(force=True)" || exit 1
FD_REG_STAMPS=1 timeout -k 10 120 python tools/reg_stamps.py 256 2>&1 | grep -v amdgpu | tail -21 | tee gpurun_out/r04/reg_stamps_256.txt
