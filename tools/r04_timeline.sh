#!/bin/bash
set -u
mkdir -p gpurun_out/r04
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/r04/tl
timeout -k 10 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $root/gpurun_out/r04/tl -o t -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-shared-factor-alternative > $root/gpurun_out/r04/tl.log 2>&1 || { tail -5 $root/gpurun_out/r04/tl.log; exit 1; }
cd $root
d=$(dirname $(find gpurun_out/r04/tl -name 't_kernel_trace.csv' | head -1))
python3 tools/trace_timeline.py $d/t 20 -2 > gpurun_out/r04/timeline_20.txt
rm -rf gpurun_out/r04/tl
cat gpurun_out/r04/timeline_20.txt | head -70
