#!/bin/bash
# Run on the GPU box via gpurun: smoke, GPU tests, microbench, short bench.
# Stops at the first step that was killed or timed out (never starts another GPU step after that).
set -u
mkdir -p gpurun_out
run() {
    local name=$1; shift
    echo "=== $name: $*" | tee -a gpurun_out/ci.log
    timeout -k 10 "${STEP_TIMEOUT:-420}" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "=== $name exit $rc" | tee -a gpurun_out/ci.log
    tail -n "${TAIL:-15}" "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / killed: stopping"; exit $rc; fi
    return 0
}
: > gpurun_out/ci.log
for step in "$@"; do
    case $step in
        smoke) run smoke python __graft_entry__.py smoke ;;
        tests) run tests python -m pytest tests -m gpu -x -q -rA --durations=15 ;;
        tests_all) run tests_all python -m pytest tests -m gpu -q -rA --durations=15 ;;
        ubench) run ubench ./tools/ubench_valu ;;
        bench) run bench python bench.py --steps 100 --warmup 10 ;;
        bench_nocpu) run bench_nocpu python bench.py --steps 100 --warmup 10 --no-cpu-baseline ;;
        prof) mkdir -p gpurun_out/prof; (cd /tmp && export TMPDIR=/tmp && run_prof() { :; }); 
              run prof rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline ;;
        *) echo "unknown step $step" ;;
    esac
done
