#!/bin/bash
# Run on the GPU box (gpurun).  Round 2: rocprofv3 output under gpurun_out/profiles_r02/:
#   bench/       rocprofv3 --kernel-trace --stats of the default bench command (+ its JSON line)
#   bench_line   the same command un-profiled (the numbers DESIGN.md quotes), with the CPU baseline
#   shared/      the shared-rig evaluation alone, 32 frames per launch (kernel trace + stats)
#   pmc_*/       PMC passes on that launch, one counter group per run, --kernel-trace only
# tools/summarise_profiles_r02.py turns these into the files committed under profiles/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r02
rm -rf $OUT
mkdir -p $OUT
python3 bench.py > $OUT/bench_c2_line.json 2> $OUT/bench_c2_line.err
echo "bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench_c2 -- python3 bench.py --no-cpu-baseline > $OUT/bench_c2_profiled.json 2> $OUT/bench_c2_profiled.err
echo "profiled bench exit $?"
python3 bench.py --no-cpu-baseline --eval-launch batched > $OUT/bench_c2_independent_line.json 2>/dev/null
echo "independent-rig bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shared -o shared_c2 -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/shared_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_mfma -o mfma -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_mfma.log 2>&1
python3 tests/tools/scaled_delta_parity.py > $OUT/scaled_delta_parity.txt 2>&1
python3 tests/tools/shared_eval_timing.py c2 8,16,24,32 > $OUT/shared_timing_c2.txt 2>&1
python3 tests/tools/shared_eval_timing.py c3 32 >> $OUT/shared_timing_c2.txt 2>&1
python3 tests/tools/shared_eval_timing.py c2 8,32 qnn >> $OUT/shared_timing_c2.txt 2>&1
python3 tests/tools/hbm_write_rate.py > $OUT/hbm_write_rate.txt 2>&1
python3 tests/tools/wide_variants_timing.py 0,1,17,49,16 6 > $OUT/wide_variants.txt 2>&1
[ -x tools/ubench_mfma16 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma16.hip -o tools/ubench_mfma16
tools/ubench_mfma16 > $OUT/ubench_mfma16.txt 2>&1
python3 tools/build_latency.py 256,512,2048 11 > $OUT/solver_latency.txt 2>&1
python3 bench.py --config c3 --no-cpu-baseline > $OUT/bench_c3_line.json 2>/dev/null
python3 bench.py --config c5 --no-cpu-baseline > $OUT/bench_c5_line.json 2>/dev/null
python3 tools/capture_bench.py > $OUT/capture_bench.txt 2>&1
find $OUT -name "*.csv" | wc -l
