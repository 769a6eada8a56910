#!/bin/bash
# Run on the GPU box (gpurun).  Round 4: rocprofv3 output under gpurun_out/profiles_r04/:
#   bench_line            the default bench command (the numbers DESIGN.md quotes), with the CPU baseline
#   bench/                rocprofv3 --kernel-trace --stats of the same command
#   bench20_line          the driver's form: --steps 20 --warmup 5
#   shared/, pmc_*/       the shared-rig evaluation alone, 32 frames per launch: kernel trace, PMC passes (one group per run)
#   reg/, pmc_reg_*/      the register-resident build in its one-workgroup form (40 builds of one model at M = 256 in a batch that leaves CUs to
#                         its builds, as bench.py's pipeline does): kernel trace, PMC passes
#   regsplit/, regsplit20/ the same model through the parallel front end (a single fd_build; a batch of 20): kernel trace
# tools/summarise_profiles_r04.py turns these into the files committed under profiles/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r04
rm -rf $OUT
mkdir -p $OUT
python3 bench.py > $OUT/bench_c2_line.json 2> $OUT/bench_c2_line.err
echo "bench exit $?"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c2_20_line.json 2>/dev/null
echo "bench 20 exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench_c2 -- python3 bench.py --no-cpu-baseline > $OUT/bench_c2_profiled.json 2> $OUT/bench_c2_profiled.err
echo "profiled bench exit $?"
python3 bench.py --no-cpu-baseline --eval-launch batched > $OUT/bench_c2_independent_line.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shared -o shared_c2 -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/shared_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -o mfma -- python3 tests/tools/shared_eval_timing.py c2 32 > $OUT/pmc_mfma.log 2>&1
# the driver's launch shape (20 frames): the same byte counters, so that the line of `--steps 20` carries measured traffic too
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch20 -o fetch -- python3 tests/tools/shared_eval_timing.py c2 20 > $OUT/pmc_fetch20.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write20 -o write -- python3 tests/tools/shared_eval_timing.py c2 20 > $OUT/pmc_write20.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma20 -o mfma -- python3 tests/tools/shared_eval_timing.py c2 20 > $OUT/pmc_mfma20.log 2>&1
echo "shared passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/reg -o reg -- python3 tools/build_profile_batched.py 256 1 40 224 > $OUT/reg.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_reg_mfma -o mfma -- python3 tools/build_profile_batched.py 256 1 40 224 > $OUT/pmc_reg_mfma.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_reg_sq -o sq -- python3 tools/build_profile_batched.py 256 1 40 224 > $OUT/pmc_reg_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_reg_fetch -o fetch -- python3 tools/build_profile_batched.py 256 1 40 224 > $OUT/pmc_reg_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_reg_write -o write -- python3 tools/build_profile_batched.py 256 1 40 224 > $OUT/pmc_reg_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/regsplit -o regsplit -- python3 tools/build_profile.py 256 cholesky 40 > $OUT/regsplit.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/regsplit20 -o regsplit20 -- python3 tools/build_profile_batched.py 256 20 40 > $OUT/regsplit20.log 2>&1
echo "register build passes done"
python3 tests/tools/reg_build_check.py > $OUT/reg_build_check.txt 2>&1
python3 tests/tools/shared_eval_timing.py c2 8,16,20,24,32 > $OUT/shared_timing_c2.txt 2>&1
python3 tests/tools/shared_eval_timing.py c2 8,32 qnn >> $OUT/shared_timing_c2.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shared20 -o shared_c2_20 -- python3 tests/tools/shared_eval_timing.py c2 20 > $OUT/shared_c2_20.log 2>&1
python3 tools/build_latency.py 256,512,2048 11 > $OUT/solver_latency.txt 2>&1
python3 tools/qnn_latency.py > $OUT/qnn_latency.txt 2>&1
[ -x tools/ubench_f64 ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/ubench_f64
tools/ubench_f64 > $OUT/ubench_f64.txt 2>&1
python3 bench.py --config c3 --no-cpu-baseline > $OUT/bench_c3_line.json 2>/dev/null
python3 bench.py --config c5 --no-cpu-baseline > $OUT/bench_c5_line.json 2>/dev/null
python3 bench.py --no-cpu-baseline --eval-cus 192 > $OUT/bench_c2_cus192.json 2>/dev/null
python3 bench.py --no-cpu-baseline --eval-cus 256 > $OUT/bench_c2_cus256.json 2>/dev/null
python3 bench.py --no-cpu-baseline --build chain > $OUT/bench_c2_chain.json 2>/dev/null
python3 bench.py --no-cpu-baseline --build one-workgroup > $OUT/bench_c2_onewg.json 2>/dev/null
python3 tools/shared_factor_timing.py 256 > $OUT/shared_factor_timing.txt 2>&1
python3 tools/host_path_timing.py > $OUT/host_path.txt 2>&1
find $OUT -name "*.csv" | wc -l
