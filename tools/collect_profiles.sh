#!/bin/bash
# Run on the GPU box (gpurun).  Writes rocprofv3 summaries under gpurun_out/profiles_r01/:
#   bench_c2_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the default bench command
#   evalonly_* / traffic        evaluation kernel alone (isolated launches), kernel trace + PMC passes
# PMC passes are separate runs with --kernel-trace only (no other tracing domain).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r01
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench_c2 -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_c2.log 2>&1
echo "bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench1 -o bench_c2_inflight1 -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight 1 > $OUT/bench_c2_inflight1.log 2>&1
echo "bench inflight1 exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/evalonly -o evalonly_c2 -- python tools/eval_variants.py 102 c2 > $OUT/evalonly_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python tools/eval_variants.py 102 c2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python tools/eval_variants.py 102 c2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python tools/eval_variants.py 102 c2 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_mfma -o mfma -- python tools/eval_variants.py 102 c3 > $OUT/pmc_mfma.log 2>&1
find $OUT -name "*.csv" | head -30
