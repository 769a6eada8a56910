#!/bin/bash
# Run on the GPU box (gpurun).  Writes rocprofv3 output under gpurun_out/profiles_r01/:
#   bench/      rocprofv3 --kernel-trace --stats of the default bench command (+ its JSON line)
#   bench1/     same with one frame per build and one lane (no batching, no overlap)
#   evalonly/   evaluation alone, as the benchmark launches it: 32 frames per launch (kernel trace)
#   pmc_*/      PMC passes on that launch, one counter group per run, --kernel-trace only
#               (no other tracing domain)
# tools/summarise_profiles.py turns these into the files committed under profiles/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/profiles_r01
rm -rf $OUT
mkdir -p $OUT
VAR=${FD_PROFILE_VARIANT:-202}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench_c2 -- python bench.py --no-cpu-baseline > $OUT/bench_c2.log 2>&1
echo "bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench1 -o bench_c2_single -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --inflight 1 --lanes 1 > $OUT/bench_c2_single.log 2>&1
echo "bench single exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/evalonly -o evalonly_c2 -- python tests/tools/batch_eval_timing.py 32 batched > $OUT/evalonly_c2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python tests/tools/batch_eval_timing.py 32 batched > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python tests/tools/batch_eval_timing.py 32 batched > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python tests/tools/batch_eval_timing.py 32 batched > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_mfma -o mfma -- python tests/tools/batch_eval_timing.py 32 batched > $OUT/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_build -o build -- python tests/tools/eval_variants.py $VAR c3 > $OUT/pmc_build.log 2>&1
find $OUT -name "*.csv" | wc -l
