cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $grp --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_shared_$(echo $grp | cut -c4-12) -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tests/tools/shared_eval_timing.py c2 32 > /dev/null 2>&1
done
