R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64" "FETCH_SIZE" "WRITE_SIZE"; do
  d=$R/gpurun_out/pp_$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $R/scripts/pmc_calib.py > /dev/null 2>&1
done
cd $R
python scripts/pmc_summary.py gpurun_out/pp_* | grep "panel" > gpurun_out/panel_counters.txt
rm -rf gpurun_out/pp_*
cat gpurun_out/panel_counters.txt
