# SQ / LDS / memory counters of the general-matrix CSR kernels on the 216^3 Laplacian: KSGPU_SPMV = csr (wave form), csrblock, sell
R=$GRAFT_REPO_ROOT
FMTS=${1:-"csr csrblock sell"}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_csr.txt
: > $OUT
for fmt in $FMTS; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_BUSY_avr TA_TA_BUSY_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    d=$R/gpurun_out/pc_${fmt}_$(echo $set | tr ' ' '_' | cut -c1-30)
    rm -rf $d
    KSGPU_SPMV=$fmt rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $R/scripts/pmc_csr_run.py > /dev/null 2>&1
  done
  echo "=== KSGPU_SPMV=$fmt" >> $OUT
  python3 $R/scripts/pmc_summary.py $R/gpurun_out/pc_${fmt}_* | grep "k_spmv" >> $OUT
  rm -rf $R/gpurun_out/pc_${fmt}_*
done
cat $OUT
