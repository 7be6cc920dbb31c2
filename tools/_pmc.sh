export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gram; mkdir -p gpurun_out/pmc_gram
cd /tmp
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gram/$tag -o run -- python3 $R/tools/bench_gram_group.py 0.8 > $R/gpurun_out/pmc_gram/$tag.log 2>&1
done
cd $R
python3 tools/summarize_pmc.py gpurun_out/pmc_gram gpurun_out/pmc_gram/summary.csv
find gpurun_out/pmc_gram -name "*.csv" ! -name summary.csv -delete
