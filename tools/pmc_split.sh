#!/bin/bash
# PMC pass over the split-conv micro-benchmark (run on the GPU box). Counters in separate runs per the guide.
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_split
cd /tmp
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_split/$tag -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_split.py 3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_split/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py gpurun_out/pmc_split gpurun_out/pmc_split/summary.csv
