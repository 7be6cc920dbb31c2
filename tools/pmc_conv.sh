#!/bin/bash
# PMC pass over the conv micro-benchmark (run on the GPU box). Counters in separate runs per the guide.
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
cd /tmp
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_conv.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/pmc | head -30
