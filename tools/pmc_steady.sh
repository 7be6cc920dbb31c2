#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
cd /tmp
for set in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_conv_steady.py > $GRAFT_REPO_ROOT/gpurun_out/pmc2/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py gpurun_out/pmc2 gpurun_out/pmc2_summary.csv
