#!/bin/bash
# PMC pass over a short bench.py run (run on the GPU box): matrix-pipe occupancy of the conv kernels in situ.
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_bench
cd /tmp
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_bench/$tag -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --no-conv-timer > $GRAFT_REPO_ROOT/gpurun_out/pmc_bench/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py gpurun_out/pmc_bench gpurun_out/pmc_bench/summary.csv
