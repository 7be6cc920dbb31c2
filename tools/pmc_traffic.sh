#!/bin/bash
# HBM traffic of the kernels of one bench run (run on the GPU box): FETCH_SIZE and WRITE_SIZE in SEPARATE
# rocprofv3 --pmc passes (they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
export TMPDIR=/tmp
WL=${1:-c3}
mkdir -p gpurun_out/traffic
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic/$c -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 4 --warmup 1 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --no-conv-timer > $GRAFT_REPO_ROOT/gpurun_out/traffic/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/summarize_pmc.py gpurun_out/traffic gpurun_out/traffic_summary.csv
