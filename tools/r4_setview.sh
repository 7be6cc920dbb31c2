#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r4b}; mkdir -p $O
cd /tmp
for wl in c3 dip; do
  rm -rf /tmp/sv_$wl
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/sv_$wl -o run -- python3 $R/tools/setview_breakdown.py run $wl 4 > $O/setview_$wl.log 2>&1
  python3 $R/tools/setview_breakdown.py parse $(find /tmp/sv_$wl -name 'run_kernel_trace.csv' | head -1) 4 >> $O/setview_$wl.log 2>&1
done
rm -rf /tmp/tr_dip
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_dip -o run -- python3 $R/bench.py --workload dip --steps 40 --warmup 10 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --no-conv-timer > $O/dip_trace.log 2>&1
python3 $R/tools/step_timeline.py $(find /tmp/tr_dip -name 'run_kernel_trace.csv' | head -1) 6 > $O/dip_step_timeline.txt 2>&1
tail -3 $O/setview_c3.log
