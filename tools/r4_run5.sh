#!/bin/bash
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/${1:-r4m}; mkdir -p $O
# adam variants
for t in adam_u1 adam_u2 adam_u4 adam_u4nt adam_u8; do
  cp build/ab/lib_$t.so stylemesh_amd/libstylemesh_hip.so
  echo "=== $t" >> $O/adam_variants.txt
  timeout 120 python tools/bench_adam_flags.py 2>&1 | grep -v amdgpu.ids >> $O/adam_variants.txt
done
cp build/ab/lib_adam_u4.so stylemesh_amd/libstylemesh_hip.so
cat $O/adam_variants.txt | grep "===\|dense\|100% flagged, runs of 1024\| 40% flagged, runs of   16\| 16% flagged, runs of   16"
# c3 timeline
cd /tmp; rm -rf /tmp/tr3
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -o run -- python3 $R/bench.py --steps 30 --warmup 10 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --no-conv-timer > $O/c3_trace.log 2>&1
F=$(find /tmp/tr3 -name "run_kernel_trace.csv" | head -1)
python3 $R/tools/trace_gaps.py $F 8 4 > $O/c3_gaps.txt 2>&1
head -12 $O/c3_gaps.txt
python3 $R/tools/step_timeline.py $F 6 > $O/c3_step_timeline.txt 2>&1
