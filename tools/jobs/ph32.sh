# 32-channel phases of the resident kernel at 3 / 4 waves per SIMD: bit-identity tests, kernel stats in situ (pipelined kernel off)
export TMPDIR=/tmp SM_RES_PIPE_MIN=0
R=$GRAFT_REPO_ROOT; O=gpurun_out/ph32; mkdir -p $R/$O
cp $R/stylemesh_amd/libstylemesh_hip.so /tmp/lib_keep.so
for t in ${1:-ph32w4 ph32w3}; do
  cp $R/build/ab/lib_$t.so $R/stylemesh_amd/libstylemesh_hip.so
  cd $R; echo "== $t"; timeout -s KILL 150 python -m pytest tests/test_resident_gpu.py -m gpu -q -x -p no:cacheprovider -k "not pipe or False" 2>&1 | tail -1
  cd /tmp
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$t -o run -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 --no-conv-timer > $R/$O/prof_$t.log 2>&1
  cp $R/$O/prof_$t/run_kernel_stats.csv $R/$O/kernel_stats_$t.csv; rm -rf $R/$O/prof_$t
  python3 $R/tools/show_kstats.py $R/$O/kernel_stats_$t.csv 23 60 | grep -E "64, 128|total"
  tail -1 $R/$O/prof_$t.log | cut -c1-120
done
cp /tmp/lib_keep.so $R/stylemesh_amd/libstylemesh_hip.so
