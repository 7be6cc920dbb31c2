# Gram backward staged once per chunk: tests, then in-situ A/B (alternating) and kernel stats of both builds
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/gram_ab; mkdir -p $R/$O
cd $R
timeout -s KILL 600 python -m pytest tests/test_kernels_gpu.py tests/test_resident_gpu.py tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "gram or Gram or engine_step or golden" > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -1 $O/tests.txt)"
cp $R/stylemesh_amd/libstylemesh_hip.so /tmp/lib_keep.so
for i in 1 2; do for t in gramold gramnew; do
  cp $R/build/ab/lib_$t.so $R/stylemesh_amd/libstylemesh_hip.so
  bash tools/ab_wl.sh c3 ${t}_$i
done; done
for t in gramold gramnew; do
  cp $R/build/ab/lib_$t.so $R/stylemesh_amd/libstylemesh_hip.so
  cd /tmp
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$t -o run -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 --no-conv-timer > $R/$O/prof_$t.log 2>&1
  cp $R/$O/prof_$t/run_kernel_stats.csv $R/$O/kernel_stats_$t.csv; rm -rf $R/$O/prof_$t
  cd $R; echo "== $t"; python3 tools/show_kstats.py $O/kernel_stats_$t.csv 23 60 | grep -E "gram_|total"
done
cp /tmp/lib_keep.so $R/stylemesh_amd/libstylemesh_hip.so
