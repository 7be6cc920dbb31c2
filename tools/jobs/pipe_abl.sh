# in-situ ablation of the pipelined resident kernel: kernel stats of a c3 bench run per library variant (build/ab/lib_<tag>.so)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/pipe_abl; mkdir -p $R/$O
cp $R/stylemesh_amd/libstylemesh_hip.so /tmp/lib_keep.so
for t in ${1:-base abl1 abl2 abl3 abl4}; do
  cp $R/build/ab/lib_$t.so $R/stylemesh_amd/libstylemesh_hip.so
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$t -o run -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 --no-conv-timer > $R/$O/prof_$t.log 2>&1
  cp $R/$O/prof_$t/run_kernel_stats.csv $R/$O/kernel_stats_$t.csv; rm -rf $R/$O/prof_$t
  echo "== $t"; python3 $R/tools/show_kstats.py $R/$O/kernel_stats_$t.csv 23 60 | grep -E "respipe|64, 128|total"
done
cp /tmp/lib_keep.so $R/stylemesh_amd/libstylemesh_hip.so
