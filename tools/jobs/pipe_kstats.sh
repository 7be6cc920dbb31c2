# kernel stats of a c3 bench run with the pipelined resident kernel on / off (usage: pipe_kstats.sh [variants="on off"])
export TMPDIR=/tmp
O=gpurun_out/pipe3; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
for v in ${1:-on off}; do
  if [ $v = off ]; then export SM_RES_PIPE_MIN=0; else unset SM_RES_PIPE_MIN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o run -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 --no-conv-timer > $R/$O/prof_$v.log 2>&1
  cp $R/$O/prof_$v/run_kernel_stats.csv $R/$O/kernel_stats_$v.csv; rm -rf $R/$O/prof_$v
  python3 $R/tools/show_kstats.py $R/$O/kernel_stats_$v.csv 23 40 | grep -E "64, 128|respipe|total"
done
