# the pipelined resident kernel: bit-identity tests, kernel stats on / off, alternating in-situ A/B
export TMPDIR=/tmp
O=gpurun_out/pipe2; mkdir -p $O
timeout 900 python -m pytest tests/test_resident_gpu.py -m gpu -q -x -p no:cacheprovider > $O/tests.txt 2>&1; echo "tests rc=$? $(tail -1 $O/tests.txt)"
grep -E "^E |Error" $O/tests.txt | head -20
R=$GRAFT_REPO_ROOT
cd /tmp
for v in on off; do
  if [ $v = off ]; then export SM_RES_PIPE_MIN=0; else unset SM_RES_PIPE_MIN; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o run -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --schedule-epochs 0 --resident-steps 0 --no-conv-timer > $R/$O/prof_$v.log 2>&1
  cp $R/$O/prof_$v/run_kernel_stats.csv $R/$O/kernel_stats_$v.csv; rm -rf $R/$O/prof_$v
  python3 $R/tools/show_kstats.py $R/$O/kernel_stats_$v.csv 23 40 | grep -E "64, 128|respipe|total"
done
unset SM_RES_PIPE_MIN
cd $R
for i in 1 2; do
  bash tools/ab_wl.sh c3 pipe_on_$i
  bash tools/ab_wl.sh c3 pipe_off_$i SM_RES_PIPE_MIN=0
done
