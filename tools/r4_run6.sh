#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?" >> $O/summary.txt
tail -4 $O/t_all.log
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
timeout -s KILL 1500 bash tools/profile_round4.sh r4n_c3 c3 3 > $O/prof_c3.log 2>&1; echo "prof c3 rc=$?" >> $O/summary.txt
timeout -s KILL 600 bash tools/profile_round4.sh r4n_c2 c2 3 > $O/prof_c2.log 2>&1; echo "prof c2 rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -5 $O/prof_c3.log
