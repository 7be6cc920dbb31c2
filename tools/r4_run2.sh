#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r4c}; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -x -q -m gpu -s > $O/t_round4.log 2>&1; echo "round4 rc=$?" >> $O/summary.txt
tail -40 $O/t_round4.log
