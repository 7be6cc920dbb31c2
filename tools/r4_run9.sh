#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
for rep in 1 2; do
for pen in 0 2 4 8; do
  SM_CONV_SPLIT_PENALTY=$pen timeout 300 python bench.py --steps 60 --warmup 10 --cpu-steps 0 --f32-steps 0 --many-views-steps 200 --late-epoch-views 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 penalty $pen', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'frac', d['roofline']['frac'])" | tee -a $O/split_penalty_ab.txt
done
done
for pen in 0 2 4 8; do
  SM_CONV_SPLIT_PENALTY=$pen timeout 300 python bench.py --workload c2 --steps 200 --warmup 40 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 penalty $pen', d['value'], d['ms_per_step'])" | tee -a $O/split_penalty_ab.txt
done
