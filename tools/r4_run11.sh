#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_engine_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2; do
for w in 0 1; do
  STYLEMESH_WEIGHTED_CLOSING=$w timeout 300 python bench.py --steps 60 --warmup 10 --cpu-steps 0 --f32-steps 0 --many-views-steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline_hbm']['kernels']; print('weighted_closing $w', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'late', d['late_epoch']['value'], 'closing', k['adam_closing(x flagged share)'], 'early us', k['adam_early(x flagged share)']['avg_us'])" | tee -a $O/weighted_closing_ab.txt
done
done
