#!/bin/bash
O=gpurun_out/r4t; mkdir -p $O
timeout 1200 python -m pytest tests/test_round4_gpu.py tests/test_round3_gpu.py tests/test_round2_gpu.py tests/test_engine_gpu.py tests/test_model_surface_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for rep in 1 2; do
timeout 600 python bench.py --workload dip --steps 200 --warmup 40 --cpu-steps 0 --f32-steps 0 --late-epoch-views 0 2>/dev/null | python -c "
import json,sys; L=sys.stdin.read().strip().splitlines(); print('stdout lines', len(L)); d=json.loads(L[-1]); print('dip', d['value'], d['ms_per_step'], 'many', d['many_views']['value'])"
done
timeout 600 python bench.py --steps 40 --cpu-steps 0 --f32-steps 0 2>/dev/null | python -c "
import json,sys; L=sys.stdin.read().strip().splitlines(); print('stdout lines', len(L)); d=json.loads(L[-1]); print('c3', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'late', d['late_epoch']['value'])"
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 2>/dev/null | python -c "
import json,sys; L=sys.stdin.read().strip().splitlines(); print('stdout lines', len(L)); d=json.loads(L[-1]); print('n2', d['value'], d['ranks_consistent'])"
