#!/bin/bash
# Run ON THE GPU BOX: side streams (style branches, early half of the update) confined to N compute units
# (STYLEMESH_SIDE_CUS, sm_stream_create_cu_subset) against sharing all 256 with the conv trunk. Usage: side_cu_sweep.sh [workload]
WL=${1:-c3}
for n in 0 32 64 96 128 192 0; do
STYLEMESH_SIDE_CUS=$n python bench.py --workload $WL --steps 200 --warmup 20 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --timer-every 9 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; le=d.get('late_epoch') or {}
print('side CUs $n:', d['value'], 'views/s', d['ms_per_step'], 'ms; conv', r['achieved'], 'TFLOP/s', r['avg_launch_us'], 'us; late_epoch', le.get('value'))"
done
