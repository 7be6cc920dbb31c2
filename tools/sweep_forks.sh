#!/bin/bash
# Run ON THE GPU BOX: where to fork the HBM-bound side work of the c3 step (engine.early_update_at / early_style_at).
for cfg in "head r11" "r31 r31" "r21 r21" "r31 r11" "head r31" "r41 r31" "r32 r32" "r41 r41"; do
  set -- $cfg
  STYLEMESH_EARLY_UPDATE_AT=$1 STYLEMESH_EARLY_STYLE_AT=$2 python bench.py --steps 100 --warmup 20 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --no-conv-timer ${LATE:-} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); le=d.get('late_epoch') or {}
print('update_at $1 style_at $2:', d['value'], 'views/s', d['ms_per_step'], 'ms; late_epoch', le.get('value'), le.get('ms_per_step'))"
done
