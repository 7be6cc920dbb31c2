#!/bin/bash
# Run ON THE GPU BOX: does side-stream overlap (loss branches, early half of the update) pay on the single-level c2 step?
run() {
  env "$@" python bench.py --workload c2 --steps 200 --warmup 20 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --no-conv-timer 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); le=d.get('late_epoch') or {}
print('$*:', d['value'], 'views/s', d['ms_per_step'], 'ms; late_epoch', le.get('value'))"
}
run A=0
run STYLEMESH_OVERLAP_MIN_PIXELS=0
run STYLEMESH_OVERLAP_MIN_PIXELS=0 STYLEMESH_EARLY_STYLE_AT=r11 STYLEMESH_EARLY_UPDATE_AT=head
run STYLEMESH_OVERLAP_MIN_PIXELS=0 STYLEMESH_SIDE_STYLE=r11,r21,r31,r41 STYLEMESH_EARLY_STYLE_AT=r41
run STYLEMESH_OVERLAP_MIN_PIXELS=0 STYLEMESH_SIDE_STYLE=r11,r21,r31 STYLEMESH_EARLY_STYLE_AT=r31
run STYLEMESH_OVERLAP_MIN_PIXELS=0 STYLEMESH_SIDE_STYLE=r11,r21 STYLEMESH_EARLY_STYLE_AT=r21
run STYLEMESH_OVERLAP_MIN_PIXELS=0 STYLEMESH_MAIN_PRIORITY=normal
run A=1
