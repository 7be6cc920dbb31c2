#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
for t in adam_contig adam_strided; do
  cp build/ab/lib_$t.so stylemesh_amd/libstylemesh_hip.so
  echo "=== $t" >> $O/adam_strided.txt
  timeout 200 python tools/bench_adam_flags.py 2>&1 | grep -v "amdgpu.ids\|No tex\|No weight" >> $O/adam_strided.txt
done
cp build/ab/lib_adam_strided.so stylemesh_amd/libstylemesh_hip.so
grep "===\|dense\|16% flagged\|40% flagged\|real" $O/adam_strided.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py -x -q -m gpu -k "adam or update or sparse" 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 40 --cpu-steps 0 --f32-steps 0 --many-views-steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'late', d['late_epoch']['value']); print(json.dumps(d.get('roofline_hbm'))[:1500])"; done
