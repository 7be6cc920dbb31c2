#!/bin/bash
O=gpurun_out/r4p; mkdir -p $O
for rep in 1 2; do
for t in adam_contig adam_strided; do
  cp build/ab/lib_$t.so stylemesh_amd/libstylemesh_hip.so
  timeout 300 python bench.py --steps 60 --warmup 10 --cpu-steps 0 --f32-steps 0 --many-views-steps 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline_hbm']['kernels']; print('$t', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'late', d['late_epoch']['value'], 'closing', k['adam_closing(x flagged share)']['avg_us'], 'early', k['adam_early(x flagged share)']['avg_us'])" | tee -a $O/adam_insitu_ab.txt
done
done
cp build/ab/lib_adam_strided.so stylemesh_amd/libstylemesh_hip.so
