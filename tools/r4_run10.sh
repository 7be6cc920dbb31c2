#!/bin/bash
O=gpurun_out/r4r; mkdir -p $O
cp build/ab/lib_texfwd_new.so stylemesh_amd/libstylemesh_hip.so
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_model_surface_gpu.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for t in texfwd_old texfwd_new; do
  cp build/ab/lib_$t.so stylemesh_amd/libstylemesh_hip.so
  timeout 300 python bench.py --steps 60 --warmup 10 --cpu-steps 0 --f32-steps 0 --many-views-steps 200 --late-epoch-views 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline_hbm']['kernels']; print('$t', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'tex_fwd us', k['tex_sample_fwd']['avg_us'])" | tee -a $O/texfwd_ab.txt
done
done
cp build/ab/lib_texfwd_new.so stylemesh_amd/libstylemesh_hip.so
