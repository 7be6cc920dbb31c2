#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r4h}; mkdir -p $O
timeout 1500 python -m pytest tests/test_round4_gpu.py tests/test_engine_gpu.py tests/test_round2_gpu.py tests/test_model_surface_gpu.py -x -q -m gpu > $O/t_some.log 2>&1; echo "some rc=$?" >> $O/summary.txt
tail -3 $O/t_some.log
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -m gpu -k "dip" > $O/t_dipfull.log 2>&1; echo "dip fullsize rc=$?" >> $O/summary.txt
tail -3 $O/t_dipfull.log
timeout 600 python bench.py --workload dip --steps 200 --warmup 40 --cpu-steps 0 --f32-steps 0 --late-epoch-views 0 > $O/bench_dip.json 2> $O/bench_dip.err; echo "bench dip rc=$?" >> $O/summary.txt
cat $O/summary.txt
python -c "
import json
for f in ['bench_dip']:
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], (d.get('many_views') or {}).get('value'))
"
