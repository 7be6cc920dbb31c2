#!/bin/bash
# usage: ab_c2.sh <workload> <label> [ENV=VAL ...]
wl=$1; label=$2; shift 2
env "$@" python bench.py --workload $wl --steps 200 --warmup 40 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --resident-steps 0 --schedule-epochs 0 > gpurun_out/ab_${wl}_${label}.json 2> gpurun_out/ab_${wl}_${label}.err
python - <<PY
import json
d=json.load(open("gpurun_out/ab_${wl}_${label}.json"))
r=d["roofline"]
print("${wl} ${label}", d["value"], d["ms_per_step"], "frac", r["frac"], "avg_launch_us", r["avg_launch_us"])
PY
