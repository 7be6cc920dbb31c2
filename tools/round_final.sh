#!/bin/bash
# Closing pass of a round ON THE GPU BOX: the whole GPU suite, smoke, the driver-form bench line, the launcher's N = 2 / 8
# modes over gloo on one GPU, the other workloads. Usage: round_final.sh [tag=r05z]   -> gpurun_out/<tag>/
export TMPDIR=/tmp
O=gpurun_out/${1:-r06z}; mkdir -p $O; rm -f $O/summary.txt
if [ "${SKIP_SUITE:-0}" != 1 ]; then
timeout -s KILL 1800 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "gpu tests rc=$?" >> $O/summary.txt
grep -E "passed|failed" $O/t_all.log | tail -1 >> $O/summary.txt
fi
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench default rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --cpu-steps 0 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; echo "bench n2 gloo rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --cpu-steps 0 --pipeline-exchange > $O/bench_n2_pipelined.json 2> $O/bench_n2_pipelined.err; echo "bench n2 pipelined exchange rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --cpu-steps 0 --deferred-exchange > $O/bench_n2_deferred.json 2> $O/bench_n2_deferred.err; echo "bench n2 deferred exchange rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 10 --warmup 5 --cpu-steps 0 --deferred-exchange > $O/bench_n8_deferred.json 2> $O/bench_n8_deferred.err; echo "bench n8 deferred exchange rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --replicas --cpu-steps 0 > $O/bench_n2_replicas.json 2> $O/bench_n2_replicas.err; echo "bench n2 replicas rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 10 --warmup 5 --cpu-steps 0 > $O/bench_n8_gloo.json 2> $O/bench_n8_gloo.err; echo "bench n8 gloo rc=$?" >> $O/summary.txt
timeout 300 python bench.py --gpus 2 --steps 5 > $O/bench_n2_refused.out 2>&1; echo "bench n2 nccl on one gpu rc=$? (expected 3)" >> $O/summary.txt
for wl in c2 c5 with_angle dip; do
  timeout 900 python bench.py --workload $wl --steps 200 --warmup 40 --cpu-steps 2 --late-epoch-views 0 --schedule-epochs 0 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?" >> $O/summary.txt
done
cat $O/summary.txt; cat $O/bench_default.time | tail -3
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    r = d.get("roofline") or {}
    print(f.split("/")[-1], d["value"], d["ms_per_step"], "frac", r.get("frac"), "many", (d.get("many_views") or {}).get("value"),
          "f32", (d.get("f32_mode") or {}).get("value"), "late", (d.get("late_epoch") or {}).get("value"),
          "cpu", (d.get("cpu_baseline") or {}).get("value"), "consistent", d.get("ranks_consistent"), "sched", (d.get("scene_schedule") or {}).get("live_schedule_s"))
PY
