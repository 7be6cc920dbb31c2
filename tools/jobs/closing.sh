# closing pass on the final tree: whole GPU suite, smoke, the driver's bench form. usage: closing.sh <tag>
export TMPDIR=/tmp
O=gpurun_out/closing_${1:-x}; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/suite.txt 2>&1; echo "gpu tests rc=$? $(grep -E 'passed|failed' $O/suite.txt | tail -1)" | tee $O/summary.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time; echo "bench rc=$?" | tee -a $O/summary.txt
tail -3 $O/bench_default.time
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); ss = d["scene_schedule"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_unit"][:80], "| schedule", ss.get("measured_schedule_s"), ss.get("measured_schedule_live"), ss.get("source", "")[:40])
PY
