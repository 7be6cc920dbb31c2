python -m pytest tests/test_round5_gpu.py -m gpu -q -x -p no:cacheprovider -k deferred 2>&1 | tail -5 | cut -c1-400
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --cpu-steps 0 --deferred-exchange > gpurun_out/bench_n2_deferred.json 2> gpurun_out/bench_n2_deferred.err
STYLEMESH_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 10 --warmup 5 --cpu-steps 0 --deferred-exchange > gpurun_out/bench_n8_deferred.json 2> gpurun_out/bench_n8_deferred.err
python - <<PY
import json
for l in ("n2_deferred","n8_deferred"):
    d=json.loads(open(f"gpurun_out/bench_{l}.json").read().strip().splitlines()[-1])
    e=d["exchange"]; print(l, d["value"], d["ms_per_step"], d.get("ranks_consistent"), e.get("bytes_per_step"), e.get("deferred_exchange"))
PY
