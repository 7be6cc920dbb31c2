export TMPDIR=/tmp
for w in none gram_bwd gram_fwd gram_bwd,gram_fwd none; do
  timeout -s KILL 200 python tools/whatif.py $w --steps 200 --warmup 40 --cpu-steps 0 --f32-steps 0 --many-views-steps 0 --late-epoch-views 0 --resident-steps 0 --schedule-epochs 0 > gpurun_out/whatif_$w.json 2> gpurun_out/whatif_$w.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/whatif_$w.json").read().strip().splitlines()[-1]); print("$w", d["value"], d["ms_per_step"])
except Exception as e: print("$w ERR", e)
PY
done
