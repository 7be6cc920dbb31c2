#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r4d}; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?" >> $O/summary.txt
tail -6 $O/t_all.log
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
python tools/host_profile.py dip 200 > $O/host_dip.txt 2>&1
python tools/host_profile.py c2 200 > $O/host_c2.txt 2>&1
head -3 $O/host_dip.txt $O/host_c2.txt | grep enqueue
for wl in dip c2 with_angle; do
timeout 600 python bench.py --workload $wl --steps 200 --warmup 40 --cpu-steps 0 --f32-steps 0 --late-epoch-views 0 > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?" >> $O/summary.txt
done
timeout 600 python bench.py --steps 40 --cpu-steps 0 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" >> $O/summary.txt
cat $O/summary.txt
python - <<'PY'
import json,sys
for wl in ['dip','c2','with_angle','c3']:
    try:
        d=json.loads(open(f'gpurun_out/%s/bench_{wl}.json' % sys.argv[1] if False else f'{"'$O'"}/bench_{wl}.json').read().strip().splitlines()[-1])
        print(wl, d['value'], d['ms_per_step'], 'many', (d.get('many_views') or {}).get('value'))
    except Exception as e: print(wl, 'ERR', e)
PY
