#!/bin/bash
# On the GPU box: alternate environment settings over a bench workload, two rounds. Usage: WL=c2 ab_env.sh "A=1" "A=0 B=2" ...
R=$GRAFT_REPO_ROOT
for round in 1 2; do
for e in "$@"; do
  env $e python3 $R/bench.py --workload ${WL:-c3} --steps ${STEPS:-40} --warmup 5 --cpu-steps 0 --f32-steps 0 --late-epoch-views 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${WL:-c3} [$e]', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'frac', d['roofline']['frac'])"
done; done
