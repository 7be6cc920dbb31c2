#!/bin/bash
# On the GPU box: alternate variant libraries over the c3 bench (200-step many-views leg), two rounds.
R=$GRAFT_REPO_ROOT
for round in 1 2; do
for t in "$@"; do
  cp $R/build/ab/lib_$t.so $R/stylemesh_amd/libstylemesh_hip.so
  python3 $R/bench.py --workload ${WL:-c3} --steps 40 --warmup 5 --cpu-steps 0 --f32-steps 0 --late-epoch-views 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t', d['value'], d['ms_per_step'], 'many', d['many_views']['value'], 'frac', d['roofline']['frac'])"
done; done
