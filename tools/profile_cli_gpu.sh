#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the CLI training path on a small on-disk c2 scene (kept by run_schedule --keep).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-c2}
python tools/run_schedule.py --workload $WL --views 61 --epochs 1 --keep > /tmp/rs.json 2>/dev/null
ROOT=$(ls -d /tmp/stylemesh_scene_* | tail -1)
CMD=$(python3 -c "import json; print(json.load(open('/tmp/rs.json'))['command'].replace('<scene-root>', '$ROOT').replace('--max_epochs 1', '--max_epochs 2'))")
cd /tmp
PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cliprof -o run -- python3 $CMD > /tmp/cliprof.log 2>&1
cd $R
grep -E "epoch|fit" /tmp/cliprof.log
python3 - <<PY
import csv
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open('/tmp/cliprof/run_kernel_trace.csv')))
heads=[e[0] for e in ev if 'step_begin_kernel' in e[2]]
import statistics
iv=[(b-a)/1e3 for a,b in zip(heads,heads[1:])]
iv_s=sorted(iv)
print('steps',len(heads),'interval us: median',statistics.median(iv),'p10',iv_s[len(iv)//10],'p90',iv_s[9*len(iv)//10],'mean',sum(iv)/len(iv))
long=[x for x in iv if x>5000]
print('intervals > 5 ms:',len(long),'mean',sum(long)/max(len(long),1))
busy=0; end=ev[0][0]
for s0,e0,_ in ev:
    if e0>end: busy+=e0-max(s0,end); end=e0
print('GPU busy fraction over the trace', busy/(ev[-1][1]-ev[0][0]))
# the kernels of ONE view change (between a step_begin and the next one more than 5 ms later)
import re
idx=[k for k,(a,b) in enumerate(zip(heads,heads[1:])) if b-a>5e6]
if idx:
    k=idx[len(idx)//2]; a,b=heads[k],heads[k+1]
    win=[e for e in ev if a<=e[0]<b]
    print('view change window: %.1f ms, %d kernels'%((b-a)/1e6,len(win)))
    last=a
    for s0,e0,n in win:
        if s0-last>300e3: print('   --- gap %.2f ms'%((s0-last)/1e6))
        if e0-s0>200e3: print('   %8.2f ms  +%.2f ms  %s'%((s0-a)/1e6,(e0-s0)/1e6,re.sub(r'\(.*','',n)[:60]))
        last=max(last,e0)
PY
python3 - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/cliprof/run_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time ms', tot/1e6)
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['TotalDurationNs'])/1e6,1), 'ms', round(float(r['AverageNs'])/1e3,1), 'us')
PY
