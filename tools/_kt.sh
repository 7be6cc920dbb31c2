export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in base; do
cp build/ab/lib_$t.so stylemesh_amd/libstylemesh_hip.so
mkdir -p gpurun_out/kt_$t; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_$t -o run -- python3 $R/tools/bench_gram_group.py 0.8 > /dev/null 2>&1
cd $R
echo "=== $t"
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open('gpurun_out/kt_$t/run_kernel_trace.csv')) if 'gram_group' in r['Kernel_Name']]
# calls come in blocks of 13 per configuration (3 warm + 10 timed); print duration sequence compactly
import collections
seq=[(r['Kernel_Name'][10:36], int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',0)), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
agg=collections.OrderedDict()
for k,g,d in seq:
    agg.setdefault((k,g),[]).append(d)
for (k,g),v in agg.items():
    print(f"{k} grid {g:8d} n {len(v):3d} avg {sum(v)/len(v):7.1f} us min {min(v):7.1f}")
PY
rm -f gpurun_out/kt_$t/run_kernel_trace.csv
done
