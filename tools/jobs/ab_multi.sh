# alternating in-situ A/B of library variants over workloads. usage: ab_multi.sh "<variants>" "<workloads>" [repeats=2] [ENV=VAL ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
V=$1; W=$2; N=${3:-2}; shift 3
cp $R/stylemesh_amd/libstylemesh_hip.so /tmp/lib_keep.so
for wl in $W; do for i in $(seq 1 $N); do for t in $V; do
  cp $R/build/ab/lib_$t.so $R/stylemesh_amd/libstylemesh_hip.so
  timeout -s KILL 200 bash tools/ab_wl.sh $wl ${t}_$i "$@"
done; done; done
cp /tmp/lib_keep.so $R/stylemesh_amd/libstylemesh_hip.so
