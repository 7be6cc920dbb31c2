# two concurrent single-rank lock-step stress processes per round; counts bad steps. usage: stress_pair.sh <rounds> [ENV=VAL ...]
N=$1; shift
export "$@" TMPDIR=/tmp
tot=0
for i in $(seq 1 $N); do
  (python tools/jobs/stress_lockstep.py 36 3 2>&1 | tail -1 > /tmp/sa_$i.txt) & (python tools/jobs/stress_lockstep.py 36 3 2>&1 | tail -1 > /tmp/sb_$i.txt) & wait
  for f in /tmp/sa_$i.txt /tmp/sb_$i.txt; do n=$(sed -n 's/.*bad \([0-9]*\) .*/\1/p' $f); tot=$((tot + ${n:-99})); done
done
echo "bad steps: $tot of $((N * 72))   [$*]"
