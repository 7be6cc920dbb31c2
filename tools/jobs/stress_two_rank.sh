# repeat the two-rank dense-vs-sparse lock-step test; print each run's per-step deviations (usage: stress_two_rank.sh [n=10] [ENV=VAL ...])
N=${1:-10}; shift
export TMPDIR=/tmp STYLEMESH_TEST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 "$@"
for i in $(seq 1 $N); do
  d=$(mktemp -d)
  timeout -s KILL 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + i)) tests/two_rank_worker.py dense_vs_sparse $d > $d/log.txt 2>&1
  rc=$?
  STYLEMESH_TEST_BACKEND=gloo python - <<PY
import torch, sys
sys.path.insert(0, "tests")
try:
    r = torch.load("$d/rank0.pt")
    L = r["lockstep"]
    bad = [(k, round(float(d[0] / (0.1 * d[1])), 6), round(float(d[4]), 4)) for k, d in enumerate(L) if float(d[0]) > 0.1 * 1e-5 * float(d[1]) or float(d[2]) != 0]
    print("run $i rc=$rc steps", L.shape[0], "stable", [round(float(d[4]), 3) for d in L], "BAD" if bad else "ok", bad)
except Exception as e:
    print("run $i rc=$rc ERR", e); print(open("$d/log.txt").read()[-1500:])
PY
  rm -rf $d
done
