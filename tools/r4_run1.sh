#!/bin/bash
# Round-4 GPU pass 1: new tests, the launcher's N = 2 paths over gloo on one GPU, the new workloads.
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_round4_gpu.py -x -q -m gpu > $O/t_round4.log 2>&1; echo "round4 rc=$?" >> $O/summary.txt
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py -x -q -m gpu -s > $O/t_fullsize.log 2>&1; echo "fullsize rc=$?" >> $O/summary.txt
cp gpurun_out/fullsize_parity.json $O/ 2>/dev/null
timeout 600 python bench.py --steps 20 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench c3 rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 > $O/bench_c3_n2_gloo.json 2> $O/bench_c3_n2_gloo.err; echo "bench n2 gloo rc=$?" >> $O/summary.txt
STYLEMESH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --replicas > $O/bench_c3_n2_replicas.json 2> $O/bench_c3_n2_replicas.err; echo "bench n2 replicas rc=$?" >> $O/summary.txt
timeout 300 python bench.py --gpus 2 --steps 5 > $O/bench_n2_nccl_refused.out 2>&1; echo "bench n2 nccl on 1 gpu rc=$? (expected 3)" >> $O/summary.txt
timeout 600 python bench.py --workload dip --steps 200 --warmup 20 --cpu-steps 2 --f32-steps 0 --late-epoch-views 0 > $O/bench_dip.json 2> $O/bench_dip.err; echo "bench dip rc=$?" >> $O/summary.txt
timeout 600 python bench.py --workload with_angle --steps 100 --warmup 20 --cpu-steps 2 --late-epoch-views 0 > $O/bench_with_angle.json 2> $O/bench_with_angle.err; echo "bench with_angle rc=$?" >> $O/summary.txt
timeout 600 python bench.py --workload c2 --steps 100 --warmup 20 --cpu-steps 0 --late-epoch-views 0 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?" >> $O/summary.txt
cat $O/summary.txt
