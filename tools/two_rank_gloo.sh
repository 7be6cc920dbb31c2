#!/bin/bash
# Functional run of the N = 2 protocol on a 1-GPU box: both ranks share cuda:0 and exchange over gloo
# (STYLEMESH_DIST_BACKEND=gloo). Timing is meaningless here; the losses must agree between the exchange variants.
# Usage (GPU box): bash tools/two_rank_gloo.sh [extra bench.py flags]
export STYLEMESH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 25 --warmup 5 --cpu-steps 0 "$@"
