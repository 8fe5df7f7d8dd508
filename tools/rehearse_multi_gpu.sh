#!/bin/bash
# Rehearsal of the N-rank bench path on ONE GPU: two ranks share the card and exchange over gloo
# (RCCL refuses two ranks on one device).  (1) trajectory gather with --verify-gather (compares the
# gathered arrays with a single-GPU trace of the whole fan); (2) deposition-profile exchange, fast
# and bit-exact forms.  The RCCL path itself runs only on the driver's multi-GPU node.
set -e
export RAYS_BENCH_SHARE_GPU=1 RAYS_BENCH_BACKEND=gloo
run() { timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 "$@"; }
run --verify-gather
run --config $PWD/configs/cfg5b_axisym256k_rk4_damp.in --exchange deposition
run --config $PWD/configs/cfg5b_axisym256k_rk4_damp.in --exchange deposition --exact-profile
