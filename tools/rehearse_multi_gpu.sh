#!/bin/bash
# Rehearsal of the N-rank bench path on ONE GPU: `python bench.py --gpus 2` starts its two ranks itself (no
# launcher); they share the card and exchange over gloo (RCCL refuses two ranks on one device).
# (1) trajectory gather with --verify-gather (compares the gathered arrays with a single-GPU trace of the
# whole fan); (2) deposition-profile exchange, fast and bit-exact forms; (3) the torchrun form the driver uses.
# The RCCL path itself runs only on the driver's multi-GPU node.
set -e
export RAYS_BENCH_SHARE_GPU=1 RAYS_BENCH_BACKEND=gloo
run() { timeout -k 10 500 python bench.py --gpus 2 --steps 3 --warmup 1 "$@"; }
O=gpurun_out/rehearse; mkdir -p $O
# (the bench line is the one line of stdout that starts with "{"; gloo prints its banner on the same stream)
run --verify-gather > $O/n2.out; grep '^{' $O/n2.out > $O/rehearse_n2_gloo.txt; head -c 300 $O/rehearse_n2_gloo.txt; echo
# four ranks on the one card (the box allows six processes on it)
timeout -k 10 700 python bench.py --gpus 4 --steps 2 --warmup 1 --verify-gather > $O/n4.out; grep '^{' $O/n4.out > $O/rehearse_n4_gloo.txt; head -c 300 $O/rehearse_n4_gloo.txt; echo
run --config $PWD/configs/cfg5b_axisym256k_rk4_damp.in --exchange deposition > $O/n2_dep.out; grep '^{' $O/n2_dep.out > $O/rehearse_n2_deposition_cfg5b_gloo.txt; head -c 300 $O/rehearse_n2_deposition_cfg5b_gloo.txt; echo
run --config $PWD/configs/cfg5b_axisym256k_rk4_damp.in --exchange deposition --exact-profile
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1
# a launcher that starts the wrong number of ranks must be refused, and so must --gpus N without N devices
if WORLD_SIZE=1 RANK=0 python bench.py --gpus 2 --steps 1 --warmup 0 2>/dev/null; then echo "ERROR: WORLD_SIZE=1 with --gpus 2 accepted"; exit 1; fi
if RAYS_BENCH_SHARE_GPU= python bench.py --gpus 2 --steps 1 --warmup 0 2>/dev/null; then echo "ERROR: --gpus 2 on one GPU accepted"; exit 1; fi
echo "rehearsal ok"
