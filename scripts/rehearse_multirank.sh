#!/bin/bash
# GPU-box rehearsal of the N>1 bench path on ONE GPU: two ranks share GPU 0, the Result gather goes through
# the host group (PVHIP_NO_RCCL=1) because RCCL refuses two ranks on one device.  Checks both launcher contracts:
# bench.py starting its own ranks (python bench.py --gpus 2), and torch.distributed.run starting them
# (RANK/LOCAL_RANK/WORLD_SIZE, barrier, max-over-ranks, one JSON line from rank 0).
set -u
export PVHIP_NO_RCCL=1
echo "== python bench.py --gpus 2 (own launcher)"
timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --batch 64 --cpu-images 0 || exit $?
echo "== torch.distributed.run"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 3 --warmup 1 --batch 64 --cpu-images 0
