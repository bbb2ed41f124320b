#!/bin/bash
# the N = 2 bench line at the full shard size with both ranks on the one GPU of the box (gloo for the 40-byte records)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --all-on-device 0 2>gpurun_out/r02_two_ranks.err | grep '^{' > gpurun_out/r02_two_ranks_one_gpu.json
cut -c1-600 gpurun_out/r02_two_ranks_one_gpu.json; tail -2 gpurun_out/r02_two_ranks.err
