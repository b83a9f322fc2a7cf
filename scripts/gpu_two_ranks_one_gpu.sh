#!/bin/bash
# Two data-parallel ranks on ONE GPU (the builder's boxes have one): exercises the multi-rank hipGraph trainer (graph segments with the
# gradient all-reduces between them) with real collectives.  RCCL refuses two ranks on one device, so gloo carries the CUDA tensors.
export PLEN_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for G in 1 0; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 -m plen_ml_walk_amd.train_vec \
     --envs 1024 --steps 100 --warmup 24 --batch 1024 --start-timesteps 2048 --replay 100000 --graphs $G 2>&1 | grep -E "metric|Error|error|Traceback" | head -5
done
python -m plen_ml_walk_amd.train_vec --envs 1024 --steps 100 --warmup 24 --batch 1024 --start-timesteps 2048 --replay 100000 --graphs 1 2>&1 | grep -E "metric|Error" | head -3
