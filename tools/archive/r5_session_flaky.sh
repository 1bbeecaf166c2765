#!/bin/bash
# Round-5 diagnosis (~4 GPU-minutes): the 8-rank reduced-width rehearsal aborted once (rank 3, SIGABRT) in the last full-suite run after
# passing in four earlier runs; repeat the exact command with the whole stderr kept.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5flaky; mkdir -p $O
for i in $(seq 1 14); do
  ( ED_DIST_BACKEND=gloo MIOPEN_FIND_MODE=FAST HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 100 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port $((29600+i)) bench.py --gpus 8 --steps 2 --warmup 1 --small --workload sd15_512x1024 --timesteps 3 --no-cpu-baseline ) > $O/run$i.out 2> $O/run$i.err
  echo "run $i rc=$? lines=$(grep -c '^{' $O/run$i.out)"
  grep -n -i "abort\|terminate\|what():\|HIP error\|hipError\|Error\|error:" $O/run$i.err | grep -v "ChildFailedError\|error_file" | head -8
done
