#!/bin/bash
# Round-6 GPU session 28 (~12 GPU-minutes): 30 more repetitions of the 8-rank one-GPU rehearsal (the command that aborted once in round 5) on the last
# tree, every rank's stderr in its own file, AMD_LOG_LEVEL=1 on every second run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s28; mkdir -p $O/hunt
ok=0; bad=0
for i in $(seq 1 30); do
  L=/tmp/hunt_$i; rm -rf $L
  lvl=0; [ $((i % 2)) -eq 0 ] && lvl=1
  ( AMD_LOG_LEVEL=$lvl ED_DIST_BACKEND=gloo MIOPEN_FIND_MODE=FAST HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port $((29700+i)) --redirects 3 --log-dir $L bench.py --gpus 8 --steps 2 --warmup 1 --small --workload sd15_512x1024 --timesteps 3 --no-cpu-baseline ) > $O/hunt/run$i.out 2> $O/hunt/run$i.err
  rc=$?
  echo "run $i rc=$rc amd_log=$lvl" >> $O/rehearsal_runs.txt
  if [ $rc -ne 0 ]; then
    bad=$((bad+1)); mkdir -p $O/hunt/fail$i; for r in 0 1 2 3 4 5 6 7; do f=$(find $L -path "*/$r/stderr.log" | head -1); [ -n "$f" ] && grep -v "MIOpen(HIP): Warning" $f | tail -c 20000 > $O/hunt/fail$i/rank$r.stderr; done
  else
    ok=$((ok+1)); rm -f $O/hunt/run$i.err $O/hunt/run$i.out
  fi
done
echo "clean $ok aborted $bad" | tee -a $O/rehearsal_runs.txt
