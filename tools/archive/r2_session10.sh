#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
( time timeout 500 python tools/r2_probe.py table=1,2,3,4,5,6,7,8,10,12,13,20,26,40 ) > $O/table.log 2>&1; grep "^{" $O/table.log
( time timeout 400 python bench.py --in-flight 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_inflight2.json 2> $O/bench_inflight2.err; tail -2 $O/bench_inflight2.err; cut -c1-400 $O/bench_inflight2.json
