#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
for spec in "32 sdxl" "20 sd15"; do
  set -- $spec
  d=$O/prof_$2_b$1; mkdir -p $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o fwd -- python $GRAFT_REPO_ROOT/tools/fwd_once.py $1 $2 > $GRAFT_REPO_ROOT/$d/run.log 2>&1)
  find $d -name "*kernel_trace.csv" -delete
  python - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f"--- {sys.argv[1]}: total {tot/3e6:.1f} ms per forward")
    for r in rows[:6]:
        print(f"{float(r['TotalDurationNs'])/3e6:8.2f} ms {int(r['Calls'])//3:5d}x {float(r['AverageNs'])/1e3:9.1f}us  {r['Name'][:110]}")
PY
done
( time timeout 200 python tools/r2_probe.py fused unet table=20,6 ) > $O/probe.log 2>&1; grep "^{" $O/probe.log
( time timeout 400 python bench.py --workload sdxl_2048x2048_tiled --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err; tail -2 $O/bench_cfg4.err; cut -c1-300 $O/bench_cfg4.json
( time timeout 300 python bench.py --workload sd15_512x1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -2 $O/bench_cfg2.err; cut -c1-300 $O/bench_cfg2.json
