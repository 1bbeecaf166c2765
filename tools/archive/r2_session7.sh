#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s7; mkdir -p $O
for spec in "32 sdxl 1" "20 sd15 1" "20 sd15 0"; do
  set -- $spec
  d=$O/prof_$2_b$1_cl$3; mkdir -p $d
  (cd /tmp && ED_CHANNELS_LAST=$3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -o fwd -- python $GRAFT_REPO_ROOT/tools/fwd_once.py $1 $2 > $GRAFT_REPO_ROOT/$d/run.log 2>&1)
  find $d -name "*kernel_trace.csv" -delete
  python - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f"--- {sys.argv[1]}: total {tot/3e6:.1f} ms per forward")
    for r in rows[:8]:
        print(f"{float(r['TotalDurationNs'])/3e6:8.2f} ms {int(r['Calls'])//3:5d}x {float(r['AverageNs'])/1e3:9.1f}us  {r['Name'][:110]}")
else:
    print("no stats for", sys.argv[1]); print(open(sys.argv[1] + '/run.log').read()[-1500:])
PY
done
( time timeout 600 python -m pytest -q -m gpu --timeout 600 -p no:cacheprovider tests/test_unet_kernels.py tests/test_real_arch_parity.py tests/test_models_and_text.py ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log; grep -E "FAILED|Error" $O/pytest.log | head
( time timeout 200 python tools/r2_probe.py attn ) > $O/probe_attn.log 2>&1; grep "^{" $O/probe_attn.log
