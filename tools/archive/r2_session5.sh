#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
( time timeout 300 python -m pytest -q -m gpu --timeout 300 -p no:cacheprovider tests/test_unet_kernels.py ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log; grep -E "FAILED|Error" $O/pytest.log | head
for mode in 0 1; do
  mkdir -p $O/prof$mode
  (cd /tmp && ED_CHANNELS_LAST=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$mode -o fwd -- python $GRAFT_REPO_ROOT/tools/fwd_once.py > $GRAFT_REPO_ROOT/$O/prof$mode/run.log 2>&1)
  find $O/prof$mode -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/fwd_cl${mode}_kernel_stats.csv
  find $O/prof$mode -name "*kernel_trace.csv" -delete
done
python - <<'PY'
import csv
for m in (0, 1):
    rows = list(csv.DictReader(open(f'gpurun_out/s5/fwd_cl{m}_kernel_stats.csv')))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f"--- channels_last={m}: total {tot/3e6:.1f} ms per forward")
    for r in rows[:22]:
        print(f"{float(r['TotalDurationNs'])/3e6:8.2f} ms {int(r['Calls'])//3:5d}x {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:100]}")
PY
( time ED_CHANNELS_LAST=1 timeout 200 python tools/r2_probe.py table=20,6 ) > $O/probe_cl.log 2>&1; grep "^{" $O/probe_cl.log
