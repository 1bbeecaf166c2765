#!/bin/bash
# Round-6 FINAL GPU session (~25 GPU-minutes), the driver's round-end commands on the final tree:
#   1 pytest -m gpu -x -q (complete)   2 __graft_entry__.smoke()   3 bench.py --gpus 1 --steps 20 --warmup 5 (the driver's round-5 command)
#   4 rocprofv3 --kernel-trace --stats of a bench run (+ trace summary)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6final2; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6final2/bench_final.json") if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("final", d["value"], d["ms_per_step"], d["config"]["images_in_flight"], d.get("latency_s_per_image"), d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("us_per_launch"), r.get("traffic"))
    print(json.dumps(d["tolerance"].get("fp32_unet_same_workload"))[:500], d["tolerance"].get("meets_1e-3"))
    print(d["extras"], d["graphs"], d.get("parity_16bit_rel_l2", {}).get("gate_vs_reference_gpu_arithmetic"))
    print(d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("sample", "")[:200])
except Exception as e:
    print("no final line", e)
PY
tail -3 $O/bench_final.err
P=/tmp/prof_bench; mkdir -p $P
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --fp32-leg off > $P/run.log 2>&1)
find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
grep "^{" $P/run.log | tail -1 > $O/bench_under_rocprofv3.json
python tools/analyze_trace.py $(find $P -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -8 $O/trace_summary.txt
head -14 $O/bench_kernel_stats.csv | cut -c1-150
du -sh $O
