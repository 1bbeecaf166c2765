#!/bin/bash
# Round-6 GPU session 12 (~25 GPU-minutes): checkpoint on the tree with the fusions and two images in flight:
#   1 the complete GPU suite   2 bench.py as the driver runs it (short)   3 rocprofv3 --kernel-trace --stats of a bench run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s12; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log
( time timeout 1200 python bench.py --gpus 1 --steps 6 --warmup 2 ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6s12/bench.json") if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("bench", d["value"], d["ms_per_step"], d["config"]["images_in_flight"], d.get("latency_s_per_image"), d.get("phase_ms_last_image"), d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("us_per_launch"))
    print(json.dumps(d["tolerance"].get("fp32_unet_same_workload"))[:700], d["tolerance"].get("meets_1e-3"))
    print(d["extras"])
    print({k: (v.get("tflops") or v.get("gbs"), v.get("ms_per_image")) for k, v in d.get("unet_kernels", {}).items()})
except Exception as e:
    print("no bench line", e)
PY
tail -3 $O/bench.err
P=/tmp/prof_bench; mkdir -p $P
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --fp32-leg off > $P/run.log 2>&1)
find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
grep "^{" $P/run.log | tail -1 > $O/bench_under_rocprofv3.json
head -30 $O/bench_kernel_stats.csv | cut -c1-150
du -sh $O
