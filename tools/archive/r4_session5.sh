#!/bin/bash
# Round-4 GPU session 5 (~11 GPU-minutes): cross-attention k / v once per image (pipeline.TEXT_KV_ONCE), the other workloads'
# lines with the round-4 kernels.  rocprofv3 writes to /tmp and only its summary CSVs come back (session 4 lost its files:
# the results db pushed gpurun_out/ past the 64 MiB that are copied back).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s5; mkdir -p $O
( time timeout 600 python -m pytest tests/test_real_arch_parity.py tests/test_hip_parity.py tests/test_models_and_text.py -x -q ) > $O/pytest_realarch_parity.log 2>&1
tail -5 $O/pytest_realarch_parity.log
( time timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_headline_2img.json 2> $O/bench_headline_2img.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4s5/bench_headline_2img.json") if l.startswith("{")][-1])
print("headline", d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"])
PY
( cd /tmp && time timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg2 -o cfg2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload sd15_512x1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
find /tmp/prof_cfg2 -name "*kernel_stats.csv" -exec cp {} $O/cfg2_kernel_stats.csv \;
( time timeout 400 python bench.py --workload sdxl_2048x2048_tiled --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err
( time timeout 300 python bench.py --workload sdxl_1024x2048_controlnet --steps 2 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python - <<'PY'
import json
for n in ("cfg2", "cfg4", "cfg5"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r4s5/bench_{n}.json") if l.startswith("{")][-1])
        r = d.get("roofline") or {}
        print(n, d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"))
    except Exception as e:
        print(n, "failed", e)
PY
du -sh $O
