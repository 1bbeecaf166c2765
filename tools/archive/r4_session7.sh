#!/bin/bash
# Round-4 GPU session 7 (~5 GPU-minutes): ed_flash_attention at SD 1.x's head dimensions (40 / 80 / 160), cfg2 with it.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s7; mkdir -p $O
( time timeout 400 python -m pytest tests/test_unet_kernels.py tests/test_real_arch_parity.py -x -q -k "head_dims or sd15_forward or flash_attention" ) > $O/pytest_head_dims.log 2>&1
tail -6 $O/pytest_head_dims.log
( cd /tmp && time timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg2 -o cfg2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload sd15_512x1024 --steps 3 --warmup 1 --no-cpu-baseline --no-extras ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
find /tmp/prof_cfg2 -name "*kernel_stats.csv" -exec cp {} $O/cfg2_kernel_stats.csv \;
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4s7/bench_cfg2.json") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("cfg2", d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"))
PY
head -8 $O/cfg2_kernel_stats.csv | cut -c1-150
