#!/bin/bash
# Round-4 GPU session 8 (~5 GPU-minutes): 8-wave workgroups for the pipelined attention (v_path 9 / 10): tests, kernel-level
# probe, in-situ A/B; cfg2 with two images in flight.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4s8; mkdir -p $O
( time timeout 300 python -m pytest tests/test_unet_kernels.py -x -q -k "flash" ) > $O/pytest_flash.log 2>&1
tail -4 $O/pytest_flash.log
( time timeout 200 python tools/probe_attention.py ) > $O/probe_attention.jsonl 2> $O/probe_attention.err
cat $O/probe_attention.jsonl
for v in 5 10 9 5 10; do
  ED_DTYPE=fp16 ED_FLASH_VARIANT=$v timeout 150 python tools/probe_unet.py sdxl 20,6 2>/dev/null | tail -2 | sed "s/^/v_path $v: /"
done > $O/attention_waves_in_unet.txt
cat $O/attention_waves_in_unet.txt
( time timeout 300 python bench.py --workload sd15_512x1024 --in-flight 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-kernel-timing ) > $O/bench_cfg2_inflight2.json 2> $O/bench_cfg2_inflight2.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4s8/bench_cfg2_inflight2.json") if l.startswith("{")][-1])
print("cfg2 in-flight 2", d["value"], d["ms_per_step"], d["latency_s_per_image"], d["graphs"])
PY
