#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -6
python bench.py --steps 1 --warmup 1 > gpurun_out/bench_r1c.json 2> gpurun_out/bench_r1c.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_r1c.json'))
print({k:d[k] for k in ('value','images_per_min','ms_per_step','phase_ms_last_image','host_ms_last_image','roofline','cpu_baseline')})
print({k:(v['us_per_launch'],v['gbs']) for k,v in d['glue_kernels'].items()})
PY
tail -2 gpurun_out/bench_r1c.err
mkdir -p gpurun_out/prof_bench50
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench50 -o bench50 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench50/run.log 2>&1)
python tools/analyze_trace.py gpurun_out/prof_bench50/bench50_kernel_trace.csv > gpurun_out/prof_bench50/trace_summary.txt 2>&1; cat gpurun_out/prof_bench50/trace_summary.txt
rm -f gpurun_out/prof_bench50/bench50_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/pmc_$c
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o glue -- python $GRAFT_REPO_ROOT/tools/pmc_glue.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c/run.log 2>&1)
  ls gpurun_out/pmc_$c
done
TORCH_ROCM_FA_PREFER_CK=1 python tools/probe_unet.py sdxl 20 2>&1 | tail -1
tar czf gpurun_out/miopen_cache.tgz miopen_cache
