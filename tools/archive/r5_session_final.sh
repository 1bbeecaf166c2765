#!/bin/bash
# Round-5 FINAL GPU session (~22 GPU-minutes): what the driver runs at round end, on the final tree, plus the evidence for profiles/:
#   1. the complete `pytest -m gpu -x -q`
#   2. smoke()
#   3. bench.py (6 timed images, extras incl. two images in flight, live fp32 leg, CPU baseline + parity legs)
#   4. rocprofv3 --kernel-trace --stats of the bench command (3 images) -> kernel stats + trace summary
#   5. PMC passes: HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) + MFMA busy over the UNet forward's launch mix (tools/pmc_unet.py),
#      SQ counters of the MFMA kernels and the VAE split path per shape (tools/r5_pmc_kernels.py)
# No product change after this run.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5final; mkdir -p $O
# (0) the UNet PMC passes first: bench.py reads profiles/r5_unet_pmc.json for roofline.traffic
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  d=/tmp/pmc_$(echo $c | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o unet -- python $GRAFT_REPO_ROOT/tools/pmc_unet.py > $d/run.log 2>&1)
  tail -1 $d/run.log
done
python tools/pmc_summarise.py $O/r5_unet_pmc.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | tail -70
cp $O/r5_unet_pmc.json profiles/r5_unet_pmc.json
( time timeout 1150 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1
tail -5 $O/pytest_gpu_full.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1
tail -3 $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 ) > $O/bench_final.json 2> $O/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5final/bench_final.json") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("final", d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("us_per_launch"), r.get("traffic"))
print(d.get("parity_16bit_rel_l2", {}).get("gate_vs_reference_gpu_arithmetic"), d["extras"], d["graphs"])
print(json.dumps(d["tolerance"].get("fp32_unet_same_workload")))
PY
tail -2 $O/bench_final.err
P=/tmp/prof_bench; mkdir -p $P
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --fp32-leg off > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/bench_under_rocprof.err)
find $P -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
python tools/analyze_trace.py $(find $P -name "*kernel_trace.csv" | head -1) > $O/trace_summary.txt 2>&1; head -14 $O/trace_summary.txt
head -20 $O/bench_kernel_stats.csv | cut -c1-170
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmck_$(echo $pass | cut -d' ' -f1); mkdir -p $d
  (cd /tmp && timeout 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -o k -- python $GRAFT_REPO_ROOT/tools/r5_pmc_kernels.py > $d/run.log 2>&1)
  tail -1 $d/run.log
done
python tools/pmc_by_kernel.py $O/r5_mfma_kernels_pmc.json /tmp/pmck_SQ_WAVE_CYCLES /tmp/pmck_SQ_VALU_MFMA_BUSY_CYCLES /tmp/pmck_FETCH_SIZE /tmp/pmck_WRITE_SIZE --match "flash_attn|gemm_8phase|geglu_persist|gn32_nhwc" 2>&1 | tail -120
du -sh $O
