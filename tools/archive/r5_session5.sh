#!/bin/bash
# Round-5 GPU session 5 (~9 GPU-minutes, evidence only -- no product code involved): `roofline.traffic` for the OTHER workloads' bench lines
# (VERDICT r4 item 8): FETCH_SIZE / WRITE_SIZE / MFMA-busy passes over each workload's own launch mix (tools/pmc_workload.py), summarised to
# profiles/r5_unet_pmc_<workload>.json (bench.py looks that name up), then the workloads' bench lines again so that they carry the number.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5s5; mkdir -p $O
for w in sd15_512x1024 sdxl_1024x2048_controlnet sdxl_2048x2048_tiled; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
    d=/tmp/pmc_${w}_$(echo $c | cut -d' ' -f1); mkdir -p $d
    (cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o w -- python $GRAFT_REPO_ROOT/tools/pmc_workload.py $w > $d/run.log 2>&1)
    tail -1 $d/run.log
  done
  python tools/pmc_summarise.py --workload $w $O/r5_unet_pmc_$w.json /tmp/pmc_${w}_FETCH_SIZE /tmp/pmc_${w}_WRITE_SIZE /tmp/pmc_${w}_SQ_VALU_MFMA_BUSY_CYCLES > $O/pmc_summarise_$w.log 2>&1
  tail -2 $O/pmc_summarise_$w.log
  cp $O/r5_unet_pmc_$w.json profiles/
done
( time timeout 200 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --workload sd15_512x1024 ) > $O/bench_cfg2.json 2> $O/bench_cfg2.err
( time timeout 300 python bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --workload sdxl_1024x2048_controlnet ) > $O/bench_cfg5.json 2> $O/bench_cfg5.err
( time timeout 300 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --workload sdxl_2048x2048_tiled ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err
python - <<'PY'
import json
for f in ("bench_cfg2", "bench_cfg5", "bench_cfg4"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r5s5/{f}.json") if l.startswith("{")][-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["traffic"], r["algorithmic_bytes_per_launch"], r.get("traffic_source"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -q -n 2 $O/bench_cfg2.err $O/bench_cfg5.err $O/bench_cfg4.err
