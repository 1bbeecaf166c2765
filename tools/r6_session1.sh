#!/bin/bash
# Round-6 GPU session 1 (~40 GPU-minutes): ABI v8 (exact raw-stream split), the leaner GEGLU epilogue, row T.
#   1 full GPU suite on the new tree      2 GEMM kernels: round-5 library vs product, bit identity + time (tools/gemm_ab/run.py)
#   3 in-situ forward A/B                 4 bench.py as the driver runs it (short)        5 fp32 legs for cfg2 / cfg5 / cfg4 (row T)
#   6 the 8-rank one-GPU rehearsal in a loop with per-rank logs (the round-5 SIGABRT)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6s1; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log
cp gpurun_out/long_schedule_parity.json $O/ 2>/dev/null; cp gpurun_out/parity_real_arch.json $O/ 2>/dev/null
timeout 300 python tools/gemm_ab/run.py --prev tools/ab/libelastic_hip_r5.so --new elasticdiffusion_official_amd/libelastic_hip.so --rounds 7 > $O/gemm_ab_r5_vs_r6.jsonl 2> $O/gemm_ab.err; cat $O/gemm_ab_r5_vs_r6.jsonl | cut -c1-200
timeout 400 python tools/fwd_ab.py --libs tools/ab/libelastic_hip_r5.so,product --batches 20,6 --modes fp16 > $O/forward_ab_r5_vs_r6.json 2> $O/fwd_ab.err; tail -5 $O/forward_ab_r5_vs_r6.json | cut -c1-400
( time timeout 900 python bench.py --gpus 1 --steps 4 --warmup 2 ) > $O/bench_s1.json 2> $O/bench_s1.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6s1/bench_s1.json") if l.startswith("{")][-1])
r = d.get("roofline") or {}
print("bench", d["value"], d["ms_per_step"], d.get("phase_ms_last_image"), d["roofline_e2e"]["frac"], r.get("kernel"), r.get("frac"), r.get("us_per_launch"))
print(json.dumps(d["tolerance"].get("fp32_unet_same_workload"))[:600], d["tolerance"].get("meets_1e-3"))
PY
for wl in sd15_512x1024 sdxl_1024x2048_controlnet sdxl_2048x2048_tiled; do
  ( time timeout 1200 python bench.py --gpus 1 --workload $wl --steps 2 --warmup 1 --fp32-leg on --no-cpu-baseline --no-extras ) > $O/bench_fp32leg_$wl.json 2> $O/bench_fp32leg_$wl.err
  python - "$O/bench_fp32leg_$wl.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(d["config"]["workload"], d["value"], d["ms_per_step"], json.dumps(d["tolerance"].get("fp32_unet_same_workload"))[:500], d["tolerance"].get("meets_1e-3"))
except Exception as e:
    print("no line", sys.argv[1], e)
PY
  tail -2 $O/bench_fp32leg_$wl.err
done
mkdir -p $O/hunt
for i in $(seq 1 40); do
  L=/tmp/hunt_$i; rm -rf $L
  lvl=0; [ $((i % 2)) -eq 0 ] && lvl=1
  ( AMD_LOG_LEVEL=$lvl ED_DIST_BACKEND=gloo MIOPEN_FIND_MODE=FAST HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port $((29600+i)) --redirects 3 --log-dir $L bench.py --gpus 8 --steps 2 --warmup 1 --small --workload sd15_512x1024 --timesteps 3 --no-cpu-baseline ) > $O/hunt/run$i.out 2> $O/hunt/run$i.err
  rc=$?
  echo "hunt $i rc=$rc amd_log=$lvl"
  if [ $rc -ne 0 ]; then
    mkdir -p $O/hunt/fail$i; for r in 0 1 2 3 4 5 6 7; do f=$(find $L -path "*/$r/stderr.log" | head -1); [ -n "$f" ] && grep -v "MIOpen(HIP): Warning" $f | tail -c 20000 > $O/hunt/fail$i/rank$r.stderr; done
    grep -l -i "abort\|terminate\|what()\|fault\|HSA_STATUS" $O/hunt/fail$i/* | head
  else
    rm -f $O/hunt/run$i.err
  fi
done
du -sh $O
