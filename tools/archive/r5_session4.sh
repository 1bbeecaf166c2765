#!/bin/bash
# Round-5 GPU session 4 (~9 GPU-minutes): the VAE decoder's upsampler convolutions on the split-operand path (saturating split of the raw
# stream with the nearest-2x upsampling folded in), and the fp16-vs-fp32 latent distance on three seeds.
#   1. tests/test_vae_split.py (incl. the new split / upsampler tests) + tests around the VAE
#   2. tools/r5_vae_ab.py decode + tiles
#   3. cfg4 bench (1 image): decode phase
#   4. headline bench, 3 timed images, fp32 leg on all three seeds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5s4; mkdir -p $O
( time timeout 600 python -m pytest tests/test_vae_split.py tests/test_models_and_text.py tests/test_abi.py -m gpu -x -q ) > $O/pytest_vae.log 2>&1
tail -8 $O/pytest_vae.log
( time timeout 400 python tools/r5_vae_ab.py --cases decode,tiles,encode ) > $O/vae_ab.jsonl 2> $O/vae_ab.err
cat $O/vae_ab.jsonl; tail -q -n 3 $O/vae_ab.err
( time timeout 500 python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --workload sdxl_2048x2048_tiled ) > $O/bench_cfg4.json 2> $O/bench_cfg4.err
( time timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --fp32-leg on --fp32-leg-seeds 3 ) > $O/bench_3seeds.json 2> $O/bench_3seeds.err
python - <<'PY'
import json
for f in ("bench_cfg4", "bench_3seeds"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r5s4/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["phase_ms_last_image"], d["extras"])
        print(json.dumps(d["tolerance"].get("fp32_unet_same_workload")))
    except Exception as e:
        print(f, "failed", e)
PY
tail -q -n 3 $O/bench_cfg4.err $O/bench_3seeds.err
