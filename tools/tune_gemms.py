"""GPU experiment: PyTorch TunableOp over the SDXL UNet's GEMMs (hipBLASLt picks one kernel per shape by heuristic; 32 % of
GPU time is its MT256x256x64 stream-K kernel).  Tunes at the 1-GPU batch shapes, writes the winners to
tunableop_cache/tunableop_results0.csv (in-tree build artefact like miopen_cache/), then A/Bs the forward with the tuned
table against the heuristic.  usage: tune_gemms.py <batches e.g. 20,6> <max_ms_per_candidate> <max_iters>"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ED_NO_TUNABLEOP"] = "1"  # the package must not switch TunableOp on by itself in this process
import torch
import torch.cuda.tunable as tunable

import elasticdiffusion_official_amd  # noqa: F401
from tools.r2_probe import build_unet, ev_time, inputs

batches = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "20,6").split(",")]
max_ms, max_it = int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 4
out_dir = os.path.join(ROOT, "tunableop_cache")
os.makedirs(out_dir, exist_ok=True)
path = os.path.join(out_dir, "tunableop_results.csv")
unet, cfg = build_unet()


def fwd(B):
    x, e, kw, t = inputs(cfg, B)
    with torch.no_grad():
        return lambda: unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)


base = {B: ev_time(fwd(B), reps=3, warm=2) / 1e3 for B in batches}
print(json.dumps({"probe": "tunableop", "stage": "heuristic", "ms": base}), flush=True)
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(path)
tunable.set_max_tuning_duration(max_ms)
tunable.set_max_tuning_iterations(max_it)
for B in batches:
    t0 = time.perf_counter()
    f = fwd(B)
    f()
    torch.cuda.synchronize()
    tunable.write_file()
    print(json.dumps({"probe": "tunableop", "stage": "tuned", "B": B, "seconds": round(time.perf_counter() - t0, 1),
                      "entries": len(tunable.get_results())}), flush=True)
tunable.tuning_enable(False)
tuned = {B: ev_time(fwd(B), reps=3, warm=2) / 1e3 for B in batches}
print(json.dumps({"probe": "tunableop", "stage": "with_table", "ms": tuned,
                  "gain": {B: round(base[B] / tuned[B], 4) for B in batches}}), flush=True)
print("files", os.listdir(out_dir))
