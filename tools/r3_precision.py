"""Round-3 precision evidence, written to gpurun_out/r3_precision.json (copied to profiles/):

  loop        tests/realarch.drift_report for cfg2 / cfg3 / cfg5 geometries: fp32 / bf16 / fp16 product vs fp32 oracle,
              the reference's call pattern driving the same 16-bit model, product vs that pattern (per timestep)
  long        the same over 12-16 denoising steps (reduced width): the trend, not just the first two steps
  full_width  one forward of the full 2.567 B-parameter SDXL UNet at batch 6 and 20: fused 16-bit path and plain torch
              16-bit path vs fp32 torch ops without MIOpen

    python tools/r3_precision.py [loop] [long] [full]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch  # noqa: E402

from tests import realarch as R  # noqa: E402


def main():
    what = set(sys.argv[1:]) or {"loop", "long", "full"}
    path = os.path.join(ROOT, "gpurun_out", "r3_precision.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    out = json.load(open(path)) if os.path.isfile(path) else {}

    def save():
        with open(path, "w") as f:
            json.dump(out, f, indent=1)

    if "loop" in what:
        for case in R.REAL_CASES:
            t0 = time.time()
            rep = R.drift_report(case, dtypes=["bf16", "fp16"])
            rep["gate"] = {d: R.gate_16bit(rep, d) for d in ("bf16", "fp16")}
            rep["seconds"] = round(time.time() - t0, 1)
            out.setdefault("loop", {})[case] = rep
            print(json.dumps({case: {k: rep[k] for k in ("fp32", "bf16", "fp16", "ref_pattern_vs_fp32_bf16",
                                                          "ref_pattern_vs_fp32_fp16", "batching_bf16", "batching_fp16", "gate")}}), flush=True)
            save()
    if "long" in what:
        for name, c in R.LONG_CASES.items():
            t0 = time.time()
            rep = R.drift_report(c, dtypes=["bf16", "fp16"], with_fp32=True)
            rep["case"] = name
            rep["gate"] = {d: R.gate_16bit(rep, d) for d in ("bf16", "fp16")}
            rep["seconds"] = round(time.time() - t0, 1)
            out.setdefault("long", {})[name] = rep
            print(json.dumps({name: {k: [float(f"{v:.3e}") for v in rep[k]] for k in
                                     ("fp32", "bf16", "fp16", "ref_pattern_vs_fp32_bf16", "ref_pattern_vs_fp32_fp16")}}), flush=True)
            save()
    if "full" in what:
        t0 = time.time()
        rep = R.full_width_forward_report("sdxl", batches=(6, 20))
        out["full_width"] = {"family": "sdxl", "seconds": round(time.time() - t0, 1), "batches": {str(k): v for k, v in rep.items()}}
        print(json.dumps(out["full_width"]), flush=True)
        save()


if __name__ == "__main__":
    main()
