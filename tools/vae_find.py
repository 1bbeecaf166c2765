"""MIOpen normal-find pass over the fp32 VAE shapes the workloads hit, recorded in the in-tree user db (miopen_cache/):
  dec_full   decoder, batch 1, latent 128 x 256     (SDXL 1024x2048 decode, ED:267-272)
  dec_tiles  decoder, batch 8, latent 128 x 128     (cfg4 tiled decode, ED:275-310, pipeline.tiled_decode tile_batch=8)
  enc_strips encoder, batch 5, pixels 256 x 1024    (SDXL 1024x2048 pad strips, pipeline.STRIP_CHUNK = 5)
  enc_sd15   encoder, batch 5, pixels 128 x 512     (SD1.5 512x1024 pad strips)
Prints the time per call before (immediate mode, current db) and after the find.   usage: vae_find.py [--nchw] [names...]
--nchw sets models.VAE_NCHW_RESIDUAL (the VAE stays NCHW after the mid-block attention): run once without and
once with the flag for the A/B -- each layout gets its own find records."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import elasticdiffusion_official_amd  # noqa: F401
from elasticdiffusion_official_amd import models as M

DEV = "cuda:0"
CASES = {"dec_full": ("XL1.0", "dec", (1, 4, 128, 256)), "dec_tiles": ("XL1.0", "dec", (8, 4, 128, 128)),
         "enc_strips": ("XL1.0", "enc", (5, 3, 256, 1024)), "enc_sd15": ("1.5", "enc", (5, 3, 128, 512))}


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    args = sys.argv[1:]
    if "--nchw" in args:
        args.remove("--nchw")
        M.VAE_NCHW_RESIDUAL = True
    names = args or list(CASES)
    vaes = {}
    with torch.no_grad():
        for name in names:
            sd, kind, shape = CASES[name]
            if sd not in vaes:
                vaes[sd] = M.build_models(sd, device=DEV)[1]
            vae = vaes[sd]
            x = torch.randn(*shape, device=DEV)
            fn = (lambda: vae.decode(x).sample) if kind == "dec" else (lambda: vae.encode(x).latent_dist.mean)
            torch.backends.cudnn.benchmark = False
            before = timed(fn)
            torch.backends.cudnn.benchmark = True
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            find_s = time.perf_counter() - t0
            after_bench = timed(fn)
            torch.backends.cudnn.benchmark = False
            after = timed(fn)   # immediate mode again, now with the find-db records
            print(json.dumps({"case": name, "vae_nchw_residual": M.VAE_NCHW_RESIDUAL, "shape": list(shape), "before_ms": round(before, 2), "find_s": round(find_s, 1),
                              "after_benchmark_mode_ms": round(after_bench, 2), "after_immediate_ms": round(after, 2)}), flush=True)


if __name__ == "__main__":
    main()
