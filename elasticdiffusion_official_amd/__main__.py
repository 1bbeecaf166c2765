"""Command line of the hot path, mirroring the reference's ``__main__`` blocks (elastic_diffusion.py:1134-1210,
elastic_diffusion_w_controlnet.py:1342-1433): same flags and defaults, PNGs + args.txt under
``<outdir>/<exp>/<timestamp>_<seed>/``.

    python -m elasticdiffusion_official_amd --prompt "..." --H 1024 --W 2048 --sd_version XL1.0 [--weights DIR]

Differences: needs a ROCm device (no CPU fallback); ``--weights DIR`` points at a local HF snapshot (unet/, vae/,
text_encoder*/ ...); without it the architecture is randomly initialised and the text embeddings are synthetic
(this build image has neither checkpoints nor network).  Boolean flags take true/false (the reference's
``type=bool`` treats every non-empty string as True).
"""
import argparse
import os
import time
from datetime import datetime

import torch


def _bool(v):
    return str(v).lower() in ("1", "true", "yes", "y", "t")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m elasticdiffusion_official_amd")
    ap.add_argument("--prompt", type=str, default="A realistic portrait of a young black woman. she has a Christmas red "
                    "hat and a red scarf. Her eyes are light brown like they're almost caramel color. Her attire, simple yet dignified.")
    ap.add_argument("--negative", type=str, default="blurry, ugly, duplicate, no details, deformed")
    ap.add_argument("--sd_version", type=str, default="XL1.0")
    ap.add_argument("--H", type=int, default=2048)
    ap.add_argument("--W", type=int, default=2048)
    ap.add_argument("--low_vram", type=_bool, default=False)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--num_sampled", type=int, default=1)
    ap.add_argument("--guidance_scale", type=float, default=10.0)
    ap.add_argument("--cosine_scale", type=float, default=10.0)
    ap.add_argument("--rrg_scale", type=float, default=4000)
    ap.add_argument("--resampling_steps", type=int, default=10)
    ap.add_argument("--new_p", type=float, default=0.3)
    ap.add_argument("--rrg_stop_t", type=float, default=0.2)
    ap.add_argument("--view_batch_size", type=int, default=16)
    ap.add_argument("--outdir", type=str, default="results_log/")
    ap.add_argument("--make_grid", type=_bool, default=False)
    ap.add_argument("--repaint_sampling", type=_bool, default=True)
    ap.add_argument("--tiled_decoder", type=_bool, default=False)
    ap.add_argument("--exp", type=str, default="ElasticDiffusion")
    ap.add_argument("--tag", type=str, default="")
    ap.add_argument("--log_freq", type=int, default=5)
    ap.add_argument("--verbose", type=_bool, default=False)
    ap.add_argument("--weights", type=str, default=None, help="local HF snapshot directory (optional)")
    ap.add_argument("--condition_image", type=str, default=None, help="pre-processed condition image => ControlNet path")
    ap.add_argument("--controlnet_conditioning_scale", type=float, default=0.2)
    ap.add_argument("--controlnet_model", type=str, default="depth")
    opt = ap.parse_args(argv)

    from . import ElasticDiffusion, ElasticDiffusionControlNet
    if not torch.cuda.is_available():
        raise SystemExit("no ROCm device: this package has no CPU path (the reference's CPU path lives in oracle/)")
    device = torch.device("cuda")
    kw = {}
    if opt.weights:
        from .text import load_clip
        kw["weights"] = opt.weights
        kw["text_encoder"] = load_clip(opt.weights, opt.sd_version.startswith("XL"), device)
    extra = {}
    if opt.condition_image:
        from PIL import Image
        sd = ElasticDiffusionControlNet(device, opt.sd_version, opt.controlnet_model, verbose=opt.verbose,
                                        log_freq=opt.log_freq, view_batch_size=opt.view_batch_size,
                                        low_vram=opt.low_vram, **kw)
        extra = dict(condition_image=Image.open(opt.condition_image),
                     controlnet_conditioning_scale=opt.controlnet_conditioning_scale)
    else:
        sd = ElasticDiffusion(device, opt.sd_version, verbose=opt.verbose, log_freq=opt.log_freq,
                              view_batch_size=opt.view_batch_size, low_vram=opt.low_vram, **kw)
    sd.seed_everything(opt.seed)
    t0 = time.time()
    imgs, image_log = sd.generate_image(prompts=[opt.prompt] * opt.num_sampled, negative_prompts=opt.negative,
                                        height=opt.H, width=opt.W, num_inference_steps=opt.steps, grid=opt.make_grid,
                                        guidance_scale=opt.guidance_scale, resampling_steps=opt.resampling_steps,
                                        new_p=opt.new_p, cosine_scale=opt.cosine_scale, rrg_init_weight=opt.rrg_scale,
                                        rrg_stop_t=opt.rrg_stop_t, repaint_sampling=opt.repaint_sampling,
                                        tiled_decoder=opt.tiled_decoder, **extra)
    torch.cuda.synchronize()
    print(f"Time taken: {time.time() - t0:.2f} seconds")
    if opt.verbose:  # the reference prints its TimeIt table here (ED:1191-1192); ours: GPU phases + host-side time
        for name, ms in sd.phase_times().items():
            print(f"  {name:<14s} {ms / 1e3:9.3f} s (GPU, HIP events)")
        for name, sec in sd.host_s.items():
            print(f"  host:{name:<20s} {sec:9.3f} s")
    save_dir = os.path.join(opt.outdir, opt.exp, f"{datetime.now().strftime('%Y-%m-%d %H:%M:%S')}_{opt.seed}")
    os.makedirs(save_dir, exist_ok=True)
    for i, img in enumerate(imgs):
        img.save(f"{save_dir}/{i}.png")
    for key, logged in image_log.items():  # ED:1201-1205: the verbose image log next to the images
        if isinstance(logged, dict):
            for label, img in logged.items():
                img.save(f"{save_dir}/{key}_{label}.png")
        else:
            logged.save(f"{save_dir}/{key}.png")
    with open(f"{save_dir}/args.txt", "w") as f:
        f.write("\n".join(f"{k}: {v}" for k, v in vars(opt).items()))
    print(f"saved {len(imgs)} image(s) to {save_dir}")
    return save_dir


if __name__ == "__main__":
    main()
