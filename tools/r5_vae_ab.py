"""The fp32 VAE with the ResnetBlock convolutions on split fp16 operands (models.VAE_SPLIT_CONV) against the library path (MIOpen fp32),
at the shapes the workloads use: the headline's pad-strip encode [5, 3, 256, 1024] (100 of them per image in 20 calls), its 1024 x 2048
decode, and cfg4's batch of 8 decode tiles of 128 x 128 latents.  Full-width SDXL VAE, seeded synthetic weights; HIP-event timing after a
warm-up; outputs compared (rel-L2) and per-kernel times of the split path reported (ops.TIMER).

    python tools/r5_vae_ab.py [--cases encode,decode,tiles] [--reps 5]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import elasticdiffusion_official_amd  # noqa: F401,E402
from elasticdiffusion_official_amd import models as M, ops  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    return ms[len(ms) // 2], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="encode,decode,tiles")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--switch", default="VAE_SPLIT_CONV", help="the models switch the two arms differ in (round 6: VAE_SPLIT_DOWNSAMPLE -- the encoder's "
                                                              "stride-2 downsamplers on the split path, with VAE_SPLIT_CONV on in both arms)")
    a = ap.parse_args()
    dev = "cuda:0"
    with torch.device("meta"):
        vae = M.AutoencoderKL(scaling_factor=0.13025, force_upcast=True)
    vae = vae.to_empty(device=dev)
    M._seeded_init(vae, 1)
    vae = vae.eval().requires_grad_(False)
    g = torch.Generator(device="cpu").manual_seed(0)
    cases = {
        "encode": ("pad-strip encode [5,3,256,1024]", lambda x: vae.encode(x).latent_dist.mean,
                   (torch.rand(5, 3, 1, 1, generator=g) * 2 - 1).expand(5, 3, 256, 1024).contiguous().to(dev)),
        "decode": ("decode [1,4,128,256] -> 1024 x 2048", lambda z: vae.decode(z).sample, torch.randn(1, 4, 128, 256, generator=g).to(dev)),
        "tiles": ("8 decode tiles [8,4,128,128] (cfg4)", lambda z: vae.decode(z).sample, torch.randn(8, 4, 128, 128, generator=g).to(dev)),
    }
    for key in a.cases.split(","):
        name, fn, inp = cases[key]
        res = {"case": name}
        outs = {}
        with torch.no_grad():
            for flag in (False, True):
                setattr(M, a.switch, flag)
                ms, out = timed(lambda: fn(inp), a.reps)
                res["split_ms" if flag else "library_ms"] = round(ms, 2)
                outs[flag] = out
            ops.TIMER.start()
            fn(inp)
            kt = ops.TIMER.stop()
        res["speedup"] = round(res["library_ms"] / res["split_ms"], 3)
        res["rel_l2_split_vs_library"] = rel(outs[True], outs[False])
        res["finite"] = bool(torch.isfinite(outs[True]).all())
        res["split_path_kernels_ms"] = {k: {"launches": v[0], "total_ms": round(v[2], 2),
                                            "tflops": round(ops.TIMER.work[k][0] / v[2] / 1e9, 1) if ops.TIMER.work.get(k, [0])[0] else None,
                                            "gbs": round(ops.TIMER.work[k][1] / v[2] / 1e6, 1) if k in ops.TIMER.work else None}
                                        for k, v in kt.items()}
        print(json.dumps(res), flush=True)
        del outs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
