"""GPU: run every BASELINE.json configuration once with the real architectures (random weights) at a few timesteps:
finite outputs, shapes, rough timings.  These are parity-test cases / scope rows, not bench lines."""
import os, sys, time
# smoke runs of shapes that are not bench lines: skip MIOpen's benchmarking search (minutes per new conv shape)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elasticdiffusion_official_amd import ElasticDiffusion, ElasticDiffusionControlNet
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from golden.cases import synthetic_condition

which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["cfg2", "cfg4", "cfg5"]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)


def run(name, pipe, **kw):
    pipe.seed_everything(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    imgs, _ = pipe.generate_image("a prompt", "bad", num_inference_steps=T, output_type="pt", **kw)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    pipe.seed_everything(1)
    imgs, _ = pipe.generate_image("a prompt", "bad", num_inference_steps=T, output_type="pt", **kw)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: image {tuple(imgs.shape)} finite={bool(torch.isfinite(imgs).all())} first {t1-t0:.1f}s second {t2-t1:.2f}s "
          f"phases {dict((k, round(v)) for k, v in pipe.phase_times().items())}", flush=True)


if "cfg2" in which:
    run("cfg2 SD1.5 512x1024 vbs4 R7", ElasticDiffusion(dev, "1.5", view_batch_size=4), height=512, width=1024,
        resampling_steps=7, rrg_init_weight=1000, cosine_scale=10.0)
if "cfg4" in which:
    run("cfg4 SDXL 2048x2048 tiled R7", ElasticDiffusion(dev, "XL1.0", view_batch_size=16), height=2048, width=2048,
        resampling_steps=7, rrg_init_weight=4000, cosine_scale=10.0, tiled_decoder=True)
if "cfg5" in which:
    p = ElasticDiffusionControlNet(dev, "XL1.0", "depth", view_batch_size=16)
    h, w = p.get_downsample_size(1024, 2048)
    run("cfg5 SDXL+ControlNet 1024x2048 R7", p, condition_image=synthetic_condition(h * 8, w * 8), height=1024, width=2048,
        resampling_steps=7, rrg_init_weight=2000, cosine_scale=10.0, controlnet_conditioning_scale=0.2)
