"""Generate the golden fixtures under tests/golden/ by running the REAL reference functions.

Run in the builder container only (needs /root/reference; it never travels to the GPU box):

    python tests/golden/make_golden.py

The reference modules are stub-imported (tests/golden/ref_loader.py), deterministic fake models from
tests/fakes.py and the DDIM restatement oracle/ddim.py are injected, and each glue function of SURVEY.md
section 8(a) plus the full ``generate_image`` loop is executed on CPU.  Only inputs (seeds / sizes) and OUTPUTS
are stored -- no reference source.  Fixture families follow SURVEY.md 8(c): G1 views+crops, G2 random nearest
downsample, G3 direction with resampling, G4 local uncond signal, G5 undo_step, G6 RRG, G7 tiled decode,
G8 end-to-end latents, G9 RNG event trace, G10 ControlNet.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ddim import DDIMOracle  # noqa: E402
from tests.fakes import FakeControlNet, FakeUNet, FakeVAE, synthetic_text_embeds  # noqa: E402
from tests.golden.ref_loader import make_reference_pipeline  # noqa: E402
from tests.golden import cases  # noqa: E402


def embed_fn(xl):
    (un, pun), (co, pco) = synthetic_text_embeds(1, xl=xl)
    state = {"n": 0}

    def fn(_prompts):
        state["n"] += 1
        return (un, pun) if state["n"] % 2 == 1 else (co, pco)

    return fn, (un, pun, co, pco)


def build(sd="1.5", sample_size=64, vbs=4, controlnet=False, patch=None):
    xl = sd.startswith("XL")
    fn, emb = embed_fn(xl)
    pipe, ref = make_reference_pipeline(FakeUNet(sample_size, xl=xl), FakeVAE(), DDIMOracle(), fn, sd_version=sd,
                                        view_batch_size=vbs, controlnet=FakeControlNet() if controlnet else None)
    if patch is not None:
        pipe.set_view_config(patch)
    pipe.random_downasmple_pre = {}
    return pipe, ref, emb


def save(name, **arrays):
    out = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def store_image(out, name, img):
    """Full image for small cases, ``cases.image_probe`` vectors for cfg4-sized ones (48 MiB fp32 otherwise)."""
    if img.numel() <= cases.FULL_IMAGE_MAX:
        out[f"{name}/image"] = img
    else:
        for k, v in cases.image_probe(img).items():
            out[f"{name}/image_probe/{k}"] = v


# ---------------------------------------------------------------------------------------------------
def g1_views():
    pipe, _, _ = build()
    rows = []
    for (H, W, sample, patch) in cases.G1_CASES:
        pipe.unet.config.sample_size = sample
        pipe.set_view_config(patch)
        vc = pipe.view_config
        Hl, Wl = H // 8, W // 8
        h_ws = Hl if vc["window_size"] + vc["context_size"] >= Hl else vc["window_size"]
        w_ws = Wl if vc["window_size"] + vc["context_size"] >= Wl else vc["window_size"]
        views = pipe.get_views(H, W, h_ws=h_ws, w_ws=w_ws, **vc)
        X = torch.arange(Hl * Wl, dtype=torch.float32).view(1, 1, Hl, Wl)
        crops = []
        for (a, b, c, d) in views:
            crop, n4 = pipe.crop_with_context(X, a, b, c, d, S=1, n=vc["context_size"] // 2)
            # a crop of an index image is fully described by its corner values + shape when contiguous;
            # store first/last element and shape, and a checksum of all gathered indices
            crops.append(dict(n4=[int(v) for v in n4], shape=list(crop.shape[-2:]), first=int(crop[0, 0, 0, 0]),
                              last=int(crop[0, 0, -1, -1]), checksum=int(crop.to(torch.float64).sum().item())))
        rows.append(dict(H=H, W=W, sample=sample, patch=patch, views=[list(map(int, v)) for v in views], crops=crops,
                         downsample_sd=list(map(int, pipe.get_downsample_size(H, W)))))
    pipe.sd_version = "XL1.0"
    for r in rows:
        r["downsample_xl"] = list(map(int, pipe.get_downsample_size(r["H"], r["W"])))
    with open(os.path.join(HERE, "g1_views.json"), "w") as f:
        json.dump(rows, f)
    print("g1_views.json", len(rows), "cases")


def g2_downsample():
    out = {}
    for name, (Hl, Wl, h, w, seed) in cases.G2_CASES.items():
        pipe, _, _ = build()
        pipe.seed_everything(seed)
        x = torch.randn(1, 4, Hl, Wl)
        prev, exclude = None, None
        for step in range(3):
            low, mask, prev = pipe.random_nearest_downsample(x, (h, w), prev_random_indices=prev, exclude_mask=exclude,
                                                             drop_p=0.7, nearest=(step == 0))
            if exclude is None:
                exclude = torch.zeros(len(prev), 4, dtype=torch.bool)
            exclude[torch.arange(len(prev)), prev] = True
            out[f"{name}/low{step}"] = low
            out[f"{name}/mask{step}"] = np.packbits(mask.numpy())
            out[f"{name}/mask_shape{step}"] = np.asarray(mask.shape)
            out[f"{name}/idx{step}"] = prev.to(torch.uint8)
        for k, v in pipe.random_downasmple_pre.items():
            out[f"{name}/table_{k}"] = v.to(torch.int64)
        out[f"{name}/rng_tail"] = torch.rand(4)
    save("g2_downsample", **out)


def g3_to_g6_functions():
    out = {}
    for name, c in cases.G3_CASES.items():
        pipe, ref, (un, pun, co, pco) = build(c["sd"], c["sample"], vbs=c["vbs"], patch=c.get("patch"))
        H, W = c["H"], c["W"]
        pipe.default_size = (4 * H, 4 * W)
        pipe.scheduler.set_timesteps(c["steps"])
        t = pipe.scheduler.timesteps[c["ti"]]
        pipe.seed_everything(c["seed"])
        x = torch.randn(1, 4, H // 8, W // 8)
        ds = pipe.get_downsample_size(H, W)
        direction, info = pipe.approximate_latent_direction_w_resampling(
            x, t, torch.cat([un, co]), resampling_steps=c["R"], downsample_size=ds,
            add_text_embeds=torch.cat([pun, pco]), drop_p=0.7)
        out[f"{name}/direction"] = direction
        out[f"{name}/init_low"] = info["init_downsampled_latent"]
        out[f"{name}/last_low"] = info["downsampled_latent"]
        out[f"{name}/uncond_score"] = info["scores"]["uncond_score"]
        out[f"{name}/low_direction"] = info["downsampled_direction"]
        local = pipe.compute_local_uncond_signal(x, t, un, pun, pipe.view_config)
        out[f"{name}/local"] = local
        ddim = pipe.scheduler.step(local + 10.0 * direction, t, x)
        out[f"{name}/x0"] = ddim["pred_original_sample"]
        out[f"{name}/prev"] = ddim["prev_sample"]
        undone = pipe.undo_step(ddim["prev_sample"], pipe.scheduler.timesteps[c["ti"] + 1])
        out[f"{name}/undone"] = undone
        grad, rinfo = pipe.reduced_resolution_guidance(
            x, t, direction, ddim["pred_original_sample"], un, pun, pipe.view_config, downsample_size=ds,
            rrg_scale=np.float64(c["rrg_w"]), guidance_scale=10.0, text_embeds=None,
            donwsampled_scores={"latent": info["downsampled_latent"], "uncond_score": info["scores"]["uncond_score"],
                                "direction": info["downsampled_direction"]}, bottom=False, right=False)
        out[f"{name}/rrg_grad"] = grad
        out[f"{name}/rrg_x0_low"] = rinfo["x0"][0]
        out[f"{name}/rng_tail"] = torch.rand(4)
    save("g3_functions", **out)


def g7_tiled_decode():
    out = {}
    for name, (Hl, Wl, sample, low_vram, seed) in cases.G7_CASES.items():
        pipe, _, _ = build(sample_size=sample)
        pipe.low_vram = low_vram  # geometry switch only (ED:283-285); latents stay fp32
        z = torch.randn(1, 4, Hl, Wl, generator=torch.Generator().manual_seed(seed))
        store_image(out, name, pipe.tiled_decode(z))
    save("g7_tiled_decode", **out)


class RngTrace:
    """Records (fn, shape) of every RNG entry point the reference touches (SURVEY 8(c) 'RNG contract')."""

    def __init__(self):
        self.events = []

    def __enter__(self):
        self._saved = dict(manual_seed=torch.manual_seed, rand=torch.rand, randn=torch.randn,
                           randn_like=torch.randn_like, randint=torch.randint, np_randint=np.random.randint)
        ev, sv = self.events, self._saved

        def wrap(name, fn, shape_of):
            def inner(*a, **k):
                ev.append([name, shape_of(*a, **k)])
                return fn(*a, **k)
            return inner

        def sz(*a, **k):
            s = a if a and isinstance(a[0], int) else (a[0] if a else k.get("size"))
            return [int(v) for v in s]

        torch.manual_seed = wrap("manual_seed", sv["manual_seed"], lambda s: int(s))
        torch.rand = wrap("rand", sv["rand"], sz)
        torch.randn = wrap("randn", sv["randn"], sz)
        torch.randn_like = wrap("randn_like", sv["randn_like"], lambda x, **k: list(x.shape))
        torch.randint = wrap("randint", sv["randint"], lambda lo, hi, size, **k: [int(lo), int(hi)] + [int(v) for v in size])
        np.random.randint = wrap("np_randint", sv["np_randint"], lambda *a, **k: [int(v) for v in a])
        return self

    def __exit__(self, *exc):
        sv = self._saved
        torch.manual_seed, torch.rand, torch.randn = sv["manual_seed"], sv["rand"], sv["randn"]
        torch.randn_like, torch.randint, np.random.randint = sv["randn_like"], sv["randint"], sv["np_randint"]


def g8_g10_end_to_end():
    out, traces = {}, {}
    for name, c in cases.E2E_CASES.items():
        cn = c.get("controlnet", False)
        pipe, ref, _ = build(c["sd"], c["sample"], vbs=c["vbs"], controlnet=cn, patch=c.get("patch"))
        pipe.seed_everything(c["seed"])
        cap = {}
        fn = "tiled_decode" if c.get("tiled") else "decode_latents"
        orig = getattr(pipe, fn)

        def grab(z, orig=orig, cap=cap):
            cap["z"] = z.clone()
            img = orig(z)
            cap["img"] = img.clone()
            return img

        setattr(pipe, fn, grab)
        kw = dict(cases.E2E_KW)
        kw.update(c.get("kw", {}))
        if cn:
            ds = pipe.get_downsample_size(c["H"], c["W"])
            cond = cases.synthetic_condition(ds[0] * 8, ds[1] * 8)
            # the reference's prepare_image goes through diffusers' VaeImageProcessor (absent); for a float tensor input
            # with do_normalize=False it is the identity, so inject that and keep the reference's CFG doubling.
            pipe.control_image_processor = type("P", (), {"preprocess": staticmethod(lambda image, height, width: image)})()
            kw.update(condition_image=cond, controlnet_conditioning_scale=0.2)
        with RngTrace() as tr:
            pipe.generate_image(prompts="p", negative_prompts="", height=c["H"], width=c["W"],
                                num_inference_steps=c["steps"], resampling_steps=c["R"], progress=lambda it: it,
                                rrg_scherduler_cls=ref.CosineScheduler, tiled_decoder=bool(c.get("tiled")), **kw)
        out[f"{name}/latent"] = cap["z"]
        if c.get("keep_image"):
            store_image(out, name, cap["img"])
        out[f"{name}/rng_tail"] = torch.rand(4)
        if c.get("trace"):
            traces[name] = tr.events
    save("g8_end_to_end", **out)
    with open(os.path.join(HERE, "g9_rng_trace.json"), "w") as f:
        json.dump(traces, f)
    print("g9_rng_trace.json", {k: len(v) for k, v in traces.items()})


def g11_generate():
    out = {}
    for name, c in cases.G11_CASES.items():
        pipe, ref, (un, pun, co, pco) = build(c["sd"], c["sample"])
        pipe.log_freq = 1
        pipe.default_size = (4 * 8 * c["h"], 4 * 8 * c["w"])
        pipe.scheduler.set_timesteps(c["steps"])
        pipe.seed_everything(c["seed"])
        z = torch.randn(1, 4, c["h"], c["w"])
        cap = {}
        orig = pipe.decode_latents

        def grab(lat, orig=orig, cap=cap):
            cap["z"] = lat.clone()
            return orig(lat)

        pipe.decode_latents = grab
        import tqdm as _tqdm  # the reference wraps the loop in tqdm (ED:767); silence it
        ref.tqdm = lambda it, *a, **k: it
        img, info = pipe.generate(z, torch.cat([un, co]), torch.cat([pun, pco]), guidance_scale=c["guidance"])
        out[f"{name}/final"] = cap["z"]
        out[f"{name}/inter_x0"] = torch.cat(info["inter_x0"])
        out[f"{name}/rng_tail"] = torch.rand(4)
    save("g11_generate", **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    g1_views()
    g2_downsample()
    g3_to_g6_functions()
    g7_tiled_decode()
    g8_g10_end_to_end()
    g11_generate()
