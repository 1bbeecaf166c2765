"""Real-architecture parity harness (test infrastructure; also used by bench.py's parity leg).

Round-1 parity ran a one-conv fp32 stand-in model (tests/fakes.py) through the loop: that proves the latent-space glue,
not the model boundary.  Here the repo's OWN ``UNet2DConditionModel`` / ``AutoencoderKL`` / ``ControlNetModel`` classes
(elasticdiffusion_official_amd/models.py, reduced width so the CPU side stays in seconds -- ``SMALL_UNET_CONFIGS``)
run inside the product loop on the MI355X and, with the SAME weights in fp32 on the CPU, inside the oracle
(oracle/elastic_oracle.py = the reference's loop, elastic_diffusion.py:1012-1078, pad strips :327-364).  That puts
under parity: the fused UNet kernels (GroupNorm/SiLU, GEGLU, LayerNorm, flash attention, fused residual adds) in the
16-bit runs, the hipGraph replay, the K = R+1 batched forward, ``_embed_rows``, and the real ``DiagonalGaussian``
pad-strip path through a real VAE encoder.

Three comparisons per geometry, each per timestep (relative L2 of the latent after every denoising step):
  * ``fp32``      product with the fp32 model on the GPU            vs oracle with the fp32 model on the CPU  (GATE 1e-3)
  * ``bf16``      product with the bf16 model on the GPU            vs the same fp32 oracle                   (reported)
  * ``batching``  product bf16 (K resampling pairs + V views in ONE forward) vs the reference's call pattern
                  (batch-2 calls per resampling step, ED:661-681; view batches of view_batch_size, ED:830-850) driving
                  the SAME bf16 GPU model                                                                      (reported)
"""
import copy

import torch
import torch.nn as nn

from oracle.ddim import DDIMOracle
from oracle.elastic_oracle import ElasticOracle
from tests.fakes import synthetic_text_embeds
from tests.golden import cases

# name -> geometry + loop settings (BASELINE.json configs[1], [2], [4] geometries; 2 steps keep the oracle in seconds)
REAL_CASES = {
    "cfg2_sd_512x1024": dict(sd="1.5", H=512, W=1024, vbs=4, steps=3, R=2, seed=0),
    "cfg3_xl_1024x2048": dict(sd="XL1.0", H=1024, W=2048, vbs=16, steps=2, R=2, seed=1),
    "cfg5_cn_xl_1024x2048": dict(sd="XL1.0", H=1024, W=2048, vbs=16, steps=2, R=1, seed=2, controlnet=True),
}
LOOP_KW = dict(guidance_scale=10.0, new_p=0.3, rrg_stop_t=0.2, rrg_init_weight=1000, cosine_scale=10.0,
               repaint_sampling=True)


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def embed_fn(xl, cross_dim=64, pooled_dim=32):
    (un, pun), (co, pco) = synthetic_text_embeds(1, cross_dim=cross_dim, pooled_dim=pooled_dim, xl=xl)
    state = {"n": 0}

    def fn(_):
        state["n"] += 1
        return (un, pun) if state["n"] % 2 == 1 else (co, pco)

    return fn


def build_small(sd, controlnet=False, seed=0):
    """fp32 CPU master copies (unet, vae, controlnet|None) of the reduced-width real architectures."""
    from elasticdiffusion_official_amd.models import build_models
    mods = build_models(sd, device="cpu", dtype=torch.float32, controlnet=controlnet, small=True, seed=seed)
    return mods if controlnet else mods + (None,)


class OnDevice(nn.Module):
    """Lets the CPU oracle loop drive a model that lives on the GPU in another dtype: inputs are moved over, the
    output comes back as fp32 CPU tensors.  Only the call PATTERN is the oracle's; the arithmetic is the GPU model's."""

    def __init__(self, mod, device, dtype):
        super().__init__()
        self.mod = copy.deepcopy(mod).to(device, dtype).eval()
        self.config = mod.config
        self.dev, self.dt = device, dtype
        if hasattr(mod, "add_embedding"):
            self.add_embedding = mod.add_embedding

    def _mv(self, v):
        if torch.is_tensor(v):
            return v.to(self.dev, self.dt if v.is_floating_point() else None)
        if isinstance(v, dict):
            return {k: self._mv(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return type(v)(self._mv(x) for x in v)
        return v

    @torch.no_grad()
    def forward(self, x, t, **kw):
        out = self.mod(self._mv(x), torch.as_tensor(t).to(self.dev), **{k: self._mv(v) for k, v in kw.items()})
        if isinstance(out, dict):
            return {"sample": out["sample"].float().cpu()}
        down, mid = out
        return [d.float().cpu() for d in down], mid.float().cpu()


class AutocastOnDevice(OnDevice):
    """The arithmetic of the reference's OWN GPU path: fp32 weights (ED:121, `torch_dtype = float32` unless low_vram) inside
    ``torch.autocast('cuda')`` (ED:1012) -- Linear / Conv2d / attention take 16-bit copies of their inputs and weights and return
    16-bit results (so the residual sums are 16-bit too), GroupNorm / LayerNorm / softmax run in fp32 -- with plain torch ops (every
    fused-kernel switch off, as diffusers would run it).  This is the comparator of the 16-bit gate since round 5 (VERDICT r4
    item 1b); rounds 2-4 compared against this repo's pure-16-bit module, which is close to but not what the reference computes."""

    def __init__(self, mod, device, dtype):
        super().__init__(mod, device, torch.float32)
        self.cast = dtype

    @torch.no_grad()
    def forward(self, x, t, **kw):
        from elasticdiffusion_official_amd import models as M
        saved = M.FUSED_KERNELS
        M.FUSED_KERNELS = False
        try:
            with torch.autocast(torch.device(self.dev).type, dtype=self.cast):
                return super().forward(x, t, **kw)
        finally:
            M.FUSED_KERNELS = saved


def run_oracle(case, unet, vae, cn):
    """The oracle loop on the CPU -> (per-timestep latents, rng tail)."""
    c = REAL_CASES[case] if isinstance(case, str) else case
    xl = c["sd"].startswith("XL")
    orc = ElasticOracle(unet, vae, DDIMOracle(), embed_fn(xl), sd_version=c["sd"], view_batch_size=c["vbs"],
                        pooled_dim=32 if xl else None, controlnet=cn)
    kw = dict(LOOP_KW)
    if cn is not None:
        ds = orc.get_downsample_size(c["H"], c["W"])
        kw.update(condition_image=cases.synthetic_condition(ds[0] * 8, ds[1] * 8), controlnet_conditioning_scale=0.2)
    orc.seed_everything(c["seed"])
    trace = []
    orc.generate_latent("p", "", height=c["H"], width=c["W"], num_inference_steps=c["steps"], resampling_steps=c["R"],
                        trace=trace, **kw)
    return trace, torch.rand(4)


def run_product(case, unet, vae, cn, dtype, device="cuda:0", residual_fp32=False):
    """The HIP product path with copies of the given modules on the GPU (UNet/ControlNet in ``dtype``, VAE fp32).
    ``residual_fp32``: the round-6 tolerance mode (fp32 residual stream under the 16-bit branches)."""
    from elasticdiffusion_official_amd import ElasticDiffusion
    c = REAL_CASES[case] if isinstance(case, str) else case
    xl = c["sd"].startswith("XL")
    pipe = ElasticDiffusion(device, c["sd"], view_batch_size=c["vbs"], unet=copy.deepcopy(unet).to(dtype),
                            vae=copy.deepcopy(vae), controlnet=None if cn is None else copy.deepcopy(cn).to(dtype),
                            text_encoder=embed_fn(xl), residual_fp32=residual_fp32)
    kw = dict(LOOP_KW)
    if cn is not None:
        ds = pipe.get_downsample_size(c["H"], c["W"])
        kw.update(condition_image=cases.synthetic_condition(ds[0] * 8, ds[1] * 8), controlnet_conditioning_scale=0.2)
    pipe.seed_everything(c["seed"])
    trace = []
    pipe.generate_latents("p", "", height=c["H"], width=c["W"], num_inference_steps=c["steps"],
                          resampling_steps=c["R"], trace=trace, **kw)
    tail = torch.rand(4)
    return [z.cpu() for z in trace], tail, pipe


DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}

# Longer runs at reduced width: how the 16-bit drift behaves over many denoising steps, not just the first two
# (VERDICT r2 item 1b).  ~5 s of CPU oracle per step.
LONG_CASES = {
    "cfg3_xl_1024x2048_12steps": dict(sd="XL1.0", H=1024, W=2048, vbs=16, steps=12, R=2, seed=1),
    "cfg2_sd_512x1024_6steps": dict(sd="1.5", H=512, W=1024, vbs=4, steps=6, R=2, seed=0),
}


def drift_report(case, dtype=torch.bfloat16, device="cuda:0", with_fp32=True, with_batching=True, dtypes=None, comparator="autocast"):
    """-> dict of per-timestep rel-L2 lists (see module docstring) + RNG-tail equality flags.

    ``dtypes`` (names from DTYPES, default: just ``dtype`` reported under its name; bf16 is also reported under the
    legacy keys "batching" / "ref_pattern_vs_fp32"): for every 16-bit dtype d
        out[d]                            product (fused kernels, K-batched, hipGraph) in d      vs fp32 oracle
        out["ref_pattern_vs_fp32_" + d]   the reference's call pattern AND GPU arithmetic (fp32 weights under torch.autocast(d),
                                          plain torch ops: AutocastOnDevice; ``comparator="pure16"``: the round 2-4 comparator,
                                          this repo's module with d weights)                      vs fp32 oracle
        out["batching_" + d]              product in d vs that reference-pattern run
    fp16 is the dtype the reference's own GPU path runs the UNet in (CUDA autocast, ED:1012)."""
    c = REAL_CASES[case] if isinstance(case, str) else case
    if dtypes is None:
        dtypes = [k for k, v in DTYPES.items() if v == dtype]
    unet, vae, cn = build_small(c["sd"], controlnet=bool(c.get("controlnet")))
    want, tail = run_oracle(c, unet, vae, cn)
    out = {"case": case if isinstance(case, str) else "custom", "steps": c["steps"], "R": c["R"]}
    if with_fp32:
        got, t32, _ = run_product(c, unet, vae, cn, torch.float32, device)
        out["fp32"] = [rel_l2(a, b) for a, b in zip(got, want)]
        out["fp32_rng_tail_equal"] = bool(torch.equal(t32, tail))
    for name in dtypes:
        dt = DTYPES[name]
        got16, t16, pipe = run_product(c, unet, vae, cn, dt, device)
        out[name] = [rel_l2(a, b) for a, b in zip(got16, want)]
        out[name + "_rng_tail_equal"] = bool(torch.equal(t16, tail))
        out[name + "_finite"] = bool(all(torch.isfinite(z).all() for z in got16))
        out["graphs"] = pipe._runner.stats()
        if with_batching:
            wrap = AutocastOnDevice if comparator == "autocast" else OnDevice
            ref_pattern, _ = run_oracle(c, wrap(unet, device, dt), vae, None if cn is None else wrap(cn, device, dt))
            out["comparator"] = comparator
            out["batching_" + name] = [rel_l2(a, b) for a, b in zip(got16, ref_pattern)]
            out["ref_pattern_vs_fp32_" + name] = [rel_l2(a, b) for a, b in zip(ref_pattern, want)]
            if name == "bf16":
                out["batching"], out["ref_pattern_vs_fp32"] = out["batching_bf16"], out["ref_pattern_vs_fp32_bf16"]
    return out


# Round 5 (VERDICT r4 "What's weak"): the comparator is the reference's own GPU arithmetic (AutocastOnDevice) and the factor is
# 1.25 + 2e-4 instead of 1.5 + 1e-3 -- measured product / comparator ratios are 0.82-1.06 (profiles/r3_precision.json,
# r4_parity_real_arch.json), so a model uniformly 1.4x worse than today's no longer passes.
# bf16 (not the benchmarked dtype) measures 1.16-1.20 x the bf16-autocast comparator, which keeps fp32 norm outputs one rounding longer
# (profiles/r5_s2_parity_real_arch_autocast_comparator.json); fp16 -- the headline -- 0.90-0.92 x.
GATE_FACTOR, GATE_SLACK = 1.25, 2e-4
GATE_FACTOR_BY_DTYPE = {"fp16": 1.25, "bf16": 1.4}


def gate_16bit(rep, name):
    """The bar a 16-bit loop has to meet (tightened in round 5): its drift against the fp32 oracle may not
    exceed GATE_FACTOR x the drift of the REFERENCE'S OWN call pattern (batch-2 + view batches, ED:661-681, 830-850) in the
    reference's own GPU arithmetic (fp32 weights under torch.autocast, ED:1012) -- i.e. K-batching, the fused kernels, the 16-bit
    weights and the hipGraph add at most a quarter to what the dtype itself costs the reference -- and the two 16-bit runs may
    differ from each other by no more than 2x that (independent rounding noise adds in quadrature: sqrt(2) expected).
    -> (ok, message)"""
    ours, ref = max(rep[name]), max(rep["ref_pattern_vs_fp32_" + name])
    both = max(rep["batching_" + name])
    ok = ours <= GATE_FACTOR_BY_DTYPE.get(name, GATE_FACTOR) * ref + GATE_SLACK and both <= 2.0 * ref + GATE_SLACK and rep[name + "_finite"]
    return ok, f"{name}: product {ours:.3e}, reference pattern {ref:.3e}, product-vs-pattern {both:.3e}"


@torch.no_grad()
def full_width_forward_report(family="sdxl", batches=(6, 20), device="cuda:0", seed=0):
    """ONE forward of the FULL-WIDTH architecture (models.UNET_CONFIGS, 2.567 B parameters for SDXL) per batch size:
    the fused 16-bit path (HIP GroupNorm / LayerNorm / GEGLU / flash attention / fused adds, channels-last) and the plain
    torch 16-bit path (all switches off) against the SAME weights in fp32 torch ops with MIOpen out of the loop
    (``cudnn.enabled = False``: convolutions as im2col + rocBLAS GEMMs -- an independent arithmetic).  Every loop-level
    comparison runs reduced-width modules; this is the check that the full-size model's 16-bit error is the dtype's and
    not a kernel's.  -> {batch: {dtype: {"fused": rel-L2, "unfused": rel-L2}}}"""
    from elasticdiffusion_official_amd import models as M
    cfg = M.UNET_CONFIGS[family]
    with torch.device("meta"):
        unet = M.UNet2DConditionModel(**cfg)
    unet = unet.to_empty(device=device)
    M._seeded_init(unet, seed)
    unet = unet.eval().requires_grad_(False)
    S = cfg["sample_size"]
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    out = {}
    saved = {k: getattr(M, k) for k in ("FUSED_KERNELS", "FLASH_ATTENTION", "FUSED_QKV", "FUSED_LAYERNORM",
                                         "FUSED_ADD_LAYERNORM", "FUSED_CONV_BIAS", "FUSED_TEMB_ADD", "FUSED_TOKENS_ADD")}
    try:
        for B in batches:
            x = torch.randn(B, 4, S, S, generator=g).to(device)
            txt = torch.randn(B, 77, cfg["cross_attention_dim"], generator=g).to(device)
            kw = None
            if cfg["pooled_projection_dim"]:
                kw = {"text_embeds": torch.randn(B, cfg["pooled_projection_dim"], generator=g).to(device),
                      "time_ids": torch.tensor([[4096., 8192., 0., 0., 4096., 8192.]], device=device).expand(B, -1)}
            t = torch.tensor(500, device=device)
            for k in saved:
                setattr(M, k, False)
            prev = torch.backends.cudnn.enabled
            torch.backends.cudnn.enabled = False
            try:
                ref = unet(x, t, encoder_hidden_states=txt, added_cond_kwargs=kw)["sample"].float().cpu()
            finally:
                torch.backends.cudnn.enabled = prev
            out[B] = {"ref_abs_mean": float(ref.abs().mean())}
            for name, dt in DTYPES.items():
                u16 = copy.deepcopy(unet).to(dt)
                kw16 = None if kw is None else {k: v.to(dt) if v.is_floating_point() and k == "text_embeds" else v for k, v in kw.items()}
                res = {}
                for mode, on in (("unfused", False), ("fused", True)):
                    for k, v in saved.items():
                        setattr(M, k, v if on else False)
                    m = u16.to(memory_format=torch.channels_last) if (on and M.CHANNELS_LAST) else u16
                    y = m(x.to(dt), t, encoder_hidden_states=txt.to(dt), added_cond_kwargs=kw16)["sample"].float().cpu()
                    res[mode] = rel_l2(y, ref)
                    res[mode + "_finite"] = bool(torch.isfinite(y).all())
                out[B][name] = res
                del u16
    finally:
        for k, v in saved.items():
            setattr(M, k, v)
    return out
