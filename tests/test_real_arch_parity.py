"""-m gpu: the repo's real UNet / VAE / ControlNet module code (reduced width) inside the product loop on the MI355X
against the oracle running the SAME weights in fp32 on the CPU -- tests/realarch.py explains what this puts under
parity that the one-conv fakes of test_hip_parity.py cannot.

Bars:
  * fp32 model on the GPU vs fp32 oracle: rel-L2 of the latent after every denoising step < 1e-3 (BASELINE.json's
    tolerance; the residual is MIOpen / hipBLASLt vs oneDNN summation order through a real conv/attention stack), and
    the host RNG stream ends in exactly the oracle's state;
  * 16-bit models (bf16 and fp16 -- the reference's own CUDA path runs the UNet under fp16 autocast, ED:1012; fused HIP
    kernels, flash attention, hipGraph replay, K-batched forwards): per-timestep drift vs the fp32 oracle at most 1.25x (realarch.GATE_FACTOR)
    the drift of the REFERENCE'S call pattern (batch-2 calls per resampling step, view batches) driving the same 16-bit
    model, and the two 16-bit runs within 2x that of each other (realarch.gate_16bit) -- 16-bit rounding through a
    random-init UNet under guidance 10 is a property of the dtype; what this repo adds on top of it is what is gated;
The measured numbers are written to gpurun_out/parity_real_arch.json for DESIGN.md.
"""
import copy
import json
import os

import pytest
import torch

from tests import realarch as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def reports():
    return {}


@pytest.mark.parametrize("case", list(R.REAL_CASES))
def test_real_architecture_in_the_loop(case, reports):
    rep = R.drift_report(case, dtypes=["bf16", "fp16"])
    reports[case] = rep
    print(json.dumps(rep))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_real_arch.json"), "w") as f:
        json.dump(reports, f, indent=1)
    assert rep["fp32_rng_tail_equal"] and rep["bf16_rng_tail_equal"] and rep["fp16_rng_tail_equal"]
    assert max(rep["fp32"]) < 1e-3, rep["fp32"]
    assert rep["graphs"]["eager"] == 0 and rep["graphs"]["captured"] >= 1, rep["graphs"]
    for name in ("bf16", "fp16"):  # fp16 = the dtype of the reference's own GPU path (CUDA autocast, ED:1012)
        ok, msg = R.gate_16bit(rep, name)
        assert ok, msg


def test_16bit_drift_over_many_steps_stays_at_the_reference_patterns():
    """6 denoising steps at reduced width (SD1.5 architecture, 512x1024; 12 steps of the SDXL architecture are in
    profiles/r3_precision.json: flat after the second step): the 16-bit loops must stay finite and within
    1.25x of the drift the reference's own call pattern shows in its own GPU arithmetic (fp32 weights under autocast) at EVERY step -- the two-step cases
    above cannot show a trend."""
    rep = R.drift_report(R.LONG_CASES["cfg2_sd_512x1024_6steps"], dtypes=["bf16", "fp16"], with_fp32=False)
    print(json.dumps({k: [float(f"{v:.3e}") for v in rep[k]] for k in rep if isinstance(rep[k], list)}))
    for name in ("bf16", "fp16"):
        ok, msg = R.gate_16bit(rep, name)
        assert ok, msg


def test_full_width_sdxl_forward_16bit_error_is_the_dtypes():
    """ONE forward of the full-width SDXL UNet (2.567 B parameters, batch 6 = the RePaint phase's batch): the fused
    16-bit path vs fp32 torch ops (MIOpen out of the loop) may err at most 1.5x as much as the plain torch 16-bit path
    does, for both 16-bit dtypes (every loop-level comparison uses reduced-width modules)."""
    rep = R.full_width_forward_report("sdxl", batches=(6,))
    print(json.dumps(rep))
    for name in ("bf16", "fp16"):
        r = rep[6][name]
        assert r["fused_finite"] and r["unfused_finite"], r
        assert r["fused"] <= 1.5 * r["unfused"] + 1e-4, (name, r)


def test_full_width_sd15_forward_16bit_error_is_the_dtypes():
    """The same for the full-width SD 1.5 UNet (BASELINE.json configs[1]): 8 heads of 40 / 80 / 160 -- since round 4 through
    ed_flash_attention's generic-head-dimension kernel instead of SDPA (AOTriton), plus the GEMM / convolution kernels at
    SD1.5's shapes."""
    from elasticdiffusion_official_amd import ops
    seen = set()
    orig = ops._call

    def spy(name, *a):
        seen.add(name)
        return orig(name, *a)

    ops._call = spy
    try:
        rep = R.full_width_forward_report("sd15", batches=(6,))
    finally:
        ops._call = orig
    print(json.dumps(rep))
    assert "ed_flash_attention" in seen and "ed_geglu_gemm" in seen
    for name in ("bf16", "fp16"):
        r = rep[6][name]
        assert r["fused_finite"] and r["unfused_finite"], r
        assert r["fused"] <= 1.5 * r["unfused"] + 1e-4, (name, r)


def test_fused_kernels_are_inside_the_bf16_loop():
    """The 16-bit run above must actually go through libelastic_hip.so's UNet kernels (not torch fallbacks)."""
    from elasticdiffusion_official_amd import models as M, ops
    seen = set()
    orig = ops._call

    def spy(name, *a):
        seen.add(name)
        return orig(name, *a)

    ops._call = spy
    try:
        unet, vae, cn = R.build_small("XL1.0")
        c = dict(R.REAL_CASES["cfg3_xl_1024x2048"], steps=1, R=1)
        R.run_product(c, unet, vae, cn, torch.bfloat16)
    finally:
        ops._call = orig
    from elasticdiffusion_official_amd import pipeline
    glue = {"ed_assemble_rows", "ed_phase_epilogue"} if pipeline.FUSED_GLUE else {"ed_pick_assemble", "ed_gather_views"}
    need = glue | M.fused_unet_entry_points()
    assert need <= seen, sorted(need - seen)


def test_text_kv_computed_once_per_image_follows_the_prompt():
    """pipeline.TEXT_KV_ONCE: the cross-attention k / v of the text rows are computed once per image, outside the hipGraph,
    into tensors the captured forward reads at fixed addresses.  A second image with ANOTHER prompt through the same pipeline
    (same graphs) must equal that prompt's image from a fresh pipeline to the 16-bit noise floor -- a stale k / v would give
    the first prompt's image -- and must match the run with the switch off (k / v re-projected in every forward)."""
    from elasticdiffusion_official_amd import ElasticDiffusion
    from tests.fakes import synthetic_text_embeds
    unet, vae, _ = R.build_small("XL1.0")
    c = dict(R.REAL_CASES["cfg3_xl_1024x2048"], steps=2, R=1)
    g = torch.Generator().manual_seed(5)
    (un, pun), (co, pco) = synthetic_text_embeds(1, cross_dim=64, pooled_dim=32, xl=True)
    prompts = {"": (un, pun), "a": (co, pco), "b": (co + torch.randn(co.shape, generator=g), pco + torch.randn(pco.shape, generator=g))}

    def make(once):
        p = ElasticDiffusion("cuda:0", c["sd"], view_batch_size=c["vbs"], unet=copy.deepcopy(unet).to(torch.float16),
                             vae=copy.deepcopy(vae), text_encoder=lambda s: prompts[s if isinstance(s, str) else s[0]])
        p.TEXT_KV_ONCE = once
        return p

    def run(p, prompt):
        p.seed_everything(c["seed"])
        return p.generate_latents(prompt, "", height=c["H"], width=c["W"], num_inference_steps=c["steps"],
                                  resampling_steps=c["R"], **R.LOOP_KW).float().cpu()

    pipe = make(True)
    za, zb = run(pipe, "a"), run(pipe, "b")           # same pipeline, same graphs, the prompt changes in between
    assert pipe._runner.stats()["eager"] == 0 and pipe._runner.stats()["captured"] >= 1
    zb_fresh, zb_off = run(make(True), "b"), run(make(False), "b")
    rel = lambda x, y: float((x - y).norm() / y.norm())   # noqa: E731
    assert rel(za, zb_fresh) > 0.05, "the two prompts must give different images for this test to mean anything"
    assert rel(zb, zb_fresh) < 0.03 and rel(zb, zb_off) < 0.03, (rel(zb, zb_fresh), rel(zb, zb_off), rel(za, zb_fresh))


# Round 6: a FULL 50-step schedule, fp16 product on the MI355X against the REFERENCE's own fp32 CPU latent.
# The fixture (tests/golden/g12_long_schedule.npz) is the real reference's generate_image driving this repo's reduced-width SDXL modules
# (tests/golden/make_long_schedule.py, builder container; the oracle's trace is bit-identical to it and supplies the checkpoints).
# Bars are ABSOLUTE: fp32 product < 1e-3 at every checkpoint (BASELINE.json's tolerance; measured ~1e-5), fp16 product at the end of the
# schedule <= LONG_FP16_BAR = 1.2 x what the first run measured (profiles/r6_s1_long_schedule_parity.json), i.e. a regression of the
# 16-bit path by 20 % fails here by name instead of surfacing as a drifting number in a bench line.
LONG_FP16_BAR = 1.64e-3     # 1.2 x 1.366e-3 (profiles/r6_s2_long_schedule_parity.json: 1.06e-3 at step 2, flat at 1.36-1.37e-3 from step 12 on)


def test_full_schedule_fp16_vs_reference_latent(golden_dir):
    import numpy as np
    from tests.golden.make_long_schedule import CASE, CHECKPOINTS, weight_fingerprint
    path = os.path.join(golden_dir, "g12_long_schedule.npz")
    if not os.path.isfile(path):
        pytest.skip("g12_long_schedule.npz not generated yet")
    g = np.load(path)
    unet, vae, _ = R.build_small(CASE["sd"])
    fp = weight_fingerprint(unet, vae)
    assert abs(fp - float(g["weight_fingerprint"])) <= 1e-9 * abs(fp), "seeded weights differ from the fixture's: regenerate it"
    assert tuple(int(k) for k in g["checkpoints"]) == tuple(CHECKPOINTS)
    ref = torch.from_numpy(g["reference_latent"])
    trace = [torch.from_numpy(z) for z in g["oracle_trace"]]
    assert torch.equal(trace[-1], ref)
    rep = {"case": CASE, "checkpoints": list(CHECKPOINTS)}
    for name, dt in (("fp32", torch.float32), ("fp16", torch.float16), ("fp16_stream32", torch.float16)):
        got, tail, pipe = R.run_product(CASE, unet, vae, None, dt, residual_fp32=name.endswith("stream32"))
        assert len(got) == CASE["steps"]
        rep[name] = [R.rel_l2(got[k - 1], z) for k, z in zip(CHECKPOINTS, trace)]
        rep[name + "_rng_tail_equal"] = bool(torch.equal(tail, torch.from_numpy(g["rng_tail"])))
        rep[name + "_finite"] = bool(torch.isfinite(got[-1]).all())
        rep[name + "_graphs"] = pipe._runner.stats()
    rep["fp16_bar"] = LONG_FP16_BAR
    print(json.dumps(rep))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "long_schedule_parity.json"), "w") as f:
        json.dump(rep, f, indent=1)
    assert rep["fp32_rng_tail_equal"] and rep["fp16_rng_tail_equal"] and rep["fp16_finite"] and rep["fp16_stream32_finite"]
    assert max(rep["fp32"]) < 1e-3, rep["fp32"]
    assert rep["fp16"][-1] <= LONG_FP16_BAR, rep["fp16"]
    # the tolerance mode (fp32 residual stream under fp16 branches) must be closer to the reference than plain fp16 at the end of the schedule
    assert rep["fp16_stream32"][-1] <= 0.95 * rep["fp16"][-1], (rep["fp16_stream32"], rep["fp16"])
