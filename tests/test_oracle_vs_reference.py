"""Side-by-side: the oracle and the REAL reference in one process, fresh seeds, exact equality demanded.
Skipped wherever /root/reference is absent (always on the GPU box)."""
import numpy as np
import pytest
import torch

from tests.golden.ref_loader import reference_available

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not reference_available(), reason="/root/reference not present")]

CASES = [
    # sd, H, W, steps, R, vbs, seed, patch, tiled, controlnet
    ("1.5", 512, 512, 3, 0, 1, 11, None, False, False),
    ("1.5", 512, 1024, 3, 3, 4, 12, None, False, False),
    ("1.5", 1024, 512, 2, 2, 3, 13, None, False, False),
    ("1.5", 1080, 1920, 2, 2, 4, 14, None, False, False),
    ("1.5", 768, 768, 2, 2, 4, 15, 48, False, False),
    ("1.5", 640, 896, 2, 1, 4, 16, None, True, False),
    ("XL1.0", 1024, 2048, 2, 2, 16, 17, None, False, False),
    ("XL1.0", 1024, 1536, 2, 1, 2, 18, 96, False, False),
    ("1.5", 512, 1024, 2, 2, 4, 19, None, False, True),
    ("XL1.0", 1024, 2048, 2, 1, 16, 20, None, False, True),
    ("XL1.0", 2048, 2048, 2, 2, 16, 21, None, True, False),   # cfg4: 16 views, 64-tile decode
]


@pytest.mark.parametrize("sd,H,W,steps,R,vbs,seed,patch,tiled,cn", CASES)
def test_generate_image_bit_identical(sd, H, W, steps, R, vbs, seed, patch, tiled, cn):
    from tests.golden.make_golden import build
    from tests.golden import cases
    from tests.test_oracle_golden import make_oracle

    xl = sd.startswith("XL")
    pipe, ref, _ = build(sd, 128 if xl else 64, vbs=vbs, controlnet=cn, patch=patch)
    pipe.seed_everything(seed)
    cap = {}
    fn = "tiled_decode" if tiled else "decode_latents"
    orig = getattr(pipe, fn)

    def grab(z):
        cap["z"] = z.clone()
        cap["img"] = orig(z)
        return cap["img"]

    setattr(pipe, fn, grab)
    kw = dict(cases.E2E_KW)
    okw = dict(kw)
    if cn:
        ds = pipe.get_downsample_size(H, W)
        cond = cases.synthetic_condition(ds[0] * 8, ds[1] * 8)
        pipe.control_image_processor = type("P", (), {"preprocess": staticmethod(lambda image, height, width: image)})()
        kw.update(condition_image=cond, controlnet_conditioning_scale=0.2)
        okw.update(condition_image=cond, controlnet_conditioning_scale=0.2)
    pipe.generate_image(prompts="p", negative_prompts="", height=H, width=W, num_inference_steps=steps,
                        resampling_steps=R, progress=lambda it: it, rrg_scherduler_cls=ref.CosineScheduler,
                        tiled_decoder=tiled, **kw)
    ref_tail = torch.rand(3)

    orc, _ = make_oracle(sd, 128 if xl else 64, vbs, controlnet=cn, patch=patch)
    orc.seed_everything(seed)
    img, info = orc.generate_image("p", "", height=H, width=W, num_inference_steps=steps, resampling_steps=R,
                                   tiled_decoder=tiled, **okw)
    assert torch.equal(info["latent"], cap["z"])
    assert torch.equal(img, cap["img"])
    assert torch.equal(torch.rand(3), ref_tail)


@pytest.mark.parametrize("case", ["cfg2_sd_512x1024", "cfg3_xl_1024x2048"])
def test_real_architecture_oracle_leg_is_the_reference(case):
    """tests/realarch.py's CPU leg (oracle + the repo's real reduced-width UNet/VAE modules) against the REAL reference
    driving the same module objects: bit-identical latents after every step, identical RNG end state.  This pins the
    oracle side of the -m gpu real-architecture parity tests to the reference itself."""
    from oracle.ddim import DDIMOracle
    from tests import realarch as R
    from tests.golden.ref_loader import make_reference_pipeline

    c = R.REAL_CASES[case]
    xl = c["sd"].startswith("XL")
    unet, vae, _ = R.build_small(c["sd"])
    want, tail = R.run_oracle(case, unet, vae, None)

    pipe, ref = make_reference_pipeline(unet, vae, DDIMOracle(), R.embed_fn(xl), sd_version=c["sd"],
                                        view_batch_size=c["vbs"], pooled_dim=32)
    pipe.random_downasmple_pre = {}
    cap = {}

    def grab(z):  # skip the (slow, irrelevant here) CPU VAE decode: keep the final latent, hand back a dummy image
        cap["z"] = z.clone()
        return torch.zeros(z.shape[0], 3, 8, 8)

    pipe.decode_latents = grab
    pipe.seed_everything(c["seed"])
    pipe.generate_image(prompts="p", negative_prompts="", height=c["H"], width=c["W"], num_inference_steps=c["steps"],
                        resampling_steps=c["R"], progress=lambda it: it, rrg_scherduler_cls=ref.CosineScheduler,
                        **R.LOOP_KW)
    assert torch.equal(cap["z"], want[-1])
    assert torch.equal(torch.rand(4), tail)


@pytest.mark.parametrize("cls_name,B", [("LinearScheduler", 1), ("ConstScheduler", 2)])
def test_other_rrg_schedulers_and_prompt_batches_bit_identical(cls_name, B):
    """The reference's other RRG weight schedules (ED:73-94, selected through ``rrg_scherduler_cls``, ED:972-979) and a
    batch of prompts (one latent per prompt, ED:981-1000): oracle == reference bit for bit."""
    from oracle import elastic_oracle as eo
    from oracle.ddim import DDIMOracle
    from tests.fakes import FakeUNet, FakeVAE, synthetic_text_embeds
    from tests.golden import cases
    from tests.golden.ref_loader import make_reference_pipeline

    def embeds():
        (un, pun), (co, pco) = synthetic_text_embeds(B)
        st = {"n": 0}

        def fn(_):
            st["n"] += 1
            return (un, pun) if st["n"] % 2 == 1 else (co, pco)
        return fn

    kw = dict(cases.E2E_KW, rrg_init_weight=600)
    prompts = ["p%d" % i for i in range(B)]
    pipe, ref = make_reference_pipeline(FakeUNet(64), FakeVAE(), DDIMOracle(), embeds(), sd_version="1.5", view_batch_size=3)
    pipe.random_downasmple_pre = {}
    cap = {"z": []}

    def grab(z):  # the reference decodes one sample per call (decode_bs = 1, ED:1090, 1121)
        cap["z"].append(z.clone())
        return torch.zeros(z.shape[0], 3, 8, 8)

    pipe.decode_latents = grab
    pipe.seed_everything(31)
    pipe.generate_image(prompts=prompts, negative_prompts="", height=512, width=768, num_inference_steps=4,
                        resampling_steps=2, progress=lambda it: it, rrg_scherduler_cls=getattr(ref, cls_name), **kw)
    tail = torch.rand(3)
    orc = eo.ElasticOracle(FakeUNet(64), FakeVAE(), DDIMOracle(), embeds(), sd_version="1.5", view_batch_size=3)
    orc.seed_everything(31)
    z = orc.generate_latent(prompts, "", height=512, width=768, num_inference_steps=4, resampling_steps=2,
                            rrg_scherduler_cls=getattr(eo, cls_name), **kw)
    assert len(cap["z"]) == B and torch.equal(z, torch.cat(cap["z"])) and torch.equal(torch.rand(3), tail)


def test_reference_controlnet_variant_rejects_prompt_batches():
    """Pinned fact about the reference (not a property this repo copies): its ControlNet ``generate_image``
    (EDC:1119-1322) cannot run a batch of prompts -- the CFG-doubled condition image has 2 rows (EDC:1029-1031) and is
    sliced ``[:x.shape[0]]`` against 2B model rows (EDC:482).  The product path builds condition rows for any B
    (pipeline._condition_rows); for B = 1 it is compared with the reference in the ControlNet goldens."""
    from oracle.ddim import DDIMOracle
    from tests.fakes import FakeControlNet, FakeUNet, FakeVAE, synthetic_text_embeds
    from tests.golden import cases
    from tests.golden.ref_loader import make_reference_pipeline
    (un, pun), (co, pco) = synthetic_text_embeds(2)
    st = {"n": 0}

    def fn(_):
        st["n"] += 1
        return (un, pun) if st["n"] % 2 == 1 else (co, pco)

    pipe, ref = make_reference_pipeline(FakeUNet(64), FakeVAE(), DDIMOracle(), fn, sd_version="1.5", view_batch_size=4,
                                        controlnet=FakeControlNet())
    pipe.random_downasmple_pre = {}
    ds = pipe.get_downsample_size(512, 1024)
    pipe.control_image_processor = type("P", (), {"preprocess": staticmethod(lambda image, height, width: image)})()
    pipe.seed_everything(41)
    with pytest.raises(RuntimeError, match="must match the size"):
        pipe.generate_image(prompts=["a", "b"], negative_prompts="", condition_image=cases.synthetic_condition(ds[0] * 8, ds[1] * 8),
                            controlnet_conditioning_scale=0.2, height=512, width=1024, num_inference_steps=2,
                            resampling_steps=1, progress=lambda it: it, rrg_scherduler_cls=ref.CosineScheduler,
                            **cases.E2E_KW)


def test_verbose_init_low_is_the_first_phase_latent():
    """ED:1023-1024: with verbose=True the reference hands ``generate`` the reduced latent of the FIRST direction
    estimate (nearest downsample of the initial noise), not the one of the RePaint phase that follows (ED:1043).  The
    oracle's ``logs['init_downsampled_latent']`` must be that tensor, bit for bit (the product path once took
    it after the second phase)."""
    from tests.golden import cases
    from tests.golden.make_golden import build
    from tests.test_oracle_golden import make_oracle

    pipe, ref, _ = build("1.5", 64, vbs=4)
    pipe.verbose, pipe.log_freq = True, 2
    cap = {}

    class _Stop(Exception):
        pass

    def grab(latent, *a, **k):  # the reference's first verbose action after the loop (ED:1095)
        cap["init"] = latent.clone()
        raise _Stop

    pipe.generate = grab
    pipe.seed_everything(23)
    with pytest.raises(_Stop):
        pipe.generate_image(prompts="p", negative_prompts="", height=512, width=1024, num_inference_steps=3,
                            resampling_steps=2, progress=lambda it: it, rrg_scherduler_cls=ref.CosineScheduler,
                            **cases.E2E_KW)
    orc, _ = make_oracle("1.5", 64, 4)
    orc.seed_everything(23)
    logs = {}
    orc.generate_latent("p", "", height=512, width=1024, num_inference_steps=3, resampling_steps=2, logs=logs,
                        **cases.E2E_KW)
    assert torch.equal(logs["init_downsampled_latent"], cap["init"])
