"""The oracle (oracle/elastic_oracle.py) against the golden vectors produced by the REAL reference functions
(tests/golden/make_golden.py).  Runs anywhere (no /root/reference, no GPU needed).

Integer / index / mask outputs must match exactly.  fp32 outputs are compared at 2e-5 absolute: the fixtures were
written with a different CPU thread count than the test may run with and oneDNN's conv reduction order is not pinned
(the side-by-side test test_oracle_vs_reference.py, same process, demands exact equality)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.ddim import DDIMOracle
from oracle import elastic_oracle as eo
from tests.fakes import FakeControlNet, FakeUNet, FakeVAE, synthetic_text_embeds
from tests.golden import cases

ATOL = 2e-5


def make_oracle(sd="1.5", sample=64, vbs=4, controlnet=False, patch=None):
    xl = sd.startswith("XL")
    (un, pun), (co, pco) = synthetic_text_embeds(1, xl=xl)
    state = {"n": 0}

    def embeds(_):
        state["n"] += 1
        return (un, pun) if state["n"] % 2 == 1 else (co, pco)

    orc = eo.ElasticOracle(FakeUNet(sample, xl=xl), FakeVAE(), DDIMOracle(), embeds, sd_version=sd, view_batch_size=vbs,
                           pooled_dim=16 if xl else None, controlnet=FakeControlNet() if controlnet else None)
    if patch is not None:
        orc.set_view_config(patch)
    return orc, (un, pun, co, pco)


def close(a, b, atol=ATOL):
    a = a.numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=0, atol=atol)


def test_g1_views_and_crops(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "g1_views.json")))
    assert len(rows) == len(cases.G1_CASES)
    for r in rows:
        H, W, sample, patch = r["H"], r["W"], r["sample"], r["patch"]
        ws = patch if patch is not None else sample // 2
        ctx = sample - ws
        Hl, Wl = H // 8, W // 8
        h_ws = Hl if ws + ctx >= Hl else ws
        w_ws = Wl if ws + ctx >= Wl else ws
        views = eo.get_views(H, W, h_ws, w_ws, ws)
        assert [list(v) for v in views] == r["views"]
        X = torch.arange(Hl * Wl, dtype=torch.float32).view(1, 1, Hl, Wl)
        for v, g in zip(views, r["crops"]):
            crop, n4 = eo.crop_with_context(X, *v, 1, ctx // 2)
            assert list(n4) == g["n4"] and list(crop.shape[-2:]) == g["shape"]
            assert int(crop[0, 0, 0, 0]) == g["first"] and int(crop[0, 0, -1, -1]) == g["last"]
            assert int(crop.to(torch.float64).sum().item()) == g["checksum"]
        assert list(eo.get_downsample_size(H, W, "1.5")) == r["downsample_sd"]
        assert list(eo.get_downsample_size(H, W, "XL1.0")) == r["downsample_xl"]


def test_get_views_rejects_non_multiple_of_8():
    with pytest.raises(ValueError):
        eo.get_views(515, 512, 32, 32, 32)


@pytest.mark.parametrize("name", list(cases.G2_CASES))
def test_g2_random_nearest_downsample(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g2_downsample.npz"))
    Hl, Wl, h, w, seed = cases.G2_CASES[name]
    orc, _ = make_oracle()
    orc.seed_everything(seed)
    x = torch.randn(1, 4, Hl, Wl)
    prev, exclude = None, None
    for step in range(3):
        low, mask, prev = orc.random_nearest_downsample(x, (h, w), prev_random_indices=prev, exclude_mask=exclude,
                                                        drop_p=0.7, nearest=(step == 0))
        if exclude is None:
            exclude = torch.zeros(len(prev), 4, dtype=torch.bool)
        exclude[torch.arange(len(prev)), prev] = True
        np.testing.assert_array_equal(low.numpy(), g[f"{name}/low{step}"])  # pure gather: exact
        np.testing.assert_array_equal(prev.numpy().astype(np.uint8), g[f"{name}/idx{step}"])
        shape = tuple(g[f"{name}/mask_shape{step}"])
        assert tuple(mask.shape) == shape
        want = np.unpackbits(g[f"{name}/mask{step}"])[: shape[0] * shape[1]].reshape(shape).astype(bool)
        np.testing.assert_array_equal(mask.numpy(), want)
    np.testing.assert_array_equal(torch.rand(4).numpy(), g[f"{name}/rng_tail"])  # same generator state
    # the four cached index tables of the reference (ED:584-604)
    rsel, rmarks = eo.downsample_axis_tables(Hl, h)
    csel, cmarks = eo.downsample_axis_tables(Wl, w)
    n_r, n_c = len(g[f"{name}/table_row_indices"]), len(g[f"{name}/table_col_indices"])
    np.testing.assert_array_equal(rsel[:n_r], g[f"{name}/table_row_indices"])
    np.testing.assert_array_equal(csel[:n_c], g[f"{name}/table_col_indices"])
    np.testing.assert_array_equal(rmarks, g[f"{name}/table_mask_row_indices"])
    np.testing.assert_array_equal(cmarks, g[f"{name}/table_mask_col_indices"])


@pytest.mark.parametrize("name", list(cases.G3_CASES))
def test_g3_to_g6_function_chain(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g3_functions.npz"))
    c = cases.G3_CASES[name]
    orc, (un, pun, co, pco) = make_oracle(c["sd"], c["sample"], c["vbs"], patch=c.get("patch"))
    H, W = c["H"], c["W"]
    orc.default_size = (4 * H, 4 * W)
    orc.scheduler.set_timesteps(c["steps"])
    t = orc.scheduler.timesteps[c["ti"]]
    orc.seed_everything(c["seed"])
    x = torch.randn(1, 4, H // 8, W // 8)
    ds = orc.get_downsample_size(H, W)
    direction, info = orc.approximate_latent_direction_w_resampling(
        x, t, torch.cat([un, co]), torch.cat([pun, pco]), ds, resampling_steps=c["R"], drop_p=0.7)
    assert direction.dtype == torch.float32
    close(direction, g[f"{name}/direction"])
    np.testing.assert_array_equal(info["init_downsampled_latent"].numpy(), g[f"{name}/init_low"])
    np.testing.assert_array_equal(info["downsampled_latent"].numpy(), g[f"{name}/last_low"])
    close(info["scores"]["uncond_score"], g[f"{name}/uncond_score"])
    close(info["downsampled_direction"], g[f"{name}/low_direction"])
    local = orc.compute_local_uncond_signal(x, t, un, pun, orc.view_config)
    close(local, g[f"{name}/local"])
    ddim = orc.scheduler.step(local + 10.0 * direction, t, x)
    close(ddim["pred_original_sample"], g[f"{name}/x0"], 2e-4)
    close(ddim["prev_sample"], g[f"{name}/prev"], 2e-4)
    undone = orc.undo_step(ddim["prev_sample"], orc.scheduler.timesteps[c["ti"] + 1])
    close(undone, g[f"{name}/undone"], 2e-4)
    grad, rinfo = orc.reduced_resolution_guidance(
        t, ddim["pred_original_sample"], guidance_scale=10.0, rrg_scale=np.float64(c["rrg_w"]),
        donwsampled_scores={"latent": info["downsampled_latent"], "uncond_score": info["scores"]["uncond_score"],
                            "direction": info["downsampled_direction"]})
    close(grad, g[f"{name}/rrg_grad"], 2e-4)
    close(rinfo["x0"][0], g[f"{name}/rrg_x0_low"], 2e-4)
    np.testing.assert_array_equal(torch.rand(4).numpy(), g[f"{name}/rng_tail"])


@pytest.mark.parametrize("name", list(cases.G7_CASES))
def test_g7_tiled_decode(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g7_tiled_decode.npz"))
    Hl, Wl, sample, low_vram, seed = cases.G7_CASES[name]
    orc, _ = make_oracle(sample=sample)
    orc.low_vram = low_vram
    z = torch.randn(1, 4, Hl, Wl, generator=torch.Generator().manual_seed(seed))
    cases.assert_image_matches(g, name, orc.tiled_decode(z), ATOL)


@pytest.mark.parametrize("name", list(cases.E2E_CASES))
def test_g8_g10_end_to_end(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g8_end_to_end.npz"))
    c = cases.E2E_CASES[name]
    orc, _ = make_oracle(c["sd"], c["sample"], c["vbs"], controlnet=c.get("controlnet", False), patch=c.get("patch"))
    orc.seed_everything(c["seed"])
    kw = dict(cases.E2E_KW)
    kw.update(c.get("kw", {}))
    if c.get("controlnet"):
        ds = orc.get_downsample_size(c["H"], c["W"])
        kw.update(condition_image=cases.synthetic_condition(ds[0] * 8, ds[1] * 8), controlnet_conditioning_scale=0.2)
    img, info = orc.generate_image("p", "", height=c["H"], width=c["W"], num_inference_steps=c["steps"],
                                   resampling_steps=c["R"], tiled_decoder=bool(c.get("tiled")), **kw)
    z = info["latent"]
    want = g[f"{name}/latent"]
    rel = np.linalg.norm(z.numpy() - want) / np.linalg.norm(want)
    assert rel < 1e-5, rel
    if c.get("keep_image"):
        cases.assert_image_matches(g, name, img, 1e-4)
    np.testing.assert_array_equal(torch.rand(4).numpy(), g[f"{name}/rng_tail"])


@pytest.mark.parametrize("name", [k for k, c in cases.E2E_CASES.items() if c.get("trace")])
def test_g9_rng_event_trace(golden_dir, name):
    """Same sequence of (rng entry point, shape/seed) events as the reference run (SURVEY 8(c) 'RNG contract')."""
    from tests.golden.make_golden import RngTrace

    want = json.load(open(os.path.join(golden_dir, "g9_rng_trace.json")))[name]
    c = cases.E2E_CASES[name]
    orc, _ = make_oracle(c["sd"], c["sample"], c["vbs"])
    orc.seed_everything(c["seed"])
    with RngTrace() as tr:
        orc.generate_latent("p", "", height=c["H"], width=c["W"], num_inference_steps=c["steps"],
                            resampling_steps=c["R"], **cases.E2E_KW)
    assert tr.events == want


def test_g12_long_schedule_fixture_belongs_to_the_seeded_weights_and_the_oracle(golden_dir):
    """tests/golden/g12_long_schedule.npz (the REFERENCE's latent after a full 50-step schedule driving the reduced-width SDXL modules,
    tests/golden/make_long_schedule.py): the seeded weights built here are the fixture's (fingerprint), the stored oracle trace ends in the
    reference's latent, and the first two timesteps of the oracle -- all a few-minute CPU suite can afford of the 10-minute run --
    reproduce the first checkpoint.  (The -m gpu test test_full_schedule_fp16_vs_reference_latent holds the product to it.)"""
    import os

    import numpy as np
    import torch
    from oracle.ddim import DDIMOracle
    from oracle.elastic_oracle import ElasticOracle
    from tests import realarch as R
    from tests.golden.make_long_schedule import CASE, CHECKPOINTS, weight_fingerprint

    g = np.load(os.path.join(golden_dir, "g12_long_schedule.npz"))
    assert np.array_equal(g["oracle_trace"][-1], g["reference_latent"]) and tuple(g["checkpoints"]) == tuple(CHECKPOINTS)
    assert g["reference_latent"].shape == (1, 4, CASE["H"] // 8, CASE["W"] // 8) and np.isfinite(g["reference_latent"]).all()
    unet, vae, _ = R.build_small(CASE["sd"])
    fp = weight_fingerprint(unet, vae)
    assert abs(fp - float(g["weight_fingerprint"])) <= 1e-9 * abs(fp)
    orc = ElasticOracle(unet, vae, DDIMOracle(), R.embed_fn(True), sd_version=CASE["sd"], view_batch_size=CASE["vbs"], pooled_dim=32)
    orc.seed_everything(CASE["seed"])
    trace = []
    orc.generate_latent("p", "", height=CASE["H"], width=CASE["W"], num_inference_steps=CASE["steps"], resampling_steps=CASE["R"],
                        trace=trace, progress=lambda ts: list(ts)[:CHECKPOINTS[0]], **R.LOOP_KW)
    got, want = trace[CHECKPOINTS[0] - 1], torch.from_numpy(g["oracle_trace"][0])
    # same torch build, same container: bit-identical in practice; the bar leaves room for a host with another BLAS / thread count
    assert float((got - want).norm() / want.norm()) < 1e-5
