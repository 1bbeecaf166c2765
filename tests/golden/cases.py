"""Case tables shared by tests/golden/make_golden.py (writer) and the tests (readers)."""
import torch

# (height_px, width_px, unet sample_size, patch_size or None)   -- SURVEY 8(c) G1
G1_CASES = [
    (512, 512, 64, None),      # cfg1
    (512, 1024, 64, None),     # cfg2
    (1024, 2048, 128, None),   # cfg3 / cfg5
    (2048, 2048, 128, None),   # cfg4
    (1080, 1920, 64, None),    # latent 135x240: windows do not tile -> overlapping centres
    (768, 2048, 64, None),
    (768, 768, 64, 48),
    (1536, 1536, 128, 96),
    (1024, 1024, 128, 120),
    (1024, 1536, 128, 32),
    (256, 512, 64, None),      # a dimension smaller than the model size
    (520, 776, 64, None),      # latent 65x97
    (536, 776, 64, None),      # latent 67x97 (the reference supports this one end to end)
    (1536, 2048, 128, 64),
]

# name -> (latent H, latent W, reduced h, reduced w, seed)   -- G2: ratios 1, 2, 135->72, 96->64, mixed
G2_CASES = {
    "ratio1": (16, 24, 16, 24, 0),
    "ratio2": (32, 64, 16, 32, 1),
    "r135_72": (135, 240, 36, 64, 2),
    "r96_64": (96, 96, 64, 64, 3),
    "r65_97": (65, 97, 42, 64, 4),
}

# function-level chain on one timestep: direction -> local -> ddim -> undo -> rrg   -- G3..G6
G3_CASES = {
    "sd_cfg2_R3": dict(sd="1.5", sample=64, vbs=4, H=512, W=1024, steps=50, ti=3, R=3, seed=0, rrg_w=1000.0),
    "sd_cfg1_R0": dict(sd="1.5", sample=64, vbs=1, H=512, W=512, steps=10, ti=0, R=0, seed=1, rrg_w=437.5),
    "sd_overlap_R2": dict(sd="1.5", sample=64, vbs=3, H=536, W=776, steps=50, ti=10, R=2, seed=2, rrg_w=250.0),
    "xl_pad_R2": dict(sd="XL1.0", sample=128, vbs=16, H=512, W=768, steps=50, ti=5, R=2, seed=3, rrg_w=4000.0),
    "sd_patch48_R1": dict(sd="1.5", sample=64, vbs=2, H=768, W=768, steps=20, ti=2, R=1, seed=4, patch=48, rrg_w=90.0),
}

# name -> (latent H, latent W, unet sample_size, low_vram geometry, seed)   -- G7
G7_CASES = {
    "tiles_exact": (16, 24, 32, False, 0),
    "tiles_ragged": (20, 28, 32, False, 1),
    "tiles_lowvram": (20, 28, 32, True, 2),
    # cfg4 size: 64 tiles of 128x128 latents (32-latent cores + 48-latent halo), 2048x2048 px image -> stored as probes
    "tiles_cfg4_256": (256, 256, 128, False, 3),
}

# images larger than this many elements are stored as ``image_probe`` vectors instead of in full
FULL_IMAGE_MAX = 1 << 20


def image_probe(img):
    """Compact fingerprint of a large decoded image (B,3,H,W): 16x16 block means, a strided sample, and the full
    rows / columns either side of the first few tile seams (core = 256 px) -- seams are where a tiled-decode bug shows.
    Same function on the writer (reference output) and the reader (HIP output) side."""
    import torch.nn.functional as F
    img = img.detach().float().cpu()
    H, W = img.shape[-2:]
    seams_r = [r for r in (255, 256, 511, 512, H - 257, H - 256) if 0 <= r < H]
    seams_c = [c for c in (255, 256, 767, 768, W - 257, W - 256) if 0 <= c < W]
    return {"pool16": F.avg_pool2d(img, 16), "stride": img[..., 5::16, 11::16].contiguous(),
            "seam_rows": img[..., seams_r, :].contiguous(), "seam_cols": img[..., :, seams_c].contiguous()}

# G11: the reference's plain CFG+DDIM ``generate`` (ED:761-796; used for its verbose "global_img" log) on a reduced
# latent that needs padding (32x64 in a 64x64 model) and on one that does not
G11_CASES = {
    "gen_sd_pad_32x64": dict(sd="1.5", sample=64, h=32, w=64, steps=4, seed=21, guidance=7.5),
    "gen_xl_64x128": dict(sd="XL1.0", sample=128, h=64, w=128, steps=3, seed=22, guidance=10.0),
    "gen_sd_nopad_64x64": dict(sd="1.5", sample=64, h=64, w=64, steps=3, seed=23, guidance=5.0),
}

E2E_KW = dict(guidance_scale=10.0, new_p=0.3, rrg_stop_t=0.4, rrg_init_weight=1000, cosine_scale=10.0,
              repaint_sampling=True)

# G8 (end-to-end latents), G9 (rng trace where trace=True), G10 (controlnet=True)
E2E_CASES = {
    "cfg1_sd_512": dict(sd="1.5", sample=64, vbs=1, H=512, W=512, steps=10, R=0, seed=0, trace=True),
    "cfg2_sd_512x1024": dict(sd="1.5", sample=64, vbs=4, H=512, W=1024, steps=4, R=3, seed=0, trace=True),
    "cfg2_seed1_vbs2": dict(sd="1.5", sample=64, vbs=2, H=512, W=1024, steps=3, R=7, seed=1),
    "cfg3_xl_1024x2048": dict(sd="XL1.0", sample=128, vbs=16, H=1024, W=2048, steps=2, R=2, seed=0, trace=True),
    "overlap_536x776": dict(sd="1.5", sample=64, vbs=4, H=536, W=776, steps=3, R=2, seed=2),
    "tall_1024x512_norepaint": dict(sd="1.5", sample=64, vbs=4, H=1024, W=512, steps=3, R=2, seed=3,
                                    kw=dict(repaint_sampling=False)),
    "patch48_768": dict(sd="1.5", sample=64, vbs=4, H=768, W=768, steps=3, R=1, seed=4, patch=48,
                        kw=dict(rrg_init_weight=4000, rrg_stop_t=0.2, cosine_scale=3.0)),
    "xl_single_view_512x1024": dict(sd="XL1.0", sample=128, vbs=16, H=512, W=1024, steps=2, R=1, seed=5),
    "tiled_sd_640x512": dict(sd="1.5", sample=32, vbs=4, H=320, W=256, steps=2, R=1, seed=6, tiled=True,
                             keep_image=True),
    "cn_sd_512x1024": dict(sd="1.5", sample=64, vbs=4, H=512, W=1024, steps=3, R=2, seed=7, controlnet=True),
    "cfg5_cn_xl_1024x2048": dict(sd="XL1.0", sample=128, vbs=16, H=1024, W=2048, steps=2, R=1, seed=8,
                                 controlnet=True),
    # BASELINE.json configs[3]: SDXL 2048x2048, 16 views, tiled decode (64 tiles); image stored as probes
    "cfg4_xl_2048x2048_tiled": dict(sd="XL1.0", sample=128, vbs=16, H=2048, W=2048, steps=2, R=2, seed=9, tiled=True,
                                    keep_image=True, kw=dict(rrg_init_weight=4000)),
    # non-tiled decode image compared too (only latents were)
    "cfg2_image_sd_512x1024": dict(sd="1.5", sample=64, vbs=4, H=512, W=1024, steps=2, R=1, seed=10, keep_image=True),
}


def synthetic_condition(h_px, w_px):
    """Synthetic ControlNet condition image: smooth RGB gradient in [0,1], (1,3,h,w)."""
    ys = torch.linspace(0, 1, h_px).view(1, 1, h_px, 1)
    xs = torch.linspace(0, 1, w_px).view(1, 1, 1, w_px)
    r = ys.expand(1, 1, h_px, w_px)
    g = xs.expand(1, 1, h_px, w_px)
    b = (0.5 + 0.5 * torch.sin(6.0 * (ys + xs))).expand(1, 1, h_px, w_px)
    return torch.cat([r, g, b], dim=1).contiguous()


def assert_image_matches(g, name, img, atol):
    """Compare a decoded image with the stored golden: in full when ``<name>/image`` exists, through
    ``image_probe`` vectors otherwise."""
    import numpy as np
    img = img.detach().float().cpu()
    if f"{name}/image" in g.files:
        np.testing.assert_allclose(img.numpy(), g[f"{name}/image"], rtol=0, atol=atol)
        return
    probes = image_probe(img)
    assert f"{name}/image_probe/pool16" in g.files, f"no golden image for {name}"
    for k, v in probes.items():
        np.testing.assert_allclose(v.numpy(), g[f"{name}/image_probe/{k}"], rtol=0, atol=atol, err_msg=k)
