"""ORACLE -- CPU restatement of the reference's patched global/local denoising loop.

THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
the product package (elasticdiffusion_official_amd) never does and fails loudly without its HIP library.

What it restates: every glue function of /root/reference/elastic_diffusion.py on the path behind
``ElasticDiffusion.generate_image``, plus the ControlNet threading of
/root/reference/elastic_diffusion_w_controlnet.py, in fp32 torch-CPU / numpy, with the reference's RNG call order
(torch CPU generator + numpy MT19937).  "ED:n" = elastic_diffusion.py line n, "EDC:n" = the ControlNet file.

Pinning: the reference ships no tests or golden vectors.  This oracle is pinned by fixtures under tests/golden/
that were produced by running the REAL reference functions in the builder container (tests/golden/make_golden.py,
stub-imported, deterministic fake UNet/VAE injected) and by tests/test_oracle_vs_reference.py which, when
/root/reference is present, drives both side by side on fresh seeds.  The DDIM scheduler arithmetic is third-party
(diffusers 0.21.4, absent) and restated in oracle/ddim.py -- that part is "parity unpinned".

The model objects (unet / vae / controlnet / scheduler) are injected; the oracle only owns the glue.
"""
import hashlib
import math
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F

LATENT_SCALE = 8  # the ControlNet file hard-codes 8 for pixel<->latent (EDC:946-949)


# --------------------------------------------------------------------------------------------------
# RRG weight schedules (ED:73-107)
# --------------------------------------------------------------------------------------------------
class CosineScheduler:
    """w(i) = factor * (0.5 (1 + cos(pi i / steps))) ** cosine_scale, 0 from ``steps`` on (ED:96-107)."""

    def __init__(self, steps, cosine_scale, factor=0.01):
        self.steps, self.cosine_scale, self.factor = steps, cosine_scale, factor

    def __call__(self, i):
        if i >= self.steps:
            return 0
        return self.factor * ((0.5 * (1 + np.cos(np.pi * i / self.steps))) ** self.cosine_scale)


class LinearScheduler:
    """start + (stop - start) / steps * i, ``stop`` from ``steps`` on (ED:73-82)."""

    def __init__(self, steps, start_val, stop_val):
        self.steps, self.start_val, self.stop_val = steps, start_val, stop_val

    def __call__(self, i):
        if i >= self.steps:
            return self.stop_val
        return self.start_val + (self.stop_val - self.start_val) / self.steps * i


class ConstScheduler(LinearScheduler):
    """``start`` until ``steps``, then ``stop`` (ED:85-94)."""

    def __call__(self, i):
        return self.stop_val if i >= self.steps else self.start_val


# --------------------------------------------------------------------------------------------------
# integer geometry (ED:198-229, 706-757, 943-950, 468-499, 446-465)
# --------------------------------------------------------------------------------------------------
def get_views(height_px, width_px, h_ws=64, w_ws=64, stride=32, scale=8, **_ignored):
    """Regular window grid in latent units, last window shifted back inside (ED:198-229).
    Extra keys of view_config (window_size / context_size) are swallowed like the reference's **kwargs."""
    if height_px % scale or width_px % scale:
        raise ValueError(f"height {height_px} and width {width_px} must be divisible by {scale}")  # ED:200-201
    H, W = height_px // scale, width_px // scale
    n_h = math.ceil((H - h_ws) / stride) + 1 if stride else 1
    n_w = math.ceil((W - w_ws) / stride) + 1 if stride else 1

    def span(k, ws, size):
        lo = int(k * stride)
        hi = lo + ws
        if hi > size:  # "adjust last crop"
            lo, hi = max(0, lo - (hi - size)), size
        return lo, hi

    out = []
    for k in range(int(n_h * n_w)):
        h0, h1 = span(k // n_w, h_ws, H)
        w0, w1 = span(k % n_w, w_ws, W)
        out.append((h0, h1, w0, w1))
    return out


def _axis_context(lo, hi, size, S, n):
    """Index arrays of the context before/after [lo,hi) along one axis (ED:716-744, same rule for both axes)."""
    if lo - n * S < 0:
        before = np.arange(max(0, lo - n * S), lo - S + 1, S)
        want_after = 2 * n - len(before)
        after = np.arange(hi - 1 + S, min(size, hi + want_after * S), S)
    else:
        after = np.arange(hi - 1 + S, min(size, hi + n * S), S)
        want_before = 2 * n - len(after)
        before = np.arange(max(0, lo - want_before * S), lo - S + 1, S)
    return before, after


def crop_with_context(X, a, b, c, d, S, n):
    """Centre [a:b, c:d] plus n strided context rows/cols per side, re-balanced at borders (ED:706-757)."""
    H, W = X.shape[-2:]
    top, bottom = _axis_context(a, b, H, S, n)
    left, right = _axis_context(c, d, W, S, n)
    rows = np.concatenate([top, np.arange(a, b), bottom]).astype(np.int64)
    cols = np.concatenate([left, np.arange(c, d), right]).astype(np.int64)
    crop = X[:, :, torch.from_numpy(rows)][:, :, :, torch.from_numpy(cols)]
    return crop, (len(top), len(bottom), len(left), len(right))


def get_downsample_size(H, W, sd_version, scale=8):
    """Reduced-resolution latent size (ED:943-950)."""
    base = 1024 if "XL" in sd_version else 512
    factor = max(max(H, W) / base, 1)
    return int((H // factor) // scale), int((W // factor) // scale)


def to_even_rational(f, max_block_sz=32):
    """keep/block ratio with both terms even (ED:468-476)."""
    fr = Fraction(f).limit_denominator(max_block_sz)
    if fr.numerator % 2 or fr.denominator % 2:
        fr = Fraction(f).limit_denominator(max_block_sz // 2)
    if fr.numerator % 2 or fr.denominator % 2:
        return fr.numerator * 2, fr.denominator * 2
    return fr.numerator, fr.denominator


def keep_offsets(block_sz, n_remove):
    """Offsets kept inside one block after removing n_remove/2 evenly spread row pairs, and the positions (in the
    kept numbering) of the pair-neighbours that must NOT be OR-merged when the mask is restored (ED:478-499)."""
    pairs = n_remove // 2
    interval = block_sz // (pairs + 1)
    interval += interval % 2
    keep = np.ones(block_sz, dtype=bool)
    marks = []
    for k in range(pairs):
        s = (k + 1) * interval - 1
        marks += [s - 1 - 2 * k, s - 2 * k]
        keep[s:s + 2] = False
    return np.nonzero(keep)[0], np.asarray(marks, dtype=np.int64)


def downsample_axis_tables(n_in, n_out):
    """For one axis: which rows of the 2x-nearest-upsampled input form the 2*n_out grid, and the restore marks
    (ED:568-613).  Returns (sel, marks) with len(sel) == 2*n_out in every case the reference supports."""
    n_keep, block = to_even_rational(n_out / n_in)
    n_blocks = (n_out * 2) // n_keep
    if n_blocks * block > n_in * 2:
        n_blocks -= 1
    covered = n_blocks * block
    offs, marks = keep_offsets(block, block - n_keep)
    sel = (np.arange(0, covered, block)[:, None] + offs[None, :]).reshape(-1)
    sel = sel[sel < n_in * 2]
    remain = n_out * 2 - len(sel)
    tail = np.arange(n_in * 2)[covered:covered + remain] if remain > 0 else np.zeros(0, dtype=np.int64)
    mask_marks = (np.arange(0, n_out * 2, n_keep)[:, None] + marks[None, :]).reshape(-1)
    return np.concatenate([sel, tail]).astype(np.int64), mask_marks.astype(np.int64)


def restore_mask_axis(M, marks, dim):
    """Fold the 2*n_out mask back towards input resolution: neighbouring pairs are OR-ed into one line unless the
    pair starts at a marked position, in which case both lines are kept (ED:446-465)."""
    M = M if dim == 0 else M.T
    out, i, j = [], 0, 0
    while i < M.shape[0]:
        if j < len(marks) and i == marks[j]:
            out.append(M[i])
            out.append(M[i + 1])
            j += 2
        else:
            out.append(M[i] | M[i + 1])
        i += 2
    R = np.stack(out, axis=0)
    return R if dim == 0 else R.T


class ElasticOracle:
    """Same public surface as the reference ``ElasticDiffusion`` for the hot path (ED:110-157, 953-965)."""

    def __init__(self, unet, vae, scheduler, text_embeds_fn=None, sd_version="1.5", view_batch_size=1,
                 pooled_dim=None, controlnet=None, low_vram=False, verbose=False):
        self.device = torch.device("cpu")
        self.unet, self.vae, self.scheduler, self.controlnet = unet, vae, scheduler, controlnet
        self.sd_version = sd_version
        self.view_batch_size = view_batch_size
        self.low_vram = low_vram
        self.verbose = verbose
        self.torch_dtype = torch.float32
        self.pooled_dim = pooled_dim
        self.get_text_embeds = text_embeds_fn
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)  # ED:156
        self.default_size = None
        self.set_view_config()

    # ---- small helpers -------------------------------------------------------------------------
    def set_view_config(self, patch_size=None):  # ED:159-163
        s = self.unet.config.sample_size
        w = patch_size if patch_size is not None else s // 2
        self.view_config = {"window_size": w, "stride": w, "context_size": s - w}

    def seed_everything(self, seed, seed_np=True):  # ED:165-171
        torch.manual_seed(seed)
        if seed_np:
            np.random.seed(seed)

    def get_views(self, h_px, w_px, h_ws=64, w_ws=64, stride=32, **kw):
        return get_views(h_px, w_px, h_ws, w_ws, stride, scale=self.vae_scale_factor)

    def get_downsample_size(self, H, W):
        return get_downsample_size(H, W, self.sd_version, self.vae_scale_factor)

    @staticmethod
    def nearest_interpolate(x, size):  # ED:868-883 (flips are never enabled on the hot path, ED:1071)
        return F.interpolate(x, size=size, mode="nearest")

    def _add_time_ids(self, dtype):  # ED:232-246, 413-418
        ids = list(self.default_size + (0, 0) + self.default_size)
        if self.pooled_dim is not None:
            want = self.unet.add_embedding.linear_1.in_features
            got = self.unet.config.addition_time_embed_dim * len(ids) + self.pooled_dim
            if want != got:
                raise ValueError(f"Model expects an added time embedding vector of length {want}, but {got} was created")
        return torch.tensor([ids], dtype=dtype)

    # ---- pad background (ED:321-391) -----------------------------------------------------------
    @staticmethod
    def string_to_number(s, num_bytes=4):
        return int(hashlib.md5(s.encode()).hexdigest()[: num_bytes * 2], 16)

    def make_denoised_background(self, size, t, id):
        H, W = size
        if H == 0 or W == 0:  # before any RNG use (ED:332-333)
            return torch.zeros(1, 4, H, W)
        self.seed_everything(self.string_to_number(f"{id}_{H}_{W}_{t}"), seed_np=False)
        s = self.vae_scale_factor
        colour = torch.rand(1, 3)[:, :, None, None].repeat(1, 1, H * s, W * s)
        enc = self.vae.encode(colour).latent_dist.sample() * self.vae.config.scaling_factor
        noise = torch.randn_like(enc)
        out = self.scheduler.add_noise(enc, noise, t.long())
        self.seed_everything(np.random.randint(100000), seed_np=False)  # ED:359
        return out

    def background_pad(self, x, pads, t):
        """pads = (left, right, top, bottom); W pair first, then H pair on the already widened tensor (ED:366-391)."""
        B = x.shape[0]
        for k, (before, after) in enumerate(zip(pads[0::2], pads[1::2])):
            dim = 3 - k
            shp_b, shp_a = list(x.shape), list(x.shape)
            shp_b[dim], shp_a[dim] = before, after
            pb = self.make_denoised_background((shp_b[-2], shp_b[-1]), t, f"{dim}_1").repeat(B, 1, 1, 1).to(x)
            pa = self.make_denoised_background((shp_a[-2], shp_a[-1]), t, f"{dim}_2").repeat(B, 1, 1, 1).to(x)
            x = torch.cat([pb, x, pa], dim=dim)
        return x

    # ---- model boundary (ED:393-432, EDC:434-524) ----------------------------------------------
    def unet_step(self, latent, t, text_embeds, add_text_embeds, condition_image=None, controlnet_conditioning_scale=1.0):
        xl = self.sd_version.startswith("XL")
        d = 128 if xl else 64
        latent = self.scheduler.scale_model_input(latent, t)
        h_p, w_p = max(d - latent.shape[-2], 0), max(d - latent.shape[-1], 0)
        l_p, t_p = w_p // 2, h_p // 2
        r_p, b_p = w_p - l_p, h_p - t_p
        x, cond = latent, condition_image
        if h_p or w_p:
            x = self.background_pad(latent, (l_p, r_p, t_p, b_p), t)
            if cond is not None:  # zero pad in pixel space (EDC:457-461)
                k = self.vae_scale_factor
                cond = F.pad(cond, (l_p * k, r_p * k, t_p * k, b_p * k))
        kw = {}
        if xl:
            ids = self._add_time_ids(text_embeds.dtype).repeat(x.shape[0], 1)
            kw["added_cond_kwargs"] = {"text_embeds": add_text_embeds, "time_ids": ids}
        if cond is not None:
            down, mid = self.controlnet(x, t, encoder_hidden_states=text_embeds, controlnet_cond=cond[: x.shape[0]],
                                        conditioning_scale=controlnet_conditioning_scale, guess_mode=False,
                                        return_dict=False, **kw)
            kw["down_block_additional_residuals"], kw["mid_block_additional_residual"] = down, mid
        y = self.unet(x, t, encoder_hidden_states=text_embeds, **kw)["sample"]
        if h_p or w_p:
            y = y[:, :, t_p: y.shape[-2] - b_p, l_p: y.shape[-1] - r_p]
        return y

    def obtain_latent_direction(self, latent, t, text_embeds, add_text_embeds, **cn):  # ED:434-443
        both = self.unet_step(torch.cat([latent, latent]), t, text_embeds, add_text_embeds, **cn)
        uncond, cond = both.chunk(2)
        return cond - uncond, {"uncond_score": uncond, "cond_score": cond}

    # ---- random reduced-resolution pick (ED:501-630) -------------------------------------------
    @staticmethod
    def random_sample_exclude_mask(N, mask=None, hi=4, max_iteration=50):  # ED:501-520
        idx = torch.randint(0, hi, (N,))
        if mask is not None:
            rows = torch.arange(N)
            bad = mask[rows, idx]
            while int(bad.sum()) > 0 and max_iteration > 0:
                idx[bad] = torch.randint(0, hi, (int(bad.sum()),))
                bad = mask[rows, idx]
                max_iteration -= 1
            bad = mask[rows, idx]
            if int(bad.sum()) > 0:  # unconstrained fallback, may repeat a pick (ED:514-518)
                idx[bad] = torch.randint(0, hi, (int(bad.sum()),))
        return idx

    def random_downsample(self, x, exclude_mask=None, prev_random_indices=None, drop_p=0.8, nearest=False):
        """One pick out of every 2x2 block (ED:522-558; the reference's factor is always 2, ED:617)."""
        B, C, H2, W2 = x.shape
        h, w = H2 // 2, W2 // 2
        N = h * w
        if nearest:
            idx = torch.zeros(N, dtype=torch.long)
        else:
            idx = self.random_sample_exclude_mask(N, exclude_mask, hi=4)
        if prev_random_indices is not None:
            drop = torch.randint(0, 101, (N,))
            drop[drop <= 100 * drop_p] = 0
            drop[drop >= 100 * drop_p] = 1
            idx = idx * drop + prev_random_indices * (1 - drop)
        blocks = x.reshape(B, C, h, 2, w, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, N, 4)
        low = torch.gather(blocks, 3, idx.view(1, 1, N, 1).expand(B, C, N, 1)).reshape(B, C, h, w)
        ii, jj = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        q = idx.view(h, w)
        mask = torch.zeros(H2, W2, dtype=torch.bool)
        mask[(ii * 2 + q // 2).reshape(-1), (jj * 2 + q % 2).reshape(-1)] = True
        return low, mask, idx

    def random_nearest_downsample(self, x, downsample_size, prev_random_indices=None, exclude_mask=None,
                                  drop_p=0.8, nearest=False):
        """2x nearest upsample -> drop evenly spread row/col pairs so the grid is exactly 2h x 2w -> 2x2 pick ->
        fold the pick mask back to the input resolution (ED:560-630)."""
        H, W = x.shape[-2:]
        up = F.interpolate(x, size=(2 * H, 2 * W), mode="nearest")
        rsel, rmarks = downsample_axis_tables(H, downsample_size[0])
        csel, cmarks = downsample_axis_tables(W, downsample_size[1])
        grid = up[:, :, torch.from_numpy(rsel)][:, :, :, torch.from_numpy(csel)]
        low, mask2, idx = self.random_downsample(grid, exclude_mask, prev_random_indices, drop_p, nearest)
        m = restore_mask_axis(mask2.numpy(), rmarks, 0)
        m = restore_mask_axis(m, cmarks, 1)
        full = np.zeros((max(H, m.shape[0]), max(W, m.shape[1])), dtype=bool)
        full[: m.shape[0], : m.shape[1]] = m
        return low, torch.from_numpy(full), idx

    def fill_in_from_downsampled_direction(self, target, direction, mask, fill_all=False):  # ED:633-647
        up = F.interpolate(direction, size=target.shape[-2:], mode="nearest")
        target = torch.where(mask, up, target)
        if fill_all:
            target = torch.where(torch.isnan(target), up, target)
        return target

    def approximate_latent_direction_w_resampling(self, latent, t, text_embeds, add_text_embeds, downsample_size,
                                                  resampling_steps=6, drop_p=0.7, **cn):
        """R+1 reduced-resolution CFG evaluations on differently picked pixels, later picks overwrite earlier ones,
        the last fills what is still NaN (ED:649-690)."""
        target = torch.full_like(latent, float("nan")).half()  # promoted to fp32 by the first where (ED:655)
        exclude, prev, info = None, None, {"init_downsampled_latent": None}
        for step in range(resampling_steps + 1):
            low, mask, prev = self.random_nearest_downsample(latent, downsample_size, prev_random_indices=prev,
                                                             exclude_mask=exclude, drop_p=drop_p, nearest=(step == 0))
            if exclude is None:
                exclude = torch.zeros(len(prev), 4, dtype=torch.bool)
            exclude[torch.arange(len(prev)), prev] = True
            if info["init_downsampled_latent"] is None:
                info["init_downsampled_latent"] = low.clone()
            direction, scores = self.obtain_latent_direction(low, t, text_embeds, add_text_embeds, **cn)
            target = self.fill_in_from_downsampled_direction(target, direction, mask, fill_all=(step == resampling_steps))
        info["downsampled_latent"] = low
        info["scores"] = scores
        info["downsampled_direction"] = F.interpolate(target, size=downsample_size, mode="nearest")
        return target, info

    # ---- local unconditional signal (ED:813-864, EDC:916-983) ----------------------------------
    def compute_local_uncond_signal(self, latent, t, uncond_text_embeds, negative_pooled, view_config,
                                    condition_image=None, controlnet_conditioning_scale=1.0):
        Hl, Wl = latent.shape[-2:]
        s = self.vae_scale_factor
        ws, ctx = view_config["window_size"], view_config["context_size"]
        h_ws = Hl if ws + ctx >= Hl else ws  # a dimension that needs no context is one window (ED:820-825)
        w_ws = Wl if ws + ctx >= Wl else ws
        views = self.get_views(Hl * s, Wl * s, h_ws=h_ws, w_ws=w_ws, stride=view_config["stride"])
        out = torch.zeros_like(latent)
        cond_up = None
        if condition_image is not None:
            cond_up = F.interpolate(condition_image[0:1], size=(Hl * s, Wl * s), mode="nearest")  # EDC:932-933
        for b0 in range(0, len(views), self.view_batch_size):
            batch = views[b0:b0 + self.view_batch_size]
            crops, ctxs, cond_crops = [], [], []
            for (h0, h1, w0, w1) in batch:
                crop, n4 = crop_with_context(latent, h0, h1, w0, w1, 1, ctx // 2)
                crops.append(crop)
                ctxs.append(n4)
                if cond_up is not None:
                    k = LATENT_SCALE
                    cc, _ = crop_with_context(cond_up, h0 * k, h1 * k, w0 * k, w1 * k, 1, (ctx * k) // 2)
                    cond_crops.append(cc)
            pred = self.unet_step(torch.cat(crops), t, torch.cat([uncond_text_embeds] * len(batch)),
                                  torch.cat([negative_pooled] * len(batch)),
                                  condition_image=torch.cat(cond_crops) if cond_crops else None,
                                  controlnet_conditioning_scale=controlnet_conditioning_scale)
            for (h0, h1, w0, w1), (n_t, n_b, n_l, n_r), p in zip(batch, ctxs, pred.chunk(len(batch))):
                centre = p[:, :, n_t: p.shape[-2] - n_b, n_l: p.shape[-1] - n_r]
                dst = out[:, :, h0:h1, w0:w1]
                free = dst == 0  # first writer wins, decided by VALUE (ED:859-861)
                dst[free] = centre[free].to(out.dtype)
        return out

    # ---- RePaint re-noising (ED:692-704) -------------------------------------------------------
    def undo_step(self, sample, timestep):
        n = self.scheduler.config.num_train_timesteps // self.scheduler.num_inference_steps
        for k in range(n):
            beta = self.scheduler.betas[timestep + k]
            noise = torch.randn(sample.shape, dtype=sample.dtype)
            sample = (1 - beta) ** 0.5 * sample + beta ** 0.5 * noise
        return sample

    # ---- reduced-resolution guidance (ED:885-940, hot path always passes the cached low-res scores) -----------
    def reduced_resolution_guidance(self, t, latent_x0_original, guidance_scale, rrg_scale, donwsampled_scores):
        low_latent = donwsampled_scores["latent"]
        eps = donwsampled_scores["uncond_score"] + guidance_scale * donwsampled_scores["direction"]
        ddim = self.scheduler.step(eps, t, low_latent)
        x0_up = F.interpolate(ddim["pred_original_sample"], size=latent_x0_original.shape[-2:], mode="nearest")
        grads = []
        for j in range(latent_x0_original.shape[0]):
            with torch.enable_grad():
                probe = latent_x0_original[j:j + 1].clone().detach().requires_grad_(True)
                loss = rrg_scale * F.mse_loss(x0_up[j:j + 1], probe)
                loss.backward()
            grads.append(probe.grad.clone() * -1.0)
        return torch.cat(grads), {"x0": [ddim["pred_original_sample"]], "rrg_latent_out": [ddim["prev_sample"]]}

    # ---- decode (ED:267-310) -------------------------------------------------------------------
    def decode_latents(self, latents):
        latents = latents.to(next(iter(self.vae.post_quant_conv.parameters())).dtype)
        img = self.vae.decode(latents / self.vae.config.scaling_factor).sample
        return (img / 2 + 0.5).clamp(0, 1)

    def tiled_decode(self, latents):
        s = self.vae_scale_factor
        core = self.unet.config.sample_size // 4
        stride = core // 2 if self.low_vram else core
        pad = core if self.low_vram else self.unet.config.sample_size // s * 3
        Hp, Wp = latents.shape[2] * s, latents.shape[3] * s
        views = self.get_views(Hp, Wp, h_ws=core, w_ws=core, stride=stride)
        padded = F.pad(latents, (pad, pad, pad, pad), "constant", 0)
        image = torch.zeros(latents.size(0), 3, Hp, Wp)
        count = torch.zeros_like(image)
        for (h0, h1, w0, w1) in views:  # one tile per VAE call (ED:281)
            tile = self.decode_latents(padded[:, :, h0:h1 + 2 * pad, w0:w1 + 2 * pad])
            c = tile[:, :, pad * s: tile.size(2) - pad * s, pad * s: tile.size(3) - pad * s]
            image[:, :, h0 * s:h1 * s, w0 * s:w1 * s] += c
            count[:, :, h0 * s:h1 * s, w0 * s:w1 * s] += 1
        return image / count

    # ---- condition image (EDC:1005-1033 after VaeImageProcessor.preprocess) --------------------
    @staticmethod
    def prepare_condition(cond_chw01):
        """The benchmark's condition image is synthetic, already a (1,3,8h,8w) float tensor in [0,1]
        (do_normalize=False, EDC:173-175); CFG-doubled like EDC:1029-1031."""
        return torch.cat([cond_chw01.to(torch.float32)] * 2)

    # ---- the loop (ED:953-1130, EDC:1119-1322) -------------------------------------------------
    @torch.no_grad()
    def generate_latent(self, prompts, negative_prompts="", height=768, width=768, num_inference_steps=50,
                        guidance_scale=10.0, resampling_steps=20, new_p=0.3, rrg_stop_t=0.2, rrg_init_weight=1000,
                        rrg_scherduler_cls=CosineScheduler, cosine_scale=3.0, repaint_sampling=True,
                        progress=lambda it: it, condition_image=None, controlnet_conditioning_scale=1.0,
                        trace=None, logs=None):
        downsample_size = self.get_downsample_size(height, width)
        self.default_size = (4 * height, 4 * width)
        vc = self.view_config
        n_rrg = num_inference_steps - int(num_inference_steps * rrg_stop_t)
        if rrg_scherduler_cls is CosineScheduler:
            rrg = CosineScheduler(steps=n_rrg, cosine_scale=cosine_scale, factor=rrg_init_weight)
        else:
            rrg = rrg_scherduler_cls(steps=n_rrg, start_val=rrg_init_weight, stop_val=0)
        if isinstance(prompts, str):
            prompts = [prompts]
        if isinstance(negative_prompts, str):
            negative_prompts = [negative_prompts] * len(prompts)
        un, pun = self.get_text_embeds(negative_prompts)
        co, pco = self.get_text_embeds(prompts)
        text_embeds = torch.cat([un, co])
        add_text_embeds = torch.cat([pun, pco], dim=0)
        s = self.vae_scale_factor
        x = torch.randn((len(prompts), self.unet.config.in_channels, height // s, width // s), dtype=self.torch_dtype)
        self.scheduler.set_timesteps(num_inference_steps)
        cn = {}
        if condition_image is not None:
            cn = dict(condition_image=self.prepare_condition(condition_image),
                      controlnet_conditioning_scale=controlnet_conditioning_scale)
        ts = self.scheduler.timesteps
        for i, t in enumerate(progress(ts)):
            direction, info = self.approximate_latent_direction_w_resampling(
                x, t, text_embeds, add_text_embeds, downsample_size, resampling_steps=resampling_steps,
                drop_p=1 - new_p, **cn)
            if logs is not None and logs.get("init_downsampled_latent") is None:  # ED:1023-1024
                logs["init_downsampled_latent"] = info["init_downsampled_latent"]
            local = self.compute_local_uncond_signal(x, t, un, pun, vc, **cn)
            out = self.scheduler.step(local + guidance_scale * direction, t, x)
            x0, nxt, cfg = out["pred_original_sample"], out["prev_sample"], guidance_scale
            if repaint_sampling and resampling_steps > 0 and i < len(ts) - 1:  # ED:1038-1056
                x = self.undo_step(nxt, ts[i + 1])
                cfg = guidance_scale / 3
                direction, info = self.approximate_latent_direction_w_resampling(
                    x, t, text_embeds, add_text_embeds, downsample_size, resampling_steps=0, drop_p=1 - new_p, **cn)
                local = self.compute_local_uncond_signal(x, t, un, pun, vc, **cn)
                out = self.scheduler.step(local + cfg * direction, t, x)
                x0, nxt = out["pred_original_sample"], out["prev_sample"]
            cascade = torch.zeros_like(nxt)
            if rrg(i) > 10:  # ED:1062
                cascade, _ = self.reduced_resolution_guidance(
                    t, x0, guidance_scale=cfg, rrg_scale=rrg(i),
                    donwsampled_scores={"latent": info["downsampled_latent"],
                                        "uncond_score": info["scores"]["uncond_score"],
                                        "direction": info["downsampled_direction"]})
            x = nxt + cascade
            if trace is not None:
                trace.append(x.clone())
        return x

    @torch.no_grad()
    def generate_image(self, prompts, negative_prompts="", tiled_decoder=False, **kw):
        """-> (float image tensor (B,3,H,W) in [0,1], {}).  PIL conversion is the caller's (ED:1121-1130)."""
        z = self.generate_latent(prompts, negative_prompts, **kw)
        dec = self.tiled_decode if tiled_decoder else self.decode_latents
        return torch.cat([dec(z[i:i + 1]) for i in range(len(z))]), {"latent": z}
