"""MI355X-native drop-in for the hot path of the reference's ``ElasticDiffusion`` class
(/root/reference/elastic_diffusion.py:110-1130, "ED:n" below; ControlNet variant "EDC:n" =
elastic_diffusion_w_controlnet.py): same constructor and ``generate_image`` surface, same RNG stream as the
reference's CPU path, but one image is a *program* (a generator, ``_program``) whose timestep is

    host   : pick-index draws (torch CPU generator, ED:501-544) + pad-strip reseed replay (ED:359)   -- no device sync
    HIP    : ed_assemble_rows       -> ONE model-input batch: K CFG pairs + V views (all d x d)
    yield  : ONE UNet forward for the whole batch (hipGraph replay; rows optionally sharded over ranks and all-gathered
             over RCCL; with several images in flight the rows of all pending calls are fused, per-row timesteps)
    HIP    : ed_phase_epilogue      -> unpad, fill, scatter, CFG + DDIM [, RRG] in one launch
    HIP    : [RePaint] ed_undo_step, then the same phase with K = 1 and guidance/3

instead of the reference's R+1 sequential batch-2 UNet calls, V/view_batch_size view calls and ~hundreds of eager
tensor ops with host syncs per step.  The resampling steps can be batched because their inputs depend only on the RNG,
never on each other's UNet outputs (ED:661-681): the sequential semantics live only in the overwrite order of the
fill, which the epilogue's per-pixel last-covering-step table reproduces.  (``FUSED_GLUE = False`` runs the same steps
through the separate entry points ed_pick_assemble ... ed_rrg_update.)

There is no CPU path in this module: without the HIP library and a ROCm device it raises.
"""
import math
import time

import numpy as np
import torch
import torch.nn as nn

from . import geometry, host_rng, ops
from .graphs import GraphedForward
from .schedule import CosineScheduler, DDIMSchedule
from .sharding import RowSharder


# Fused latent-space glue: ed_assemble_rows before the model call and ed_phase_epilogue (+ RRG) after it -- 2 launches
# per phase instead of 6-7.  False = the separate entry points (kept for A/B and for the per-kernel parity tests).
FUSED_GLUE = True

# Timesteps per pad-strip VAE call (see _strip_frames: a fixed call shape on every rank count)
STRIP_CHUNK = 5


def _identity_progress(it):
    return it


def _default_progress(it):
    """``progress=tqdm`` is the reference's default (ED:963); the identity when tqdm is not importable."""
    try:
        from tqdm import tqdm
    except ImportError:
        return it
    return tqdm(it)


def _make_grid(imgs, nrow=8, padding=2):
    """torchvision.utils.make_grid for a (N,C,H,W) batch with its defaults (what ED:1124 calls): a single image is
    returned as is; otherwise ``min(nrow, N)`` images per row, each cell offset by ``padding`` px of zeros (one shared
    2-px line between neighbours, a 2-px border top/left and bottom/right)."""
    N, C, H, W = imgs.shape
    if N == 1:
        return imgs[0]
    xmaps = min(nrow, N)
    ymaps = -(-N // xmaps)
    h, w = H + padding, W + padding
    grid = imgs.new_zeros(C, h * ymaps + padding, w * xmaps + padding)
    for k in range(N):
        y, x = divmod(k, xmaps)
        grid[:, y * h + padding:y * h + padding + H, x * w + padding:x * w + padding + W] = imgs[k]
    return grid


def _on_own_device(fn):
    """Run a pipeline entry point with the pipeline's device current: hipGraph capture, side streams, events and the
    torch ops of the model all follow the calling thread's current device, and the reference accepts any device
    (``ElasticDiffusion(torch.device('cuda:1'))`` in a process whose current device is 0)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        if self.device.index is None or self.device.index == torch.cuda.current_device():
            return fn(self, *a, **k)
        with torch.cuda.device(self.device):
            return fn(self, *a, **k)

    return wrapper


class _Plan:
    """Per-call geometry: host plans (geometry.py) + their int32 device tables + the pick sampler."""


class _ModelCall:
    """One request of a denoising program to the model boundary: ``rows`` [n,C,d,d] in the model dtype at timestep
    ``t`` (0-d device tensor) with per-row text / pooled / ControlNet-condition rows.  Programs are generators that
    yield these and receive the model output rows (see ``_phase_steps``): the driver decides whether a call runs on its
    own (one image in flight) or fused with the calls of other images (``generate_latents_interleaved``)."""
    __slots__ = ("rows", "t", "text", "pooled", "cond", "direct")

    def __init__(self, rows, t, text, pooled, cond=None, direct=False):
        self.rows, self.t, self.text, self.pooled, self.cond, self.direct = rows, t, text, pooled, cond, direct


class _HostRng:
    """The host generator state of ONE image in flight (torch CPU generator + numpy MT19937).  The reference runs
    images one after the other on the global generators; with several images interleaved, each program must see
    exactly the stream it would have seen alone, so the globals are swapped in and out around every turn."""

    def __init__(self, seed):
        outer = (torch.get_rng_state(), np.random.get_state())
        # CPU generator + numpy only: ``torch.manual_seed`` would also re-seed every device generator, which this
        # path never draws from, and the caller's device RNG state is not ours to change
        torch.default_generator.manual_seed(int(seed))
        np.random.seed(seed)
        self.state = (torch.get_rng_state(), np.random.get_state())
        torch.set_rng_state(outer[0])
        np.random.set_state(outer[1])

    def __enter__(self):
        self.outer = (torch.get_rng_state(), np.random.get_state())
        torch.set_rng_state(self.state[0])
        np.random.set_state(self.state[1])

    def __exit__(self, *exc):
        self.state = (torch.get_rng_state(), np.random.get_state())
        torch.set_rng_state(self.outer[0])
        np.random.set_state(self.outer[1])


class _Stager:
    """Pinned host staging ring: host RNG results are written into pinned memory and uploaded with an async copy;
    a slot is reused only after its copy event has completed, so the host may run ahead of the GPU."""

    def __init__(self, depth=4):
        self.depth = depth
        self.rings = {}
        self.slot_of = {}
        self.waited = 0.0  # seconds the host spent blocked because it was a full ring ahead of the GPU

    def host(self, shape, dtype):
        key = (tuple(shape), dtype)
        ring = self.rings.setdefault(key, {"slots": [], "next": 0})
        k = ring["next"] % self.depth
        ring["next"] += 1
        if k >= len(ring["slots"]):
            buf = torch.empty(shape, dtype=dtype, pin_memory=True)
            ring["slots"].append([buf, None])
            self.slot_of[buf.data_ptr()] = ring["slots"][-1]
        slot = ring["slots"][k]
        if slot[1] is not None:
            w0 = time.perf_counter()
            slot[1].synchronize()  # the async upload that last used this slot must have finished
            self.waited += time.perf_counter() - w0
            slot[1] = None
        return slot[0]

    def upload(self, host_buf, device):
        dev = torch.empty(host_buf.shape, dtype=host_buf.dtype, device=device)
        dev.copy_(host_buf, non_blocking=True)
        with torch.cuda.device(device):  # the copy ran on ``device``'s current stream: record the event there
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(device))
        self.slot_of[host_buf.data_ptr()][1] = ev
        return dev


class ElasticDiffusion(nn.Module):
    """Constructor signature of ED:111-115 plus keyword-only injection points (no pretrained weights, ``diffusers`` or
    network exist in the build environment): ``unet``, ``vae``, ``scheduler``, ``text_encoder`` (callable
    ``prompts -> (embeds, pooled)``), ``controlnet``, ``process_group``."""

    def __init__(self, device, sd_version="2.0", verbose=False, log_freq=5, view_batch_size=1, low_vram=False, *,
                 unet=None, vae=None, scheduler=None, text_encoder=None, controlnet=None, process_group=None,
                 model_dtype=None, weights=None, cache_backgrounds=False, use_graphs=True, residual_fp32=None):
        super().__init__()
        device = torch.device(device)
        if device.type == "cuda" and device.index is None and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
        if device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("elasticdiffusion_official_amd runs the hot path in HIP kernels on an MI355X; "
                               f"device={device} is not a ROCm device and there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test infrastructure only)")
        from . import _hip
        _hip.lib()  # fail here, loudly, if the extension is missing
        self.device = device
        self.sd_version = sd_version
        self.verbose = verbose
        self.log_freq = log_freq
        self.view_batch_size = view_batch_size
        self.low_vram = low_vram
        self.torch_dtype = torch.float32  # latent-space state is always fp32 (ED:121 uses fp16 only for low_vram)
        xl = sd_version.startswith("XL")
        if unet is None or vae is None:
            from .models import build_models
            built_unet, built_vae = build_models(sd_version, device=device, dtype=model_dtype, weights=weights)
            unet = unet if unet is not None else built_unet
            vae = vae if vae is not None else built_vae
        self.unet = self._model_layout(unet.to(device))
        # Precision mode of a 16-bit UNet (round 6).  residual_fp32 = True: the residual stream in fp32 under the 16-bit branches
        # (models._stream32) -- the mode that meets north_star's 1e-3 where plain fp16 does not.  None (default) = by MEASUREMENT, per model
        # family (bench.py's live fp32 leg): the SDXL workloads' plain-fp16 latents end 7.2e-4 ... 9.1e-4 from the
        # fp32 loop's, SD 1.x's 1.09-1.12e-3 -- so the SD 1.x / 2.x family runs the stream mode (9.7e-4, +4.3 % time) and SDXL plain fp16.
        from .models import UNet2DConditionModel as _OwnUNet
        p0 = next(iter(self.unet.parameters()), None) if isinstance(self.unet, torch.nn.Module) else None
        is16 = p0 is not None and p0.dtype in (torch.float16, torch.bfloat16)
        if residual_fp32 is None:
            residual_fp32 = not xl
        self.residual_fp32 = bool(residual_fp32) and is16 and isinstance(self.unet, _OwnUNet)
        if isinstance(self.unet, _OwnUNet):
            self.unet.residual_fp32 = self.residual_fp32
        self.vae = vae.to(device)
        if isinstance(self.vae, torch.nn.Module) and self.device.type == "cuda":
            from .models import prepare_vae_split
            prepare_vae_split(self.vae)     # split fp16 weights of the fp32 VAE's MFMA convolutions, built at load time
        self.controlnet = self._model_layout(controlnet.to(device)) if controlnet is not None else None
        if scheduler is None:
            scheduler = DDIMSchedule.from_config_dir(weights) if weights else DDIMSchedule()  # ED:153
        self.scheduler = scheduler
        if text_encoder is None and weights:
            # real UNet/VAE weights with synthetic prompt embeddings would be prompt-independent garbage: a snapshot
            # must carry its CLIP encoders (ED:145-151); a failure to load them is an error, not a silent fallback
            from .text import load_clip
            text_encoder = load_clip(weights, xl, device)
        self.text_encoder = text_encoder
        self.model_size = 128 if xl else 64  # d_H, d_W of ED:398-400
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)  # ED:156
        self.sharder = RowSharder(process_group)
        # The noised-background frames depend only on (pad geometry, timestep schedule, VAE): with cache_backgrounds
        # they are computed for the first image of a size and reused (the reference's own TODO, ED:340).  Off by
        # default so that every image pays for its own frames.
        self.cache_backgrounds = cache_backgrounds
        self._frame_cache = {}
        # one hipGraph per model batch shape (graphs.py); falls back to eager launches if a capture fails
        self._runner = GraphedForward(self._forward_rows, enabled=use_graphs, prepare=self._text_kv)
        self._time_ids = torch.zeros(1, 6, dtype=torch.float32, device=device)  # persistent: captured by the graphs
        self.set_view_config()
        self.default_size = None
        self._stager = _Stager()
        self.last_latents = None

    @staticmethod
    def _model_layout(module):
        """16-bit UNet / ControlNet weights in the memory format the forward runs in (models.CHANNELS_LAST), so that no
        convolution has to re-lay-out its filter on every call; idempotent, and a no-op for fp32 / injected stand-ins."""
        from . import models
        p = next(module.parameters(), None)
        if models.CHANNELS_LAST and p is not None and p.dtype in (torch.bfloat16, torch.float16):
            module = module.to(memory_format=torch.channels_last)
        return module

    def _mark(self, name):
        """Phase markers (HIP events on the current stream; read only by ``phase_times`` after a sync)."""
        if name == "start":
            self._marks = []
        with torch.cuda.device(self.device):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
        self._marks.append((name, ev))

    def phase_times(self):
        """-> {phase: ms} of the last generate_image call (synchronises)."""
        torch.cuda.synchronize(self.device)
        m = self._marks
        return {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(m[:-1], m[1:])}

    # ---- small API surface the reference's callers use (APP:35-39, ED:159-171, 943-950) -----------------
    def set_view_config(self, patch_size=None):
        s = self.unet.config.sample_size
        w = patch_size if patch_size is not None else s // 2
        self.view_config = {"window_size": w, "stride": w, "context_size": s - w}

    def seed_everything(self, seed, seed_np=True):
        host_rng.seed_everything(seed, seed_np)

    def get_downsample_size(self, H, W):
        return geometry.reduced_size(H, W, self.sd_version, self.vae_scale_factor)

    def get_views(self, panorama_height, panorama_width, h_ws=64, w_ws=64, stride=32, **kwargs):
        s = self.vae_scale_factor
        if panorama_height % s or panorama_width % s:
            raise ValueError(f"height {panorama_height} and Width {panorama_width} must be divisable by {s}")
        rows = geometry.axis_windows(panorama_height // s, h_ws, stride)
        cols = geometry.axis_windows(panorama_width // s, w_ws, stride)
        return [(r[0], r[1], c[0], c[1]) for r in rows for c in cols]

    @property
    def model_dtype(self):
        return next(self.unet.parameters()).dtype

    @torch.no_grad()
    def get_text_embeds(self, prompt):
        """ED:254-265.  With no CLIP weights available the default is a deterministic synthetic embedding per prompt
        string (shape-compatible: (B,77,cross_attention_dim) and the pooled (B,projection_dim))."""
        if self.text_encoder is not None:
            e, p = self.text_encoder(prompt)
            return e.to(self.device), p.to(self.device)
        cfg = self.unet.config
        D = getattr(cfg, "cross_attention_dim", 768)
        P = getattr(cfg, "pooled_projection_dim", None)
        embeds, pooled = [], []
        for s in ([prompt] if isinstance(prompt, str) else prompt):
            g = torch.Generator().manual_seed(host_rng.strip_seed("prompt", 0, 0, 0, s))
            e = torch.randn(1, 77, D, generator=g)
            embeds.append(e)
            pooled.append(torch.randn(1, P, generator=g) if P else e)
        return torch.cat(embeds).to(self.device), torch.cat(pooled).to(self.device)

    # ---- per-image setup ---------------------------------------------------------------------------
    def _dev_i32(self, a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(self.device)

    def _plan(self, height, width):
        s = self.vae_scale_factor
        if height % s or width % s:
            raise ValueError(f"height {height} and Width {width} must be divisable by {s}")  # ED:200-201
        Hl, Wl = height // s, width // s
        h, w = self.get_downsample_size(height, width)
        vc = self.view_config
        P = _Plan()
        P.Hl, P.Wl, P.h, P.w = Hl, Wl, h, w
        P.pick = geometry.PickPlan(Hl, Wl, h, w)
        P.views = geometry.ViewPlan(Hl, Wl, vc["window_size"], vc["stride"], vc["context_size"])
        P.gpad = geometry.PadPlan(h, w, self.model_size)
        P.vpad = geometry.PadPlan(P.views.Sh, P.views.Sw, self.model_size)
        P.one_batch = (P.gpad.PH, P.gpad.PW) == (P.vpad.PH, P.vpad.PW)
        d = self._dev_i32
        P.src_row, P.src_col = d(P.pick.src_row), d(P.pick.src_col)
        P.inv_row, P.inv_col = d(P.pick.inv_row), d(P.pick.inv_col)
        P.up_row, P.up_col = d(P.pick.up_row), d(P.pick.up_col)
        P.down_row, P.down_col = d(P.pick.down_row), d(P.pick.down_col)
        P.win_y0, P.win_x0 = d(P.views.win_y0), d(P.views.win_x0)
        rb, rs, cb, cs = P.views.cover_tables(P.vpad.top, P.vpad.left)
        P.row_blk, P.row_src, P.col_blk, P.col_src = d(rb), d(rs), d(cb), d(cs)
        P.sampler = host_rng.PickSampler(P.pick.N)
        return P

    def _strip_chunk(self, Hs, Ws, T):
        """Timesteps per pad-strip VAE call: ~16 MiB of fp32 pixels, at most STRIP_CHUNK."""
        s = self.vae_scale_factor
        return max(1, min(T, STRIP_CHUNK, (16 << 20) // max(1, 3 * Hs * s * Ws * s * 4)))

    def strip_pools(self, pad, T):
        """-> {(Hs, Ws, chunk): [strip indices]}: the strips of one PadPlan whose encodes share a call shape.  Each pool is
        ONE sharded unit list, i.e. one ``sharder.run`` (one exchange when sharded) per pool and image."""
        pools = {}
        for k, (dim, side_id, Hs, Ws, y0, x0) in enumerate(pad.strips):
            pools.setdefault((Hs, Ws, self._strip_chunk(Hs, Ws, T)), []).append(k)
        return pools

    @torch.no_grad()
    def _strip_frames(self, pad, timesteps, C):
        """All noised-background frames [T,C,PH,PW] of one PadPlan (ED:327-391), computed once per image instead of
        2 VAE encodes per global UNet call (the reference's TODO at ED:340).  Contents depend only on
        (dim, side, size, t); the draws come from md5-seeded private generators (host_rng.strip_draws).
        (Overlapping these encodes with the loop on a side stream was measured in round 2 -- 12.73 vs 12.79 s per image,
        the UNet already saturates the chip -- and removed.)"""
        T = len(timesteps)
        if not pad.padded:
            return None
        key = (pad.h, pad.w, pad.d, C, tuple(int(t) for t in timesteps))
        if self.cache_backgrounds and key in self._frame_cache:
            return self._frame_cache[key]
        dev = self.device
        frames = torch.zeros(T, C, pad.PH, pad.PW, device=dev, dtype=torch.float32)
        sf = self.vae.config.scaling_factor
        s = self.vae_scale_factor
        vae_dtype = next(self.vae.parameters()).dtype
        coef_host = torch.tensor([self.scheduler.add_noise_coefficients(t) for t in timesteps], dtype=torch.float32)
        plans = []
        for (dim, side_id, Hs, Ws, y0, x0) in pad.strips:
            draws = [host_rng.strip_draws(dim, side_id, Hs, Ws, t, C) for t in timesteps]
            # The unit of work is one VAE call over ``chunk`` consecutive timesteps, and EVERY call has exactly that
            # batch (the last chunk repeats the final timestep): the convolution shapes MIOpen sees are then the same on
            # 1, 2, 4 or 8 ranks -- for a shape its find-db has no record of, MIOpen runs a full find on first use (30-130 s
            # per process), which is what round 2's per-rank timestep split (25 -> 21+4 / 13,12 / 7,6 per call) would have
            # paid on the first real multi-GPU run.  ~16 MiB of fp32 pixels per call, at most STRIP_CHUNK timesteps so that a 50-step
            # schedule still splits into >= 10 units for 8 ranks.
            chunk = self._strip_chunk(Hs, Ws, T)
            plans.append((Hs, Ws, y0, x0, chunk, torch.cat([dr[0] for dr in draws]), torch.cat([dr[1] for dr in draws]),
                          torch.cat([dr[2] for dr in draws])))
        coef = coef_host.to(dev)
        # Strips of the same size (the two strips of a padded axis, unless the pad is odd) form ONE pool of units: 2 x 10
        # units on 8 ranks split 3,3,3,3,2,2,2,2 instead of 2 x (2,2,1,1,1,1,1,1) -- 15 % of the encodes on the busiest rank
        # instead of 20 % -- and one exchange per pool instead of one per strip.
        pools = self.strip_pools(pad, T)
        for (Hs, Ws, chunk), members in pools.items():
            n_units = -(-T // chunk)
            # draws of the pool's strips back to back: row p*T + t = strip p of the pool at timestep index t
            colour = torch.cat([plans[k][5] for k in members]).to(dev)
            post = torch.cat([plans[k][6] for k in members]).to(dev)
            fwd = torch.cat([plans[k][7] for k in members]).to(dev)
            # unit (p, u) covers timesteps u*chunk .. u*chunk+chunk-1 of strip p, clamped to T-1 (repeats dropped below)
            t_idx = torch.arange(n_units * chunk, device=dev).clamp_(max=T - 1).view(n_units, chunk)
            sel_all = torch.cat([t_idx + p * T for p in range(len(members))])          # [P * n_units, chunk] draw rows
            t_all = t_idx.repeat(len(members), 1)                                      # ... and their timestep indices

            def encode(sel_rows, _a, _b, _c, t_rows):
                """Noised strips of the units ``sel_rows`` [n, chunk] (units are sharded over ranks like model rows)."""
                outs = []
                for sel, tt in zip(sel_rows, t_rows):
                    img = colour[sel][:, :, None, None].expand(chunk, 3, Hs * s, Ws * s).contiguous().to(vae_dtype)
                    dist = self.vae.encode(img).latent_dist
                    enc = (dist.mean.float() + dist.std.float() * post[sel]) * sf
                    cf = coef[tt]
                    outs.append(cf[:, 0].view(-1, 1, 1, 1) * enc + cf[:, 1].view(-1, 1, 1, 1) * fwd[sel])
                return torch.stack(outs)

            strips = self.sharder.run(encode, sel_all, None, None, None, t_all, out_like=((chunk, C, Hs, Ws), torch.float32))
            strips = strips.reshape(len(members), n_units * chunk, C, Hs, Ws)
            for p, k in enumerate(members):
                y0, x0 = plans[k][2], plans[k][3]
                frames[:, :, y0:y0 + Hs, x0:x0 + Ws] = strips[p, :T]
        if self.cache_backgrounds:
            self._frame_cache[key] = frames
        return frames

    def _embed_rows(self, K, V, un, co, pun, pco):
        """Text rows for one fused model batch: K x [uncond(B), cond(B)] then V x uncond(B)  (ED:436-438, 846-847)."""
        dt = self.model_dtype
        text = torch.cat([un, co] * K + [un] * V).to(dt).contiguous()
        pooled = torch.cat([pun, pco] * K + [pun] * V).to(dt).contiguous()
        return text, pooled

    # ---- model boundary ----------------------------------------------------------------------------
    # The cross-attention k / v of the text rows are the same in every forward of an image: computed once per image and batch
    # shape, outside the hipGraph (models.UNet2DConditionModel.cross_attention_kv; 70 projections per SDXL forward otherwise)
    TEXT_KV_ONCE = True

    def _text_kv(self, txt, into=None):
        """GraphedForward's ``prepare``: {"unet": {...}, "controlnet": {...}} of precomputed cross-attention k|v tensors for
        the text rows ``txt``, or None for models that do not offer it (injected stand-ins) / when switched off."""
        if not self.TEXT_KV_ONCE or not hasattr(self.unet, "cross_attention_kv"):
            return None
        out = {} if into is None else into
        # a part that is missing from ``into`` (a ControlNet attached after this batch shape was captured) is allocated here;
        # the captured graph cannot read it, but the runner's entries are dropped when the ControlNet changes (set_controlnet)
        out["unet"] = self.unet.cross_attention_kv(txt, out.get("unet"))
        if self.controlnet is not None and hasattr(self.controlnet, "cross_attention_kv"):
            out["controlnet"] = self.controlnet.cross_attention_kv(txt, out.get("controlnet"))
        return out

    def _forward_rows(self, x, t_dev, txt, pl, cond, text_kv=None):
        """ED:413-426 / EDC:476-518 for an arbitrary batch of d x d rows (this is what a hipGraph captures)."""
        kw, ckw = {}, {}
        if self.sd_version.startswith("XL"):
            ids = self._time_ids.to(txt.dtype).expand(x.shape[0], -1)
            kw["added_cond_kwargs"] = {"text_embeds": pl, "time_ids": ids}
        if text_kv is not None:
            kw["cross_kv"] = text_kv["unet"]
            ckw["cross_kv"] = text_kv.get("controlnet")
        if cond is not None:
            cn_kw = {k: v for k, v in kw.items() if k != "cross_kv"}
            down, mid = self.controlnet(x, t_dev, encoder_hidden_states=txt, controlnet_cond=cond,
                                        conditioning_scale=self._cn_scale, guess_mode=False, return_dict=False, **cn_kw, **ckw)
            kw["down_block_additional_residuals"], kw["mid_block_additional_residual"] = down, mid
        return self.unet(x, t_dev, encoder_hidden_states=txt, **kw)["sample"].contiguous()

    def _run_model(self, x_rows, t_dev, text, pooled, cond_rows=None, fresh_side=False):
        """Rows are sharded across ranks (sharding.py); each rank's share runs as one (graph-replayed) forward.
        ``t_dev``: 0-d timestep shared by all rows, or one timestep per row (fused batches of several images)."""
        if t_dev.dim() == 0:
            return self.sharder.run(lambda x, txt, pl, cond: self._runner(x, t_dev, txt, pl, cond, fresh_side),
                                    x_rows, text, pooled, cond_rows)
        return self.sharder.run(lambda x, txt, pl, cond, t: self._runner(x, t, txt, pl, cond, fresh_side),
                                x_rows, text, pooled, cond_rows, t_dev)

    # ---- one estimation phase (ED:1016-1035 or ED:1043-1056) ---------------------------------------
    def _phase_steps(self, P, x, ti, K, g, drop_p, emb, cond=None, direct=True, rrg_w=None, rrg_norm=0.0, frames=None):
        """Generator: pre-model glue -> ``yield _ModelCall`` (receives the model output rows) -> post-model glue;
        returns (prev, x0, info).  ``direct``: assemble straight into the hipGraph's static input (one image in flight,
        one rank); otherwise into a scratch batch the driver concatenates / the sharder slices.  ``rrg_w``: this is the
        last phase of a timestep with Reduced-Resolution Guidance active -- with FUSED_GLUE the epilogue then also
        produces info["x_next"] = prev + RRG term (ED:1061-1078)."""
        B, C = x.shape[:2]
        dev, mdt = self.device, self.model_dtype
        n_g, n_v = 2 * K * B, P.views.V * B
        # host draws first (they never wait for the GPU)
        h0 = time.perf_counter()
        idx_host = self._stager.host((K, P.pick.N), torch.uint8)
        stamp_host = self._stager.host((P.pick.N, 4), torch.int8)
        P.sampler.draw(K, drop_p, lambda: host_rng.replay_strip_reseeds(len(P.gpad.strips)), out=idx_host, stamp=stamp_host)
        self.host_s["picks"] += time.perf_counter() - h0
        stamp = self._stager.upload(stamp_host, dev)
        idx = self._stager.upload(idx_host, dev)
        # view batches as the reference forms them: only their pad-strip reseeds are observable (ED:830, 359)
        if P.vpad.strips:
            for _ in range(math.ceil(P.views.V / self.view_batch_size)):
                host_rng.replay_strip_reseeds(len(P.vpad.strips))
        gframes, vframes = frames if frames is not None else (self._gframes, self._vframes)
        gframe = None if gframes is None else gframes[ti]
        vframe = None if vframes is None else vframes[ti]
        low = torch.empty(K, B, C, P.h, P.w, device=dev, dtype=torch.float32)
        direct = direct and P.one_batch and self.sharder.world_size == 1
        if P.one_batch:
            shape = (n_g + n_v, C, P.gpad.PH, P.gpad.PW)
            rows = (self._runner.input_rows(shape, mdt, dev, None if cond is None else cond[K]) if direct
                    else torch.empty(shape, device=dev, dtype=mdt))
            g_rows, v_rows = rows[:n_g], rows[n_g:]
        else:
            g_rows = torch.empty(n_g, C, P.gpad.PH, P.gpad.PW, device=dev, dtype=mdt)
            v_rows = torch.empty(n_v, C, P.vpad.PH, P.vpad.PW, device=dev, dtype=mdt)
        if FUSED_GLUE:
            ops.assemble_rows(x, idx, P.src_row, P.src_col, g_rows, P.h, P.w, P.gpad.top, P.gpad.left, gframe, low,
                              v_rows, P.win_y0, P.win_x0, P.views.Sh, P.views.Sw, P.vpad.top, P.vpad.left, vframe)
        else:
            ops.pick_assemble(x, idx, P.src_row, P.src_col, g_rows, P.h, P.w, P.gpad.top, P.gpad.left, gframe, low)
            ops.gather_views(x, v_rows, P.win_y0, P.win_x0, P.views.Sh, P.views.Sw, P.vpad.top, P.vpad.left, vframe)
        text, pooled = emb[K]
        t_dev = self._t_dev[ti]
        self.host_s["phase_total"] += time.perf_counter() - h0
        if P.one_batch:
            out = yield _ModelCall(rows, t_dev, text, pooled, None if cond is None else cond[K], direct)
            g_out, v_out = out[:n_g], out[n_g:]
        else:
            g_out = yield _ModelCall(g_rows, t_dev, text[:n_g].contiguous(), pooled[:n_g].contiguous(),
                                     None if cond is None else cond[K][0])
            g_out = g_out.clone()  # the two calls may share one graph output buffer when their shapes coincide
            v_out = yield _ModelCall(v_rows, t_dev, text[n_g:].contiguous(), pooled[n_g:].contiguous(),
                                     None if cond is None else cond[K][1])
        h0 = time.perf_counter()
        uncond_last = torch.empty(B, C, P.h, P.w, device=dev, dtype=torch.float32)
        low_dir = torch.empty(B, C, P.h, P.w, device=dev, dtype=torch.float32)
        prev, x0 = torch.empty_like(x), torch.empty_like(x)
        x_next, direction, local = None, None, None
        if FUSED_GLUE:
            keep = self.verbose  # the full-resolution direction / local score are only by-products for the image logs
            direction = torch.empty_like(x) if keep else None
            local = torch.empty_like(x) if keep else None
            x_next = torch.empty_like(x) if rrg_w is not None else None
            ops.phase_epilogue(g_out, v_out, x, stamp,
                               (P.inv_row, P.inv_col, P.up_row, P.up_col, P.down_row, P.down_col),
                               (P.row_blk, P.row_src, P.col_blk, P.col_src), P.views.n_col_blocks,
                               (P.gpad.top, P.gpad.left), K, P.h, P.w, np.float32(g), self._step_coef[ti], prev, x0,
                               low_dir=low_dir, uncond_last=uncond_last, direction=direction, local=local, x_next=x_next,
                               low_latent=low[K - 1] if rrg_w is not None else None, rrg_norm=rrg_norm,
                               rrg_weight=0.0 if rrg_w is None else np.float32(rrg_w))
        else:
            dirs = torch.empty(K, B, C, P.h, P.w, device=dev, dtype=torch.float32)
            ops.unpad_direction(g_out, dirs, uncond_last, P.gpad.top, P.gpad.left)
            direction = torch.empty_like(x)
            ops.fill_directions(dirs, stamp, P.inv_row, P.inv_col, P.up_row, P.up_col, P.down_row, P.down_col, direction,
                                low_dir)
            local = torch.empty_like(x)
            ops.scatter_centres(v_out, local, P.views.n_col_blocks, P.row_blk, P.row_src, P.col_blk, P.col_src)
            ops.cfg_ddim_step(local, direction, x, prev, x0, np.float32(g), *self._step_coef[ti])
        self.host_s["phase_total"] += time.perf_counter() - h0
        info = {"low_latent": low[K - 1], "uncond_score": uncond_last, "low_direction": low_dir,
                "direction": direction, "local": local, "init_low": low[0], "x_next": x_next}
        return prev, x0, info

    def _drive(self, program):
        """Run one program alone: every model call is its own (graph-replayed, row-sharded) forward."""
        try:
            call = next(program)
            while True:
                call = program.send(self._run_model(call.rows, call.t, call.text, call.pooled, call.cond))
        except StopIteration as stop:
            return stop.value

    def _undo(self, x, ti_next):
        """ED:692-704; noise drawn on the host generator in the reference's order, staged through pinned memory."""
        n_sub = self._undo_coef.shape[1]
        h0 = time.perf_counter()
        host = self._stager.host((n_sub,) + tuple(x.shape), torch.float32)
        self.host_s["stager_wait"] += time.perf_counter() - h0
        host_rng.draw_noise_into(host)
        self.host_s["noise"] += time.perf_counter() - h0
        noise = self._stager.upload(host, self.device)
        out = torch.empty_like(x)
        ops.undo_step(x, noise, self._undo_coef[ti_next], out)
        return out

    # ---- ControlNet condition rows (EDC:457-461, 932-949) -- constant over the whole image ------------
    def _condition_rows(self, P, cond_img, B, Ks):
        s = self.vae_scale_factor
        cond_img = cond_img.to(self.device, torch.float32).contiguous()
        assert cond_img.shape[0] == 1 and tuple(cond_img.shape[-2:]) == (P.h * s, P.w * s), \
            "condition image must be (1,3,8h,8w) at the reduced resolution (EDC:1183-1193)"
        mdt = next(self.controlnet.parameters()).dtype
        # global rows: zero pad in pixel space
        gp = P.gpad
        rows_g = np.full((1, gp.PH * s), -1, dtype=np.int32)
        cols_g = np.full((1, gp.PW * s), -1, dtype=np.int32)
        rows_g[0, gp.top * s:(gp.top + P.h) * s] = np.arange(P.h * s)
        cols_g[0, gp.left * s:(gp.left + P.w) * s] = np.arange(P.w * s)
        g_img = torch.empty(1, 3, gp.PH * s, gp.PW * s, device=self.device, dtype=mdt)
        ops.gather2d(cond_img, g_img, self._dev_i32([0]), self._dev_i32(rows_g), self._dev_i32(cols_g))
        # view rows: nearest upsample to full resolution (index map), crop per view at pixel scale, zero pad
        up_r = geometry.nearest_index_map(P.h * s, P.Hl * s)
        up_c = geometry.nearest_index_map(P.w * s, P.Wl * s)
        vp, V = P.vpad, P.views.V
        n = (self.view_config["context_size"] * 8) // 2
        rows_v = np.full((V, vp.PH * s), -1, dtype=np.int32)
        cols_v = np.full((V, vp.PW * s), -1, dtype=np.int32)
        for v, (h0, h1, w0, w1) in enumerate(P.views.views):
            bt, at = geometry.axis_context(h0 * 8, h1 * 8, P.Hl * s, n)
            bl, al = geometry.axis_context(w0 * 8, w1 * 8, P.Wl * s, n)
            rr = up_r[h0 * 8 - bt: h1 * 8 + at]
            cc = up_c[w0 * 8 - bl: w1 * 8 + al]
            rows_v[v, vp.top * s: vp.top * s + len(rr)] = rr
            cols_v[v, vp.left * s: vp.left * s + len(cc)] = cc
        v_img = torch.empty(V, 3, vp.PH * s, vp.PW * s, device=self.device, dtype=mdt)
        ops.gather2d(cond_img, v_img, self._dev_i32([0] * V), self._dev_i32(rows_v), self._dev_i32(cols_v))
        out = {}
        for K in Ks:
            g_part = g_img.expand(2 * K * B, -1, -1, -1)
            v_part = v_img.repeat_interleave(B, dim=0)
            out[K] = torch.cat([g_part, v_part]).contiguous() if P.one_batch else (g_part.contiguous(), v_part.contiguous())
        return out

    # ---- the loop (ED:953-1078) --------------------------------------------------------------------
    def _setup_run(self, height, width, num_inference_steps, guidance_scale, resampling_steps, new_p, rrg_stop_t,
                   rrg_init_weight, rrg_scherduler_cls, cosine_scale, repaint_sampling, controlnet_conditioning_scale):
        """Everything of one ``generate_image`` call that does not depend on the prompt, the seed or the condition
        image: geometry tables, schedules, the noised pad-background frames.  Shared by all images in flight."""
        self._mark("start")
        self.host_s = {"picks": 0.0, "phase_total": 0.0, "noise": 0.0, "stager_wait": 0.0}
        self._stager.waited = 0.0
        S = _Plan()
        S.P = P = self._plan(height, width)
        self.default_size = (4 * height, 4 * width)  # ED:969
        n_rrg = num_inference_steps - int(num_inference_steps * rrg_stop_t)
        if rrg_scherduler_cls is CosineScheduler or getattr(rrg_scherduler_cls, "__name__", "") == "CosineScheduler":
            S.rrg = rrg_scherduler_cls(steps=n_rrg, cosine_scale=cosine_scale, factor=rrg_init_weight)
        else:
            S.rrg = rrg_scherduler_cls(steps=n_rrg, start_val=rrg_init_weight, stop_val=0)
        S.C = C = self.unet.config.in_channels
        dev = self.device
        ts = self.scheduler.set_timesteps(num_inference_steps)
        S.T = len(ts)
        self._timesteps = list(ts)
        self._t_dev = ts.to(dev)
        self._step_coef = [self.scheduler.step_coefficients(t) for t in ts]
        S.R = R = int(resampling_steps)
        if not 0 <= R <= host_rng.MAX_RESAMPLING_STEPS:
            raise ValueError(f"resampling_steps must be in [0, {host_rng.MAX_RESAMPLING_STEPS}] (the per-pixel "
                             f"last-covering-step table is int8), got {resampling_steps}")
        S.repaint = bool(repaint_sampling) and R > 0
        if S.repaint:
            # undo_step is only ever entered with timesteps[i+1] (ED:1040): row 0 is a placeholder
            rows = [self.scheduler.undo_coefficients(t) for t in ts[1:]]
            self._undo_coef = torch.stack(rows[:1] + rows).to(dev) if rows else None
        d0, d1 = self.default_size
        self._time_ids.copy_(torch.tensor([[d0, d1, 0, 0, d0, d1]], dtype=torch.float32))  # ED:232-246, 414-418
        self._gframes = self._strip_frames(P.gpad, self._timesteps, C)
        self._vframes = self._strip_frames(P.vpad, self._timesteps, C)
        self._mark("setup_done")
        S.Ks = sorted({R + 1, 1} if S.repaint else {R + 1})
        if getattr(self, "_cn_scale", controlnet_conditioning_scale) != controlnet_conditioning_scale:
            self._runner.entries.clear()  # the scale is a constant inside captured graphs
        self._cn_scale = controlnet_conditioning_scale
        S.guidance, S.drop_p = guidance_scale, 1 - new_p
        S.norm = np.float32(2.0 / (C * P.Hl * P.Wl))
        return S

    def _program(self, S, prompts, negative_prompts, condition_image=None, trace=None, progress=_identity_progress,
                 direct=True, frames=None):
        """Generator: the denoising loop of ONE image (ED:981-1078), yielding ``_ModelCall``s; returns the final latent.
        All host RNG draws happen inside, in the reference's order."""
        P = S.P
        if isinstance(prompts, str):
            prompts = [prompts]
        if isinstance(negative_prompts, str):
            negative_prompts = [negative_prompts] * len(prompts)
        un, pun = self.get_text_embeds(negative_prompts)
        co, pco = self.get_text_embeds(prompts)
        B = len(prompts)
        # initial latent from the host generator (ED:998-1000)
        x_host = self._stager.host((B, S.C, P.Hl, P.Wl), torch.float32)
        x_host.normal_()
        x = self._stager.upload(x_host, self.device)
        emb = {K: self._embed_rows(K, P.views.V, un, co, pun, pco) for K in S.Ks}
        cond = None
        if condition_image is not None:
            if self.controlnet is None:
                raise ValueError("condition_image given but no controlnet was supplied")
            cond = self._condition_rows(P, condition_image, B, S.Ks)
        logs = {"x0": [], "rrg_x0": [], "init_low": None, "embeds": (un, co, pun, pco)} if self.verbose else None
        self._logs = logs
        for i, t in enumerate(progress(self._timesteps)):
            w_i = S.rrg(i)
            rrg_w = w_i if w_i > 10 else None  # ED:1061-1062
            two_phase = S.repaint and i < S.T - 1
            prev, x0, info = yield from self._phase_steps(P, x, i, S.R + 1, S.guidance, S.drop_p, emb, cond, direct,
                                                          None if two_phase else rrg_w, S.norm, frames)
            if logs is not None and logs["init_low"] is None:
                # ED:1023-1024: taken right after the FIRST direction estimate, i.e. before the RePaint phase replaces
                # ``info`` with the one of the undone sample (ED:1043)
                logs["init_low"] = info["init_low"].clone()
            cfg = S.guidance
            if two_phase:  # ED:1038-1056
                x = self._undo(prev, i + 1)
                cfg = S.guidance / 3
                prev, x0, info = yield from self._phase_steps(P, x, i, 1, cfg, S.drop_p, emb, cond, direct, rrg_w, S.norm,
                                                              frames)
            if logs is not None:  # verbose image logs (ED:1058-1059, 1073-1076)
                if i % self.log_freq == 0:
                    logs["x0"].append(x0.clone())
                    if rrg_w is not None:  # the reduced-resolution x0 the guidance pulls towards (ED:909-921)
                        sb, sa = self._step_coef[i][0], self._step_coef[i][1]
                        eps = info["uncond_score"] + np.float32(cfg) * info["low_direction"]
                        logs["rrg_x0"].append((info["low_latent"] - np.float32(sb) * eps) / np.float32(sa))
            if info["x_next"] is not None:  # RRG already applied inside the fused epilogue
                x = info["x_next"]
            elif rrg_w is not None:  # ED:1061-1078
                sb, sa = self._step_coef[i][0], self._step_coef[i][1]
                nxt = torch.empty_like(prev)
                ops.rrg_update(prev, x0, info["low_latent"], info["uncond_score"], info["low_direction"], P.up_row,
                               P.up_col, nxt, np.float32(cfg), sb, sa, S.norm, np.float32(w_i))
                x = nxt
            else:
                x = prev
            if trace is not None:
                trace.append(x.clone())
        return x

    @_on_own_device
    @torch.no_grad()
    def generate_latents(self, prompts, negative_prompts="", height=768, width=768, num_inference_steps=50,
                         guidance_scale=10.0, resampling_steps=20, new_p=0.3, rrg_stop_t=0.2, rrg_init_weight=1000,
                         rrg_scherduler_cls=CosineScheduler, cosine_scale=3.0, repaint_sampling=True,
                         progress=_identity_progress, condition_image=None, controlnet_conditioning_scale=1.0,
                         trace=None):
        S = self._setup_run(height, width, num_inference_steps, guidance_scale, resampling_steps, new_p, rrg_stop_t,
                            rrg_init_weight, rrg_scherduler_cls, cosine_scale, repaint_sampling,
                            controlnet_conditioning_scale)
        self._runner.new_image()
        x = self._drive(self._program(S, prompts, negative_prompts, condition_image, trace, progress, direct=True))
        self.last_latents = x
        self.host_s["blocked_ahead_of_gpu"] = self._stager.waited
        self._mark("loop_done")
        return x

    @_on_own_device
    @torch.no_grad()
    def generate_latents_interleaved(self, jobs, in_flight=2, height=768, width=768, num_inference_steps=50,
                                     guidance_scale=10.0, resampling_steps=20, new_p=0.3, rrg_stop_t=0.2,
                                     rrg_init_weight=1000, rrg_scherduler_cls=CosineScheduler, cosine_scale=3.0,
                                     repaint_sampling=True, controlnet_conditioning_scale=1.0, on_done=None):
        """Several images of the SAME size / settings in flight at once (new; the reference has nothing like it).

        ``jobs`` = list of dicts {prompts, negative_prompts="", seed, condition_image=None}.  Each job is one
        ``_program`` with its own host RNG stream (``_HostRng``: exactly the stream ``seed_everything(seed)`` +
        ``generate_latents`` would consume, so every image's latents are those of running it alone, up to the model's
        own batch-shape dependent rounding).  Per tick, the pending model calls of all live programs -- e.g. the 20-row
        phase A batch of one image and the 6-row RePaint batch of another -- are concatenated into ONE forward, which
        is then row-sharded over the ranks like any other batch.  Why: with N ranks a lone image leaves 20/N and 6/N rows
        per rank, where the UNet runs at a fraction of its batch-20 rate; m images in flight multiply the rows per
        forward by ~m while the exchange stays one all-gather per tick.  Returns the final latents in job order
        (``on_done(index, latent)`` is called as each finishes, e.g. to decode it)."""
        S = self._setup_run(height, width, num_inference_steps, guidance_scale, resampling_steps, new_p, rrg_stop_t,
                            rrg_init_weight, rrg_scherduler_cls, cosine_scale, repaint_sampling,
                            controlnet_conditioning_scale)
        self._runner.new_image()
        results = [None] * len(jobs)
        queue = list(range(len(jobs)))
        live = []  # [job index, program, rng, pending call]

        started = [0]

        def start(j):
            job = jobs[j]
            rng = _HostRng(job["seed"])
            # Every image pays for its own noised pad-background frames, exactly as when it runs alone (they depend only
            # on geometry and schedule, so the first job takes the ones _setup_run just made and each later job
            # recomputes identical ones); cache_backgrounds=True is the explicit opt-in to share them (ADVICE r2:
            # otherwise the N-GPU bench line amortises 100 VAE encodes the 1-GPU line pays per image)
            with torch.cuda.device(self.device):  # a job's clock starts BEFORE its pad-strip encodes
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream(self.device))
            self.job_events[j] = [ev0, None]
            frames = None
            if started[0] and not self.cache_backgrounds:
                frames = (self._strip_frames(S.P.gpad, self._timesteps, S.C), self._strip_frames(S.P.vpad, self._timesteps, S.C))
            started[0] += 1
            prog = self._program(S, job["prompts"], job.get("negative_prompts", ""), job.get("condition_image"),
                                 direct=False, frames=frames)
            with rng:
                call = next(prog)
            live.append([j, prog, rng, call])

        self.job_events = {}  # job -> [start event, end event] on this device's stream (``job_latencies``)
        while queue and len(live) < max(1, in_flight):
            start(queue.pop(0))
        self.ticks = 0
        while live:
            # one fused forward per group of shape-compatible pending calls (normally exactly one group)
            groups = {}
            for ent in live:
                c = ent[3]
                groups.setdefault((tuple(c.rows.shape[1:]), c.cond is None), []).append(ent)
            for ents in groups.values():
                calls = [e[3] for e in ents]
                sizes = [c.rows.shape[0] for c in calls]
                if len(calls) == 1:
                    c = calls[0]
                    outs = [self._run_model(c.rows, c.t, c.text, c.pooled, c.cond, fresh_side=True)]
                else:
                    rows = torch.cat([c.rows for c in calls])
                    t = torch.cat([c.t.reshape(1).expand(n) for c, n in zip(calls, sizes)])
                    text = torch.cat([c.text for c in calls])
                    pooled = torch.cat([c.pooled for c in calls])
                    cond = None if calls[0].cond is None else torch.cat([c.cond for c in calls])
                    outs = self._run_model(rows, t, text, pooled, cond, fresh_side=True).split(sizes)
                self.ticks += 1
                for ent, out in zip(ents, outs):
                    j, prog, rng = ent[0], ent[1], ent[2]
                    try:
                        with rng:
                            ent[3] = prog.send(out)
                    except StopIteration as stop:
                        results[j] = stop.value
                        live.remove(ent)
                        if on_done is not None:
                            on_done(j, stop.value)
                        with torch.cuda.device(self.device):
                            ev1 = torch.cuda.Event(enable_timing=True)
                            ev1.record(torch.cuda.current_stream(self.device))
                        self.job_events[j][1] = ev1
                        if queue:
                            start(queue.pop(0))
        self.last_latents = results[-1] if results else None
        self.host_s["blocked_ahead_of_gpu"] = self._stager.waited
        self._mark("loop_done")
        return results

    def job_latencies(self, jobs=None):
        """-> seconds from the start of each job of the last ``generate_latents_interleaved`` call (its first pad-strip
        encode; the first job's frames are made by the shared setup just before) to the end of its ``on_done`` (decode), in
        job order; synchronises.  ``jobs``: only these job indices (a rank that did not decode a job has no decode inside
        that job's interval).  With m images in flight the throughput is ~m images per latency: bench.py reports both so a
        throughput-scaling line cannot be read as a per-image speed-up."""
        torch.cuda.synchronize(self.device)
        return [1e-3 * a.elapsed_time(b) for j, (a, b) in sorted(self.job_events.items())
                if b is not None and (jobs is None or j in jobs)]

    @_on_own_device
    @torch.no_grad()
    def generate(self, latent, text_embeds, add_text_embeds, guidance_scale=7.5):
        """ED:761-796: plain CFG + DDIM generation of ``latent`` (B,C,h,w; the reference calls it on the initial
        reduced-resolution latent for its ``verbose`` image log) over the scheduler's current timesteps.
        ``text_embeds`` / ``add_text_embeds`` = cat([uncond, cond]) like the reference's.  A latent smaller than the
        model's native size is padded with the noised-background frames (ED:404-411).
        -> (PIL image of the first sample, {"inter_x0": [pred_original_sample every log_freq steps]})"""
        dev, mdt = self.device, self.model_dtype
        x = latent.to(dev, torch.float32).contiguous()
        B, C, h, w = x.shape
        if self.scheduler.timesteps is None:
            raise RuntimeError("generate(): call scheduler.set_timesteps / generate_image first (ED:776 iterates them)")
        ts = list(self.scheduler.timesteps)
        pad = geometry.PadPlan(h, w, self.model_size)
        frames = self._strip_frames(pad, ts, C)
        xl = self.sd_version.startswith("XL")
        text = text_embeds.to(dev, mdt).contiguous()
        pooled = add_text_embeds.to(dev, mdt).contiguous() if xl else text
        if self.default_size is None:
            self.default_size = (4 * h * self.vae_scale_factor, 4 * w * self.vae_scale_factor)
        d0, d1 = self.default_size
        self._time_ids.copy_(torch.tensor([[d0, d1, 0, 0, d0, d1]], dtype=torch.float32))
        self._runner.new_image()
        inter = []
        for i, t in enumerate(ts):
            if pad.padded:  # the reference re-seeds the global generators once per pad strip (ED:359)
                host_rng.replay_strip_reseeds(len(pad.strips))
            rows = torch.empty(2 * B, C, pad.PH, pad.PW, device=dev, dtype=mdt)
            if frames is not None:
                rows.copy_(frames[i].to(mdt).expand(2 * B, -1, -1, -1))
            rows[:, :, pad.top:pad.top + h, pad.left:pad.left + w] = torch.cat([x, x]).to(mdt)
            out = self._run_model(rows, torch.as_tensor(t, device=dev), text, pooled)
            out = out[:, :, pad.top:pad.top + h, pad.left:pad.left + w].float()
            uncond, cnd = out[:B].contiguous(), out[B:].contiguous()
            prev, x0 = torch.empty_like(x), torch.empty_like(x)
            ops.cfg_ddim_step(uncond, (cnd - uncond).contiguous(), x, prev, x0, np.float32(guidance_scale),
                              *self.scheduler.step_coefficients(t))
            x = prev
            if i % self.log_freq == 0:
                inter.append(x0.cpu())
        return self._to_pil(self.decode_latents(x))[0], {"inter_x0": inter}

    @staticmethod
    def _to_pil(imgs):
        from PIL import Image
        arr = imgs.mul(255).byte().permute(0, 2, 3, 1).cpu().numpy()  # what ToPILImage does for float CHW (ED:1125)
        return [Image.fromarray(a) for a in arr]

    def _image_log(self, decode_fn, guidance_scale):
        """ED:1092-1118: the ``verbose`` image log of the last single-image run (grids via make_grid's defaults)."""
        logs, image_log = getattr(self, "_logs", None), {}
        if not logs:
            return image_log

        def grid_of(latents):
            dec = torch.cat([decode_fn(z[k:k + 1]) for z in latents for k in range(len(z))])
            return self._to_pil(_make_grid(dec.clamp(0, 1))[None])[0]

        if logs["init_low"] is not None:
            un, co, pun, pco = logs["embeds"]
            image_log["global_img"], info = self.generate(logs["init_low"], torch.cat([un, co]), torch.cat([pun, pco]),
                                                          guidance_scale=guidance_scale)
            if info["inter_x0"]:
                image_log["global_img_inter_x0_imgs"] = grid_of([z.to(self.device) for z in info["inter_x0"]])
        if logs["x0"]:
            image_log["intermediate_x0_imgs"] = grid_of(logs["x0"])
        image_log["intermediate_cascade_x0_imgs"] = {}
        if logs["rrg_x0"]:
            image_log["intermediate_cascade_x0_imgs"]["rrg"] = grid_of(logs["rrg_x0"])
        return image_log

    # ---- decode (ED:267-310) -----------------------------------------------------------------------
    @_on_own_device
    @torch.no_grad()
    def decode_latents(self, latents):
        vdt = next(self.vae.parameters()).dtype
        img = self.vae.decode((latents / self.vae.config.scaling_factor).to(vdt)).sample
        return (img.float() / 2 + 0.5).clamp(0, 1)

    @_on_own_device
    @torch.no_grad()
    def tiled_decode(self, latents, tile_batch=8):
        """ED:275-310 with the tiles gathered in one launch, decoded ``tile_batch`` at a time (and sharded over ranks),
        and accumulated + normalised in one launch."""
        B, C, Hl, Wl = latents.shape
        s = self.vae_scale_factor
        tp = geometry.TilePlan(Hl, Wl, self.unet.config.sample_size, s, low_vram=self.low_vram)
        vdt = next(self.vae.parameters()).dtype
        tiles = torch.empty(tp.T * B, C, tp.Ts, tp.Ts, device=self.device, dtype=vdt)
        ops.tile_gather_pad(latents.contiguous(), tiles, self._dev_i32(tp.tile_y0), self._dev_i32(tp.tile_x0),
                            self.vae.config.scaling_factor)

        def dec(rows, *_):
            outs = [self.vae.decode(rows[a:a + tile_batch]).sample for a in range(0, rows.shape[0], tile_batch)]
            return torch.cat(outs).contiguous()

        decoded = self.sharder.run(dec, tiles, None, None, None, out_like=((3, tp.Ts * s, tp.Ts * s), vdt))
        image = torch.empty(B, decoded.shape[1], Hl * s, Wl * s, device=self.device, dtype=torch.float32)
        rt, rs, ct, cs = tp.pixel_tables()
        ops.tile_accumulate_normalise(decoded, image, tp.n_col_tiles, self._dev_i32(rt), self._dev_i32(rs),
                                      self._dev_i32(ct), self._dev_i32(cs))
        return image

    @_on_own_device
    @torch.no_grad()
    def generate_image(self, prompts, negative_prompts="", height=768, width=768, num_inference_steps=50,
                       guidance_scale=10.0, resampling_steps=20, new_p=0.3, rrg_stop_t=0.2, rrg_init_weight=1000,
                       rrg_scherduler_cls=CosineScheduler, cosine_scale=3.0, repaint_sampling=True,
                       progress=_default_progress, tiled_decoder=False, grid=False, *, condition_image=None,
                       controlnet_conditioning_scale=1.0, output_type="pil"):
        """ED:953-965 signature (ControlNet keywords of EDC:1120-1134 are keyword-only here).
        Returns ``(images, image_log)``; images are PIL by default, a float tensor with ``output_type='pt'``."""
        z = self.generate_latents(prompts, negative_prompts, height, width, num_inference_steps, guidance_scale,
                                  resampling_steps, new_p, rrg_stop_t, rrg_init_weight, rrg_scherduler_cls,
                                  cosine_scale, repaint_sampling, progress, condition_image,
                                  controlnet_conditioning_scale)
        dec = self.tiled_decode if tiled_decoder else self.decode_latents
        image_log = self._image_log(dec, guidance_scale) if self.verbose else {}  # ED:1092-1118 (before the final decode)
        imgs = torch.cat([dec(z[i:i + 1]) for i in range(len(z))])  # decode_bs = 1 (ED:1090, 1121)
        self._mark("decode_done")
        if grid:
            imgs = _make_grid(imgs)[None]  # ED:1124: torchvision make_grid(imgs, nrow=8, padding=2, pad_value=0)
        if output_type == "pt":
            return imgs, image_log
        return self._to_pil(imgs), image_log


class ElasticDiffusionControlNet(ElasticDiffusion):
    """Drop-in for the ControlNet variant (/root/reference/elastic_diffusion_w_controlnet.py:118-196, 1119-1322):
    ``controlnet_model`` is the third positional constructor argument and ``condition_image`` /
    ``controlnet_conditioning_scale`` sit where the reference has them in ``generate_image``.

    ``condition_image`` is the already pre-processed condition (float tensor (1,3,8h,8w) in [0,1] at the reduced
    resolution, EDC:1183-1193; a PIL image / numpy array of that size is converted).  The canny / depth extraction of
    ``process_condition_image`` (EDC:1102-1117: cv2 / a HF depth pipeline) is pre-processing outside the loop and out
    of scope; ``process_condition_image`` raises with that explanation."""

    def __init__(self, device, sd_version="2.0", controlnet_model="canny", verbose=False, log_freq=5,
                 view_batch_size=1, low_vram=False, *, controlnet=None, **kw):
        if controlnet is None:
            from .models import build_models
            dev = torch.device(device)
            unet, vae, controlnet = build_models(sd_version, device=dev, dtype=kw.get("model_dtype"), controlnet=True,
                                                 weights=kw.get("weights"))
            kw.setdefault("unet", unet)
            kw.setdefault("vae", vae)
        super().__init__(device, sd_version, verbose, log_freq, view_batch_size, low_vram, controlnet=controlnet, **kw)
        self.controlnet_model = controlnet_model

    def process_condition_image(self, condition_image, controlnet_model):
        raise NotImplementedError("canny / depth extraction (EDC:1102-1117) is pre-processing outside the hot path; "
                                  "pass the processed condition image to generate_image")

    def _to_condition_tensor(self, image, h_px, w_px):
        if isinstance(image, torch.Tensor):
            t = image.float()
            t = t[None] if t.dim() == 3 else t
        else:  # PIL image or HWC uint8 array -> what VaeImageProcessor.preprocess(do_normalize=False) yields:
            # RGB, Lanczos resize to the requested size, [0,1] floats (EDC:173-175, 1017)
            if hasattr(image, "convert"):
                image = image.convert("RGB")
                if image.size != (w_px, h_px):
                    from PIL import Image
                    image = image.resize((w_px, h_px), resample=Image.LANCZOS)
            arr = np.asarray(image)
            t = torch.from_numpy(np.ascontiguousarray(arr)).float().div(255.0).permute(2, 0, 1)[None]
        if tuple(t.shape[-2:]) != (h_px, w_px):
            t = torch.nn.functional.interpolate(t, size=(h_px, w_px), mode="bilinear", align_corners=False)
        return t

    @torch.no_grad()
    def generate_image(self, prompts, negative_prompts="", condition_image=None, height=768, width=768,
                       num_inference_steps=50, guidance_scale=10.0, controlnet_conditioning_scale=1.0,
                       resampling_steps=20, new_p=0.3, rrg_stop_t=0.2, rrg_init_weight=1000,
                       rrg_scherduler_cls=CosineScheduler, cosine_scale=3.0, repaint_sampling=True,
                       progress=_default_progress, tiled_decoder=False, grid=False, *, output_type="pil"):
        if condition_image is None:
            raise ValueError("condition_image is required (EDC:1183-1193)")
        h, w = self.get_downsample_size(height, width)
        s = self.vae_scale_factor
        cond = self._to_condition_tensor(condition_image, h * s, w * s)
        return super().generate_image(prompts, negative_prompts, height, width, num_inference_steps, guidance_scale,
                                      resampling_steps, new_p, rrg_stop_t, rrg_init_weight, rrg_scherduler_cls,
                                      cosine_scale, repaint_sampling, progress, tiled_decoder, grid,
                                      condition_image=cond, controlnet_conditioning_scale=controlnet_conditioning_scale,
                                      output_type=output_type)
