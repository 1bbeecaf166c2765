"""Torch-ROCm model modules behind the hot path's model boundary (``unet_step``, elastic_diffusion.py:393-432;
``decode_latents`` :267-272; ``make_denoised_background`` :350; ControlNet elastic_diffusion_w_controlnet.py:476-518).

The reference takes these from ``diffusers==0.21.4`` (``UNet2DConditionModel``, ``AutoencoderKL``,
``ControlNetModel``), which is neither vendored nor installable here, and no pretrained weights exist offline.  These
are from-scratch pure-torch modules with the SD-1.x / SD-2.x-base / SDXL-base architecture hyper-parameters and
HF-layout parameter names (so a ``diffusion_pytorch_model.safetensors`` state dict loads with ``load_weights``).  For
benchmarks they are randomly initialised from a fixed seed ("synthetic" in bench.py).

This is plumbing, not the product: dense contractions (conv / linear / attention) go to MIOpen / hipBLASLt / SDPA,
i.e. the MFMA pipes, through PyTorch-ROCm; the memory-bound glue between them (GroupNorm+SiLU, GroupNorm->token
layout, GEGLU) is fused into hand-written HIP kernels (csrc/unet_kernels.hip) for 16-bit activations.
"""
import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# Fused HIP kernels inside the UNet (csrc/unet_kernels.hip): used for 16-bit CUDA activations; anything else (the fp32
# VAE, the CPU-baseline copy of the UNet, odd shapes) takes the plain torch ops.  Toggle for A/B measurements.
FUSED_KERNELS = True


# Channels-last activations end to end: MIOpen's CK convolutions are NHWC kernels, and an NCHW problem costs a layout
# transpose on either side of every convolution (batched_transpose_*: 5.9 ms of a 178 ms batch-20 forward,
# profiles/r2_s5_unet_fwd_b20_nchw_kernel_stats.csv).  With channels-last activations those launches disappear, GroupNorm's
# token-layout output and the transformer's closing residual add become plain views / contiguous adds, and the 1x1
# shortcut convolutions are plain GEMMs.  Measured (hipGraph replay, profiles/r2_s5_probe_channels_last.jsonl):
# 168.6 vs 174.9 ms at batch 20, 60.0 vs 61.4 ms at batch 6.  The in-tree MIOpen db carries NHWC entries derived from the
# tuned NCHW ones (tools/miopen_nhwc_from_nchw.py).  ED_CHANNELS_LAST=0 switches back (A/B).
CHANNELS_LAST = os.environ.get("ED_CHANNELS_LAST", "1") == "1"


def _fusable(x):
    return FUSED_KERNELS and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.is_contiguous()


def _fusable_nhwc(x):
    return (FUSED_KERNELS and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 4
            and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last))


# ---- fp32 residual stream under a 16-bit model (round 6, ``UNet2DConditionModel.residual_fp32``) ---------------------------------------
# The tolerance mode for configurations whose fp16 latent ends outside north_star's 1e-3 (cfg2, SD 1.5 512 x 1024: 1.09e-3 against the
# fp32 loop, profiles/bench_r6_s1_cfg2_fp32leg.json): the tensor that persists from block to block -- x in `x = x + branch(norm(x))` -- is
# kept in fp32, every branch still computes in the 16-bit model dtype (same MFMA kernels, same weights), i.e. the adds no longer round
# (a third of the 16-bit loop's drift by profiles/r5_precision_attribution.json).  Everything that READS the stream takes fp32 in and
# hands the model dtype on: the normalisation layers through torch ops in this mode (their HIP kernels are 16-bit in / out), the
# up / down-sampling and shortcut convolutions after a cast; everything that WRITES it adds a 16-bit branch result to the fp32 stream.
def _stream32(x, module_dtype):
    """is ``x`` an fp32 residual stream feeding a module whose parameters are 16-bit?"""
    return x.dtype == torch.float32 and module_dtype in (torch.float16, torch.bfloat16)


def _f32_params(norm):
    """fp32 copies of a normalisation layer's (weight, bias), rebuilt when they change"""
    w, b = norm.weight, norm.bias
    key = (id(w), w.data_ptr(), w._version, id(b), b._version)
    hit = norm.__dict__.get("_p32")
    if hit is None or hit[0] != key:
        hit = (key, w.detach().float(), b.detach().float())
        norm.__dict__["_p32"] = hit
    return hit[1], hit[2]


def _f32_bias(mod):
    """fp32 copy of a module's bias (None without one), rebuilt when it changes"""
    b = mod.bias
    if b is None:
        return None
    key = (id(b), b.data_ptr(), b._version)
    hit = mod.__dict__.get("_b32")
    if hit is None or hit[0] != key:
        hit = (key, b.detach().float())
        mod.__dict__["_b32"] = hit
    return hit[1]


def group_norm_act(norm, x, silu=False, tokens=False, chan_bias=None, conv_bias=None):
    """GroupNorm [+ SiLU] [-> (N, H*W, C) token layout] of ``x``, or of ``(x + conv_bias[c]) + chan_bias[n, c]``: the
    bias of the (bias-free) convolution that produced x and the time-embedding add of ResnetBlock2D, folded into the
    kernel.  HIP: ed_groupnorm / ed_groupnorm_nhwc; torch otherwise."""
    N, C, H, W = x.shape
    if _stream32(x, norm.weight.dtype):   # fp32 residual stream: statistics and affine in fp32, the result in the model dtype
        dt = norm.weight.dtype
        if conv_bias is not None:
            x = x + conv_bias[None, :, None, None]
        if chan_bias is not None:
            x = x + chan_bias[:, :, None, None]
        cpg32 = C // norm.num_groups
        if (FUSED_KERNELS and x.is_cuda and C % 8 == 0 and cpg32 >= 8 and C <= 4096 and norm.num_groups <= 256
                and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)):
            from . import ops      # HIP: ed_groupnorm_nhwc_s32 (fp32 stream in, model dtype out; the 16-bit kernel's three launches)
            y = ops.groupnorm_nhwc_s32(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu)
            return y.permute(0, 2, 3, 1).reshape(N, H * W, C) if tokens else y
        w32, b32 = _f32_params(norm)
        y = F.group_norm(x, norm.num_groups, w32, b32, norm.eps)
        if silu:
            y = F.silu(y)
        y = y.to(dt)
        if CHANNELS_LAST and y.is_cuda:
            y = y.contiguous(memory_format=torch.channels_last)
        return y.permute(0, 2, 3, 1).reshape(N, H * W, C) if tokens else y
    cpg = C // norm.num_groups
    nhwc = _fusable_nhwc(x) and C % 8 == 0 and cpg >= 8
    if (chan_bias is not None or conv_bias is not None) and not (
            FUSED_TEMB_ADD and (nhwc or (_fusable(x) and (H * W) % 8 == 0 and not tokens))):
        if conv_bias is not None:
            x = x + conv_bias[None, :, None, None]
        if chan_bias is not None:
            x = x + chan_bias[:, :, None, None]
        chan_bias = conv_bias = None
        nhwc = _fusable_nhwc(x) and C % 8 == 0 and cpg >= 8
    if nhwc:
        from . import ops
        y = ops.groupnorm_nhwc(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu,
                               chan_bias=None if chan_bias is None else chan_bias.contiguous(), conv_bias=conv_bias)
        return y.permute(0, 2, 3, 1).reshape(N, H * W, C) if tokens else y  # a view: NHWC memory is the token layout
    if chan_bias is not None or conv_bias is not None:
        from . import ops
        return ops.groupnorm(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu,
                             chan_bias=None if chan_bias is None else chan_bias.contiguous(), conv_bias=conv_bias)
    if _fusable(x) and (H * W) % 8 == 0 and (not tokens or cpg % 4 == 0):
        from . import ops
        return ops.groupnorm(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu, tokens=tokens)
    if (VAE_SPLIT_CONV and FUSED_KERNELS and x.is_cuda and x.dtype == torch.float32 and not tokens and not x.is_contiguous()
            and x.is_contiguous(memory_format=torch.channels_last) and norm.weight.dtype == torch.float32):
        from . import ops  # the channels-last fp32 VAE (VAE_SPLIT_CONV): the normalisation layers that do not feed a split convolution
        if ops.groupnorm_nhwc_f32_ok(C, norm.num_groups):
            return ops.groupnorm_nhwc_f32(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu)
    if (VAE_HIP_GROUPNORM and FUSED_KERNELS and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and not tokens
            and (H * W) % 4 == 0 and norm.weight.dtype == torch.float32 and N * norm.num_groups <= 65535):
        from . import ops  # the fp32 VAE: split statistics + apply (+SiLU) instead of torch's one-block-per-group moments
        return ops.groupnorm_f32(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu=silu)
    y = F.group_norm(x, norm.num_groups, norm.weight, norm.bias, norm.eps)
    if silu:
        y = F.silu(y)
    return y.permute(0, 2, 3, 1).reshape(N, H * W, C) if tokens else y


FUSED_LAYERNORM = True
FUSED_TEMB_ADD = True      # ResnetBlock2D: h + temb folded into norm2 (ed_groupnorm chan_bias)
FUSED_CONV_BIAS = True     # ResnetBlock2D: bias-free convolutions, biases folded into norm2 / the closing residual add
FUSED_ADD_LAYERNORM = True  # BasicTransformerBlock: residual add + next LayerNorm in one kernel (ed_add_layernorm)
FUSED_TOKENS_ADD = True     # Transformer2DModel: tokens -> NCHW + residual in one kernel (ed_tokens_add_nchw)
FLASH_ATTENTION = True      # Attention: ed_flash_attention (head_dim 64, 16-bit) instead of SDPA / AOTriton
FUSED_QKV = True            # Attention: one projection GEMM for q,k,v (self) / k,v (cross)
HIP_GEGLU_GEMM = True       # GEGLU: projection GEMM with the gated product in its epilogue (ed_geglu_gemm) instead of hipBLASLt + ed_geglu
HIP_LINEAR = True           # the projections where ed_linear measured faster than hipBLASLt (ops.linear_wins)
FUSED_SKIP_CAT = True        # up blocks: ResnetBlock2D(cat([hidden, skip])) without the concatenated tensor (ed_groupnorm_nhwc_cat + a K-split shortcut)
FUSED_PROJ_OUT_ADD = True    # Transformer2DModel (channels-last): the closing x + proj_out(h) in ed_linear's residual epilogue
# Downsample2D (stride 2, padding 1), channels-last, as ed_conv3x3_nhwc_s2 instead of MIOpen's CK kernel: correct (tests) and a TIE in the forward
# (profiles/r6_s10_switch_ab_downsample.jsonl: +0.3 % at 40 rows, +0.2 % at 12, -0.1 % at 20, -0.2 % at 6 -- N = 320 fills 62 % of its two
# 256-column tiles), so the library call stays; the entry point and this switch are kept for A/B
HIP_DOWNSAMPLE_CONV = False
# BasicTransformerBlock: x + to_out(...) / x + ff(...) inside ed_linear's epilogue where that kernel runs the projection (the 640-channel level),
# the following LayerNorm then a plain one.  Measured NO gain in the forward (profiles/r6_s11_switch_ab_residual_linear.jsonl: -0.4 % at 40 rows,
# -0.3 % at 12, +0.1 % at 20, 0 at 6): the epilogue's residual read costs the projection what ed_add_layernorm -> ed_layernorm saves
# (profiles/r6_s11_addmm_probe.jsonl: [20480, 1280 -> 1280] projection + add-LayerNorm 94.8 us, own projection with residual + LayerNorm 96.7,
# library beta = 1 + LayerNorm 103.7).  Off; kept as an A/B switch.
FUSED_RESIDUAL_LINEAR = False
FUSED_UPSAMPLE_CONV = True   # Upsample2D: nearest 2x + conv 3x3 as one ed_conv3x3_nhwc_up2x launch (the A operand is gathered from the source)
HIP_CONV3X3 = True          # ResnetBlock2D / Upsample2D 3x3 convolutions, channels-last: ed_conv3x3_nhwc (+bias, +temb, +residual) instead of MIOpen
VAE_HIP_GROUPNORM = True    # VAE GroupNorm(+SiLU), fp32 NCHW: ed_groupnorm_f32 instead of torch's moments + affine + SiLU kernels
VAE_HIP_ATTENTION = True    # VAE mid-block attention: fp32 GEMM + ed_softmax_rows + fp32 GEMM instead of SDPA (AOTriton)
# _VaeAttention's residual: `x + attn` (True: the VAE stays NCHW -- ed_groupnorm_f32 everywhere, MIOpen's NCHW solvers) or
# `attn + x` (False, rounds 1-3: the permuted first operand makes the sum, and so the rest of the encoder / decoder,
# channels-last -- MIOpen's NHWC fp32 igemm convolutions, torch GroupNorm through strided copies).  Same values bit for
# bit.  Measured on the MI355X, each layout with its own MIOpen find (profiles/r4_s1_vae_layout_ab.jsonl): 8 decode tiles of
# 128 x 128 latents 979 -> 760 ms, the 128 x 256 decode 202 -> 198 ms, 5 pad-strip encodes 54.9 -> 54.7 ms.
VAE_NCHW_RESIDUAL = True
# Round 5: the VAE's ResnetBlock convolutions (94 % of the encoder's and 71 % of the decoder's convolution FLOPs) on the 16-bit MFMA
# pipe at fp32 accuracy: the activation after GroupNorm + SiLU and the weights are each carried as an fp16 (hi, lo) pair and
# x.w = xh.wh + xl.wh + xh.wl is ONE ed_conv3x3_nhwc main loop over 3 Cin channels with an fp32 epilogue (csrc/vae_kernels.hip).  The
# ResnetBlocks then pass channels-last fp32 activations to each other (the kernel's layout); the stream-fed up / down-sampling
# convolutions, conv_in / conv_out and the mid-block attention stay with the fp32 libraries in NCHW (_to_nchw: one layout copy per block
# boundary).  As accurate against fp64 as the library's fp32 convolution (tests/test_vae_split.py).
VAE_SPLIT_CONV = True
VAE_SPLIT_DOWNSAMPLE = True   # round 6: the encoder's stride-2 downsamplers on the split-operand path as well (ed_conv3x3_nhwc_f32out_s2)
# A/B switch from the environment: ED_DISABLE=FLASH_ATTENTION,FUSED_QKV,... turns the named module switches off
for _name in filter(None, os.environ.get("ED_DISABLE", "").split(",")):
    if _name not in globals() or not isinstance(globals()[_name], bool):
        raise RuntimeError(f"ED_DISABLE: unknown switch {_name!r}")
    globals()[_name] = False


def fused_unet_entry_points():
    """C-ABI entry points a 16-bit UNet forward reaches with the current switches (the real-architecture parity test
    checks that they were actually launched, i.e. that no torch fallback silently took over).  ed_linear / ed_conv3x3_nhwc
    take only the shapes they win on (ops.linear_wins / conv3x3_ok), so they are not REQUIRED of every model; where the
    convolution kernel runs, its epilogue has replaced ed_bias_residual_add."""
    hip_conv = FUSED_KERNELS and HIP_CONV3X3 and CHANNELS_LAST
    on = [("ed_groupnorm_nhwc" if CHANNELS_LAST else "ed_groupnorm", FUSED_KERNELS),
          ("ed_geglu_gemm" if HIP_GEGLU_GEMM else "ed_geglu", FUSED_KERNELS),
          ("ed_layernorm", FUSED_KERNELS and FUSED_LAYERNORM),
          ("ed_flash_attention", FUSED_KERNELS and FLASH_ATTENTION),
          ("ed_add_layernorm", FUSED_KERNELS and FUSED_ADD_LAYERNORM),
          ("ed_tokens_add_nchw", FUSED_KERNELS and FUSED_TOKENS_ADD and not CHANNELS_LAST),
          ("ed_bias_residual_add", FUSED_KERNELS and FUSED_CONV_BIAS and FUSED_TEMB_ADD and not hip_conv)]
    return {n for n, flag in on if flag}


def linear_(x, weight, bias=None):
    """F.linear, through ed_linear for the 16-bit shapes it measured faster on (ops.linear_wins), hipBLASLt otherwise."""
    if HIP_LINEAR and FUSED_KERNELS and _fusable(x) and weight.dtype == x.dtype and weight.is_contiguous():
        from . import ops
        if ops.linear_wins(x.numel() // x.shape[-1], x.shape[-1], weight.shape[0]):
            return ops.linear(x, weight, bias)
    return F.linear(x, weight, bias)


def linear_res_(x, weight, bias, residual):
    """(F.linear(x, weight, bias) + residual, True) in ONE launch where ed_linear takes the projection (its residual epilogue: bias + product +
    residual rounded once), else (F.linear(x, weight, bias), False) and the caller adds -- the transformer blocks' `x = x + branch(...)`: with
    the add inside the projection, the LayerNorm that follows reads and writes 2 units of traffic instead of ed_add_layernorm's 4."""
    if (FUSED_RESIDUAL_LINEAR and residual is not None and HIP_LINEAR and FUSED_KERNELS and _fusable(x) and _fusable(residual)
            and weight.dtype == x.dtype == residual.dtype and weight.is_contiguous()
            and residual.shape == x.shape[:-1] + (weight.shape[0],)):
        from . import ops
        if ops.linear_wins(x.numel() // x.shape[-1], x.shape[-1], weight.shape[0]):
            return ops.linear(x, weight, bias, residual=residual), True
    return linear_(x, weight, bias), False


def _hip_conv3x3(x, conv, shape_only=False):
    """Can ``conv`` (3x3, stride 1, padding 1) on the channels-last 16-bit activation ``x`` run as ed_conv3x3_nhwc?
    ``shape_only``: ``x`` only stands for the shape / dtype of an activation that does not exist yet."""
    if not (HIP_CONV3X3 and (shape_only or _fusable_nhwc(x))):
        return False
    w = conv.weight
    if w.dtype != x.dtype or not w.is_contiguous(memory_format=torch.channels_last):
        return False
    from . import ops
    B, C, H, W = x.shape
    return ops.conv3x3_wins(B, H, W, C, w.shape[0])


def _to_nchw(x):
    """The split VAE blocks (VAE_SPLIT_CONV) hand on channels-last fp32 activations; everything around them -- the stream-fed up / down-
    sampling convolutions, the mid-block attention, conv_norm_out / conv_out -- runs NCHW, the layout the in-tree MIOpen find-db has
    fp32 records for (a shape without a record costs a minutes-long find on first use).  One layout copy per block boundary."""
    return x if x.is_contiguous() else x.contiguous()


def _vae_split_ok(x, blk):
    """Can this fp32 VAE ResnetBlock2D (no time embedding) run its two convolutions as split-fp16 MFMA convolutions on ``x``?"""
    if not (VAE_SPLIT_CONV and FUSED_KERNELS and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and blk.time_emb_proj is None and blk.conv1.weight.dtype == torch.float32):
        return False
    from . import ops
    B, cin, H, W = x.shape
    cout = blk.conv1.out_channels
    return (ops.groupnorm_nhwc_f32_ok(cin, blk.norm1.num_groups) and ops.groupnorm_nhwc_f32_ok(cout, blk.norm2.num_groups)
            and cin % 64 == 0 and cout % 64 == 0 and H * W * 3 * max(cin, cout) * 2 < 2 ** 31 - 16)


def _split_weight(conv):
    """(split fp16 weight, 2^-k) of a Conv2d for ops.conv3x3_f32out, rebuilt when the weight changes"""
    w = conv.weight
    # id(w): a fresh Parameter the caching allocator placed at the freed address (same data_ptr, version 0) must not hit
    key = (id(w), w.data_ptr(), w._version, str(w.device))
    hit = conv.__dict__.get("_split_w")
    if hit is None or hit[0] != key:
        from . import ops
        hit = (key,) + ops.split_conv_weight(w)
        conv.__dict__["_split_w"] = hit
    return hit[1], hit[2]


def prepare_vae_split(vae):
    """Build the split fp16 weights of every convolution of an fp32 VAE that can take the split-operand path NOW (model load), so that
    the one host synchronisation of ``ops.split_conv_weight`` (the weight's absolute maximum) never falls into a forward -- or a stream
    capture -- later.  A no-op for a CPU / 16-bit VAE or with the switch off; a weight that changes afterwards is re-split
    on its next use (``_split_weight``'s key)."""
    if not (VAE_SPLIT_CONV and FUSED_KERNELS):
        return 0
    n = 0
    for m in vae.modules():
        convs = ()
        if isinstance(m, ResnetBlock2D) and m.time_emb_proj is None:
            convs = (m.conv1, m.conv2)
        elif isinstance(m, Upsample2D) and m.vae:
            convs = (m.conv,)
        elif isinstance(m, Downsample2D) and m.padding == 0:     # the VAE encoder's downsamplers (VAE_SPLIT_DOWNSAMPLE)
            convs = (m.conv,)
        for c in convs:
            w = c.weight
            if w.is_cuda and w.dtype == torch.float32 and w.shape[1] % 64 == 0 and w.shape[0] % 8 == 0 and tuple(w.shape[2:]) == (3, 3):
                _split_weight(c)
                n += 1
    return n


def layer_norm(norm, x):
    """LayerNorm over the last dim.  HIP: ed_layernorm (one wave per row); torch otherwise."""
    D = x.shape[-1]
    if norm.elementwise_affine and _stream32(x, norm.weight.dtype):   # fp32 residual stream -> model dtype
        if FUSED_LAYERNORM and FUSED_KERNELS and x.is_cuda and x.is_contiguous() and D % 8 == 0 and D <= 2048:
            from . import ops
            return ops.layernorm_s32(x, norm.weight, norm.bias, norm.eps)
        w32, b32 = _f32_params(norm)
        return F.layer_norm(x, (D,), w32, b32, norm.eps).to(norm.weight.dtype)
    if FUSED_LAYERNORM and _fusable(x) and D % 8 == 0 and D <= 2048 and norm.elementwise_affine:
        from . import ops
        return ops.layernorm(x, norm.weight, norm.bias, norm.eps)
    return norm(x)


def add_layer_norm(norm, a, b):
    """(a + b, LayerNorm(a + b)).  HIP: ed_add_layernorm; torch otherwise."""
    D = a.shape[-1]
    if (FUSED_ADD_LAYERNORM and _fusable(a) and _fusable(b) and a.shape == b.shape and D % 8 == 0 and D <= 2048
            and norm.elementwise_affine):
        from . import ops
        return ops.add_layernorm(a, b, norm.weight, norm.bias, norm.eps)
    if (FUSED_ADD_LAYERNORM and FUSED_KERNELS and norm.elementwise_affine and _fusable(a) and _stream32(b, a.dtype) and b.is_cuda
            and b.is_contiguous() and a.shape == b.shape and D % 8 == 0 and D <= 2048 and norm.weight.dtype == a.dtype):
        from . import ops      # fp32 residual stream: 16-bit branch result + fp32 stream -> (fp32 stream, 16-bit LayerNorm)
        return ops.add_layernorm_s32(a, b, norm.weight, norm.bias, norm.eps)
    s = a + b          # (fp32 residual stream: a 16-bit branch result + the fp32 stream -> fp32)
    return s, layer_norm(norm, s)


class ModelOutput(dict):
    """``out['sample']`` and ``out.sample`` (the reference indexes the former, ED:422)."""

    __getattr__ = dict.__getitem__


# ---------------------------------------------------------------------------------------------------
# shared blocks
# ---------------------------------------------------------------------------------------------------
def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000.0):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    sin, cos = torch.sin(emb), torch.cos(emb)
    return torch.cat([cos, sin], dim=-1) if flip_sin_to_cos else torch.cat([sin, cos], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


# 1x1 shortcut convolutions as one strided-batched GEMM with the residual folded in (beta = 1): replaces MIOpen's
# im2col + GEMM + bias kernels and the separate residual add.  Toggle for A/B measurements.
SHORTCUT_AS_GEMM = False  # measured on MI355X: 233.1 vs 231.1 ms at B=20 -- no gain over MIOpen + add, kept for A/B


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim=None, eps=1e-5, groups=32):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout) if temb_dim else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        if _vae_split_ok(x, self):
            return self._forward_split(x)
        tb = self.time_emb_proj(F.silu(temb)) if self.time_emb_proj is not None else None
        sc = self.conv_shortcut
        cout = self.conv1.out_channels
        if _stream32(x, self.conv1.weight.dtype):
            # fp32 residual stream (UNet2DConditionModel.residual_fp32): both branches in the model dtype -- the same GroupNorm -> convolution
            # kernels from norm1's output on -- and ONE fp32 add onto the stream
            a = group_norm_act(self.norm1, x, silu=True)
            if _hip_conv3x3(a, self.conv1) and _hip_conv3x3(a[:, :1].expand(-1, cout, -1, -1), self.conv2, shape_only=True) \
                    and cout % 8 == 0 and cout // self.norm2.num_groups >= 8:
                from . import ops
                h = ops.conv3x3_nhwc(a, self.conv1.weight, self.conv1.bias, sample_bias=None if tb is None else tb.contiguous())
                a2 = group_norm_act(self.norm2, h, silu=True)
                B_, _, H_, W_ = a2.shape
                if a2.dtype == torch.float16 and ops.conv3x3_f32out_ok(B_, H_, W_, cout, cout):
                    # conv2 through the fp32-output epilogue (ed_conv3x3_nhwc_f32out: fp16 operands, fp32 bias / residual / result): the
                    # branch result is never rounded to 16 bits and the add onto the stream costs no pass of its own
                    res = x if sc is None else linear_(x.to(h.dtype).permute(0, 2, 3, 1), sc.weight.reshape(cout, -1), sc.bias).permute(0, 3, 1, 2).float()
                    res = res.contiguous(memory_format=torch.channels_last)
                    return ops.conv3x3_f32out(a2, self.conv2.weight, _f32_bias(self.conv2), res, 1.0)
                h = ops.conv3x3_nhwc(a2, self.conv2.weight, self.conv2.bias)
            else:
                h = self.conv1(a)
                h = self.conv2(group_norm_act(self.norm2, h, silu=True, chan_bias=tb))
            if sc is None:
                return x + h
            xs = x.to(h.dtype)
            res = (linear_(xs.permute(0, 2, 3, 1), sc.weight.reshape(cout, -1), sc.bias).permute(0, 3, 1, 2)
                   if xs.is_contiguous(memory_format=torch.channels_last) and not xs.is_contiguous() else sc(xs))
            return res.float() + h
        cl = _fusable_nhwc(x) and cout % 8 == 0 and cout // self.norm2.num_groups >= 8
        # both norms must take the channels-last HIP kernel (its torch fallback hands back an NCHW tensor the convolution
        # wrapper rejects) and BOTH convolutions must be shapes the kernel takes: conv2's Cin is cout
        if (cl and x.shape[1] // self.norm1.num_groups >= 8 and _hip_conv3x3(x, self.conv1)
                and _hip_conv3x3(x[:, :1].expand(-1, cout, -1, -1), self.conv2, shape_only=True)):
            # both convolutions as ed_conv3x3_nhwc: conv1's epilogue adds its bias and the time embedding, conv2's its bias and
            # the block's residual -- the two broadcast adds and the closing add never exist as separate passes
            from . import ops
            h = ops.conv3x3_nhwc(group_norm_act(self.norm1, x, silu=True), self.conv1.weight, self.conv1.bias,
                                 sample_bias=None if tb is None else tb.contiguous())
            a = group_norm_act(self.norm2, h, silu=True)
            res = x if sc is None else linear_(x.permute(0, 2, 3, 1), sc.weight.reshape(cout, -1), sc.bias).permute(0, 3, 1, 2)
            return ops.conv3x3_nhwc(a, self.conv2.weight, self.conv2.bias, residual=res)
        if FUSED_CONV_BIAS and FUSED_TEMB_ADD and (cl or (_fusable(x) and (x.shape[2] * x.shape[3]) % 8 == 0)):
            # MIOpen adds a convolution's bias with a separate broadcast kernel: run the convolutions bias-free and fold
            # conv1's bias (+ temb) into norm2's passes, conv2's and the shortcut's into the closing residual add
            from . import ops
            h = F.conv2d(group_norm_act(self.norm1, x, silu=True), self.conv1.weight, None, padding=1)
            a = group_norm_act(self.norm2, h, silu=True, chan_bias=tb, conv_bias=self.conv1.bias)
            h = F.conv2d(a, self.conv2.weight, None, padding=1)
            if sc is None:
                return ops.bias_residual_add(h, self.conv2.bias, x)
            if cl:  # channels-last: the 1x1 shortcut is a plain GEMM over the token view, no layout change
                w = sc.weight.reshape(cout, -1)
                res = F.linear(x.permute(0, 2, 3, 1), w).permute(0, 3, 1, 2)
            else:
                res = F.conv2d(x, sc.weight, None)
            return ops.bias_residual_add(h, self.conv2.bias, res, sc.bias)
        h = self.conv1(group_norm_act(self.norm1, x, silu=True))
        a = group_norm_act(self.norm2, h, silu=True, chan_bias=tb)  # GroupNorm(h + temb) (+SiLU)
        if sc is not None and SHORTCUT_AS_GEMM and _fusable(x):
            # out = W_sc x + (conv2(a) + b_conv2 + b_sc): the shortcut bias rides on conv2's bias, the residual sum is
            # the GEMM's C operand
            N, C, H, W = x.shape
            h = F.conv2d(a, self.conv2.weight, self.conv2.bias + sc.bias, padding=1)
            cout = h.shape[1]
            w = sc.weight.view(1, cout, C).expand(N, -1, -1)
            return torch.baddbmm(h.view(N, cout, H * W), w, x.view(N, C, H * W)).view(N, cout, H, W)
        h = self.conv2(a)
        return (x if sc is None else sc(x)) + h


def _resnet_forward_cat(self, x, skip, temb=None):
    """``self(torch.cat([x, skip], 1), temb)`` -- the up blocks' call -- without writing the concatenation (round 6): norm1 reads the two
    channels-last sources in place (ed_groupnorm_nhwc_cat, bit-identical to the kernel on the materialised cat) and the 1x1 shortcut is
    split along K, W_sc [x | skip] = W_1 x + W_2 skip: the second product accumulates onto the first (ed_linear's residual epilogue or
    the library's beta = 1), which rounds the first partial sum to 16 bits once -- the only arithmetic difference from the cat path.
    Everything else is ``forward``'s channels-last HIP path; any shape / layout it does not take falls back to the concatenation."""
    sc = self.conv_shortcut
    if FUSED_SKIP_CAT and FUSED_KERNELS and sc is not None and _fusable_nhwc(x) and _fusable_nhwc(skip) and x.dtype == skip.dtype:
        from . import ops
        C1, C = x.shape[1], x.shape[1] + skip.shape[1]
        cout = self.conv1.out_channels
        like = x[:, :1]
        if (ops.groupnorm_nhwc_cat_ok(x, skip, self.norm1.num_groups) and cout % 8 == 0 and cout // self.norm2.num_groups >= 8
                and self.conv1.weight.shape[1] == C and sc.weight.dtype == x.dtype
                and _hip_conv3x3(like.expand(-1, C, -1, -1), self.conv1, shape_only=True)
                and _hip_conv3x3(like.expand(-1, cout, -1, -1), self.conv2, shape_only=True)):
            tb = self.time_emb_proj(F.silu(temb)) if self.time_emb_proj is not None else None
            a = ops.groupnorm_nhwc_cat(x, skip, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps, silu=True)
            h = ops.conv3x3_nhwc(a, self.conv1.weight, self.conv1.bias, sample_bias=None if tb is None else tb.contiguous())
            a = group_norm_act(self.norm2, h, silu=True)
            w1, w2 = _split_shortcut(sc, C1)
            xt, st = x.permute(0, 2, 3, 1), skip.permute(0, 2, 3, 1)          # contiguous [N, H, W, C] views of channels-last memory
            y = linear_(xt, w1, sc.bias)
            M = st.numel() // st.shape[-1]
            if HIP_LINEAR and ops.linear_wins(M, st.shape[-1], cout):
                y = ops.linear(st, w2, None, residual=y)
            else:
                y = torch.addmm(y.reshape(M, cout), st.reshape(M, -1), w2.t()).view(y.shape)
            return ops.conv3x3_nhwc(a, self.conv2.weight, self.conv2.bias, residual=y.permute(0, 3, 1, 2))
    return self(torch.cat([x, skip], dim=1), temb)


def _split_shortcut(sc, C1):
    """the 1x1 shortcut's weight [cout, C1 + C2, 1, 1] as two contiguous matrices [cout, C1], [cout, C2]; rebuilt when the weight changes"""
    w = sc.weight
    key = (id(w), w.data_ptr(), w._version, C1)
    hit = sc.__dict__.get("_ksplit")
    if hit is None or hit[0] != key:
        w2d = w.detach().reshape(w.shape[0], -1)
        hit = (key, w2d[:, :C1].contiguous(), w2d[:, C1:].contiguous())
        sc.__dict__["_ksplit"] = hit
    return hit[1], hit[2]


ResnetBlock2D.forward_cat = _resnet_forward_cat


def _resnet_forward_split(self, x):
    """The fp32 VAE block with both 3x3 convolutions on the MFMA pipe (VAE_SPLIT_CONV): GroupNorm + SiLU write the split fp16 operand,
    the convolution's fp32 epilogue adds the bias and (conv2) the block's residual.  Batches whose operand would exceed the kernel's
    32-bit offsets are processed in slices (GroupNorm is per sample)."""
    from . import ops
    cl = torch.channels_last
    if not x.is_contiguous(memory_format=cl):
        x = x.contiguous(memory_format=cl)
    B, cin, H, W = x.shape
    cout = self.conv1.out_channels
    per = H * W * 3 * max(cin, cout) * 2
    nb = max(1, (2 ** 31 - 16) // per)
    if B > nb:
        return torch.cat([_resnet_forward_split(self, x[i:i + nb]) for i in range(0, B, nb)]).contiguous(memory_format=cl)
    w1, s1 = _split_weight(self.conv1)
    w2, s2 = _split_weight(self.conv2)
    a = ops.groupnorm_nhwc_f32(x, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps, silu=True, split=True)
    h = ops.conv3x3_f32out(a, w1, self.conv1.bias, None, s1)
    a = ops.groupnorm_nhwc_f32(h, self.norm2.weight, self.norm2.bias, self.norm2.num_groups, self.norm2.eps, silu=True, split=True)
    sc = self.conv_shortcut
    if sc is None:
        res = x
    else:   # 1x1 convolution = an fp32 GEMM over the NHWC view
        res = F.linear(x.permute(0, 2, 3, 1), sc.weight.reshape(cout, cin), sc.bias).permute(0, 3, 1, 2)
    return ops.conv3x3_f32out(a, w2, self.conv2.bias, res, s2)


ResnetBlock2D._forward_split = _resnet_forward_split


class Attention(nn.Module):
    def __init__(self, dim, heads, head_dim, cross_dim=None, bias=False):
        super().__init__()
        inner = heads * head_dim
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=bias)
        self.to_k = nn.Linear(cross_dim or dim, inner, bias=bias)
        self.to_v = nn.Linear(cross_dim or dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Dropout(0.0)])

    def _fused_weight(self, names, q_scale=None):
        """cat of the projection weights (bias-free in every SD / SDXL attention), rebuilt when a weight changes.
        ``q_scale``: factor folded into the first (query) weight -- softmax scale * log2 e for the exponent-domain attention
        kernel (ops.flash_prescale): the product is formed in fp32 and rounded once, so the scaled 16-bit weight is as close
        to c * W as the stored one is to W."""
        mods = [getattr(self, n) for n in names]
        key = tuple((m.weight.data_ptr(), m.weight._version, m.weight.dtype, str(m.weight.device)) for m in mods)
        cache = self.__dict__.setdefault("_fused", {})
        slot = (names, q_scale)
        if cache.get(slot, (None, None))[0] != key:
            ws = [m.weight.detach() for m in mods]
            if q_scale is not None:
                ws[0] = (ws[0].float() * q_scale).to(ws[0].dtype)
            cache[slot] = (key, torch.cat(ws, dim=0).contiguous())
        return cache[slot][1]

    def _out_proj(self, o):
        """to_out[0], with the block's residual in its epilogue when one is pending (forward(residual=...)) and ed_linear runs this shape"""
        res = self.__dict__.get("_res")
        if res is not None:
            y, fused = linear_res_(o, self.to_out[0].weight, self.to_out[0].bias, res)
            if fused:
                self.__dict__.pop("_res", None)
            return y
        return linear_(o, self.to_out[0].weight, self.to_out[0].bias)

    def project_kv(self, context, out=None):
        """k, v of a cross-attention as ONE [B, Nk, 2 inner] tensor (fused weight).  They depend on the text rows only -- not on
        the latent, not on the timestep -- so a caller may compute them once per image and hand them to ``forward`` (``kv``)
        instead of re-projecting the same 77 tokens in every one of the ~100 forwards of an image (UNet.cross_attention_kv)."""
        w = self._fused_weight(("to_k", "to_v"))
        if out is None:   # the same library call for the first projection and every refresh: one hipBLASLt solution, same bits
            out = torch.empty(context.shape[:-1] + (w.shape[0],), dtype=context.dtype, device=context.device)
        torch.matmul(context, w.t(), out=out)
        return out

    def forward(self, x, context=None, kv=None, residual=None):
        """``residual`` given: -> (out, fused) -- out = residual + attention(...) when the output projection took the add (fused), else the
        plain attention output and the caller adds."""
        if residual is not None:
            self.__dict__["_res"] = residual
            try:
                out = self.forward(x, context, kv)
            finally:
                res_state = self.__dict__.pop("_res", None)
            return out, res_state is None          # _out_proj consumed the residual
        B, N, _ = x.shape
        inner = self.to_q.out_features
        if (FLASH_ATTENTION and _fusable(x) and inner % self.heads == 0 and inner // self.heads in (40, 64, 80, 160)
                and self.to_q.bias is None
                and (context is None or (_fusable(context) and context.dtype == x.dtype))):
            from . import ops
            if context is None:
                # exponent-domain kernel: softmax scale * log2 e lives in the query weights (None: natural-domain q)
                c = ops.flash_prescale(N, N, 3 * inner if FUSED_QKV else inner) if inner == self.heads * 64 else None
                if FUSED_QKV:
                    qkv = linear_(x, self._fused_weight(("to_q", "to_k", "to_v"), c))
                    q, k, v = qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:]
                else:
                    q = linear_(x, self.to_q.weight if c is None else self._fused_weight(("to_q",), c))
                    k, v = self.to_k(x), self.to_v(x)
                o = ops.flash_attention(q, k, v, self.heads, prescaled=c is not None)
                return self._out_proj(o)
            else:
                q = linear_(x, self.to_q.weight)
                if kv is not None or FUSED_QKV:
                    if kv is None:
                        kv = self.project_kv(context)
                    k, v = kv[..., :inner], kv[..., inner:]
                else:
                    k, v = self.to_k(context), self.to_v(context)
            return self._out_proj(ops.flash_attention(q, k, v, self.heads))
        ctx = x if context is None else context
        q = self.to_q(x).view(B, N, self.heads, -1).transpose(1, 2)
        if kv is not None:
            k, v = (t.reshape(B, ctx.shape[1], self.heads, -1).transpose(1, 2) for t in (kv[..., :inner], kv[..., inner:]))
        else:
            k = self.to_k(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
            v = self.to_v(ctx).view(B, ctx.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, -1))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        if HIP_GEGLU_GEMM and FUSED_KERNELS and _fusable(x) and self.proj.weight.dtype == x.dtype:
            from . import ops
            if ops.geglu_gemm_wins(x.numel() // x.shape[-1], x.shape[-1], self.proj.out_features // 2):
                return ops.geglu_gemm(x, self.proj.weight, self.proj.bias)
        y = self.proj(x)
        if _fusable(y) and (y.shape[-1] // 2) % 8 == 0:
            from . import ops
            return ops.geglu(y, y.shape[-1] // 2)
        h, gate = y.chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x, residual=None):
        if residual is not None:     # -> (out, fused): out = residual + ff(x) when the second projection took the add
            return linear_res_(self.net[0](x), self.net[2].weight, self.net[2].bias, residual)
        return linear_(self.net[0](x), self.net[2].weight, self.net[2].bias)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, head_dim, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context, pending=None, kv=None):
        """-> (ff_out, x): the block's output is ``ff_out + x``; the caller either hands the pair to the next block
        (whose norm1 then runs fused with that add) or sums it -- or (None, x) when the feed-forward's second projection already
        added (FUSED_RESIDUAL_LINEAR: wherever ed_linear runs a residual-producing projection the add rides in its epilogue and the
        next LayerNorm is a plain one).  ``pending``: the previous block's ff_out.  ``kv``: dict
        {id(attention module): precomputed k|v} (UNet.cross_attention_kv) or None."""
        if pending is not None:
            x, h = add_layer_norm(self.norm1, pending, x)                 # x = prev_ff + x ; norm1(x)
        else:
            h = layer_norm(self.norm1, x)
        a, fused = self.attn1(h, residual=x)                              # x = attn1(norm1(x)) + x ; norm2(x)
        x, h = (a, layer_norm(self.norm2, a)) if fused else add_layer_norm(self.norm2, a, x)
        a, fused = self.attn2(h, context, None if kv is None else kv.get(id(self.attn2)), residual=x)   # x = attn2(norm2(x)) + x ; norm3(x)
        x, h = (a, layer_norm(self.norm3, a)) if fused else add_layer_norm(self.norm3, a, x)
        a, fused = self.ff(h, residual=x)
        return (None, a) if fused else (a, x)                              # (None, x): the add already happened in ff's projection


class Transformer2DModel(nn.Module):
    def __init__(self, ch, heads, depth, cross_dim, linear_proj):
        super().__init__()
        self.linear_proj = linear_proj
        self.norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, ch) if linear_proj else nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, ch // heads, cross_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(ch, ch) if linear_proj else nn.Conv2d(ch, ch, 1)

    def forward(self, x, context, kv=None):
        B, C, H, W = x.shape
        if self.linear_proj:
            h = linear_(group_norm_act(self.norm, x, tokens=True), self.proj_in.weight, self.proj_in.bias)
        else:
            h = self.proj_in(group_norm_act(self.norm, x)).permute(0, 2, 3, 1).reshape(B, H * W, C)
        s32 = _stream32(x, self.proj_out.weight.dtype)
        if s32:
            h = h.float()      # the blocks' own residual stream in fp32 as well (their LayerNorms hand the model dtype to the branches)
        pend = None
        for blk in self.transformer_blocks:
            pend, h = blk(h, context, pend, kv)
        if pend is not None:
            h = pend + h
        if s32:
            h = h.to(self.proj_out.weight.dtype)
        if self.linear_proj:
            if (FUSED_PROJ_OUT_ADD and HIP_LINEAR and FUSED_KERNELS and _fusable_nhwc(x) and _fusable(h) and self.proj_out.weight.dtype == x.dtype == h.dtype
                    and self.proj_out.weight.is_contiguous()):
                from . import ops
                if ops.linear_wins(B * H * W, C, C):
                    # channels-last: x's memory IS the token layout, so the block's closing `x + proj_out(h)` rides in ed_linear's residual
                    # epilogue (one rounding of bias + product + residual instead of two) -- no separate add pass
                    y = ops.linear(h, self.proj_out.weight, self.proj_out.bias, residual=x.permute(0, 2, 3, 1).reshape(B, H * W, C))
                    return y.view(B, H, W, C).permute(0, 3, 1, 2)
            h = linear_(h, self.proj_out.weight, self.proj_out.bias)
            if (FUSED_TOKENS_ADD and _fusable(x) and _fusable(h) and C % 64 == 0 and (H * W) % 64 == 0):
                from . import ops
                return ops.tokens_add_nchw(x, h)  # x + h^T in one pass through an LDS tile
            h = h.view(B, H, W, C).permute(0, 3, 1, 2)
        else:
            h = self.proj_out(h.view(B, H, W, C).permute(0, 3, 1, 2))
        return x + h  # x first: the sum keeps x's NCHW layout (a permuted first operand would make it channels-last)


class Downsample2D(nn.Module):
    def __init__(self, ch, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0 and VAE_SPLIT_CONV and VAE_SPLIT_DOWNSAMPLE and FUSED_KERNELS and x.is_cuda and x.dtype == torch.float32 \
                and x.dim() == 4 and self.conv.weight.dtype == torch.float32:
            # the VAE encoder's downsampler on the MFMA pipe (round 6): the raw stream is split like the decoder's upsampler input (per-tensor
            # power-of-two scale from its absolute maximum), the pad-(0, 1, 0, 1) stride-2 convolution is template value CONV = 4 of the
            # split-operand main loop; channels-last in and out -- no NCHW copies around the library call
            from . import ops
            B, C, H, W = x.shape
            if C % 64 == 0 and ops.conv3x3_f32out_s2_ok(1, H, W, 3 * C, self.conv.weight.shape[0]):
                w, sc = _split_weight(self.conv)
                x = x.contiguous(memory_format=torch.channels_last)
                nb = max(1, (2 ** 31 - 16) // (H * W * 3 * C * 2))
                outs = []
                for i in range(0, B, nb):
                    xs = x[i:i + nb]
                    am = ops.absmax_f32(xs)
                    outs.append(ops.conv3x3_f32out_s2(ops.split_f32(xs, absmax=am), w, self.conv.bias, sc, act_absmax=am))
                return outs[0] if len(outs) == 1 else torch.cat(outs).contiguous(memory_format=torch.channels_last)
        if self.padding == 0:  # VAE encoder: asymmetric pad (diffusers Downsample2D)
            x = F.pad(_to_nchw(x), (0, 1, 0, 1))
        if _stream32(x, self.conv.weight.dtype):   # fp32 residual stream: the convolution in the model dtype, its result widened (exact)
            return self.forward(x.to(self.conv.weight.dtype)).float()
        if HIP_DOWNSAMPLE_CONV and HIP_CONV3X3 and self.padding == 1 and _fusable_nhwc(x) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            w = self.conv.weight
            if w.dtype == x.dtype and w.is_contiguous(memory_format=torch.channels_last):
                from . import ops
                B, C, H, W = x.shape
                if ops.conv3x3_s2_wins(B, H // 2, W // 2, C, w.shape[0]):
                    return ops.conv3x3_nhwc_s2(x, w, self.conv.bias)       # the stride-2 convolution on the MFMA main loop
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch, vae=False):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)
        self.vae = vae   # the VAE decoder's upsampler (the split-operand path is the VAE's; an fp32 UNet stays plain torch: it is the
                         # tolerance reference of bench.py's fp32 leg and of the parity tests)

    def _split_ok(self, x):
        """fp32 VAE, channels % 64 == 0, and ONE sample's upsampled split operand within the kernel's 32-bit offsets (the 1024 x 2048
        decode's last upsampler -- [1, 768, 1024, 2048] fp16 = 3.2 GB -- is not: it stays with the library)"""
        if not (self.vae and VAE_SPLIT_CONV and FUSED_KERNELS and x.is_cuda and x.dim() == 4 and self.conv.weight.dtype == torch.float32):
            return False
        B, C, H, W = x.shape
        return C % 64 == 0 and 4 * H * W * 3 * C * 2 < 2 ** 31 - 16     # per sample; larger batches are processed in slices

    def forward(self, x):
        if _stream32(x, self.conv.weight.dtype):   # fp32 residual stream of a 16-bit UNet
            return self.forward(x.to(self.conv.weight.dtype)).float()
        if x.dtype == torch.float32:
            if self._split_ok(x):
                # the VAE decoder's upsampler on the MFMA pipe (VAE_SPLIT_CONV): the raw stream is split (scaled by a per-tensor power of
                # two taken from its absolute maximum on the device: the real decoder stream leaves fp16's range) and upsampled in one pass, the convolution is the same split-operand main loop as the ResnetBlocks'; channels-last in and out
                from . import ops
                w, sc = _split_weight(self.conv)
                cl = torch.channels_last
                x = x.contiguous(memory_format=cl)
                B, C, H, W = x.shape
                nb = max(1, (2 ** 31 - 16) // (4 * H * W * 3 * C * 2))
                outs = []
                for i in range(0, B, nb):
                    xs = x[i:i + nb]
                    am = ops.absmax_f32(xs)     # per-slice power-of-two scale: the split is exact over the whole fp32 range
                    outs.append(ops.conv3x3_f32out(ops.split_f32(xs, upsample2x=True, absmax=am), w, self.conv.bias, None, sc, act_absmax=am))
                return outs[0] if len(outs) == 1 else torch.cat(outs).contiguous(memory_format=cl)
            x = _to_nchw(x)  # NCHW for MIOpen's fp32 solvers (the UNet's 16-bit activations stay channels-last)
        if FUSED_UPSAMPLE_CONV and HIP_CONV3X3 and _fusable_nhwc(x):
            w = self.conv.weight
            if w.dtype == x.dtype and w.is_contiguous(memory_format=torch.channels_last):
                from . import ops
                B, C, H, W = x.shape
                if ops.conv3x3_up2x_wins(B, 2 * H, 2 * W, C, w.shape[0]):
                    return ops.conv3x3_nhwc_up2x(x, w, self.conv.bias)   # the upsampled tensor (4x the source) is never written
        up = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if _hip_conv3x3(up, self.conv):
            from . import ops
            return ops.conv3x3_nhwc(up, self.conv.weight, self.conv.bias)
        return self.conv(up)


# ---------------------------------------------------------------------------------------------------
# UNet2DConditionModel
# ---------------------------------------------------------------------------------------------------
UNET_CONFIGS = {
    "sd15": dict(sample_size=64, in_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 attn=(True, True, True, False), transformer_depth=(1, 1, 1, 1), heads=(8, 8, 8, 8),
                 cross_attention_dim=768, use_linear_projection=False, addition_time_embed_dim=None,
                 pooled_projection_dim=None),
    "sd2": dict(sample_size=64, in_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                attn=(True, True, True, False), transformer_depth=(1, 1, 1, 1), heads=(5, 10, 20, 20),
                cross_attention_dim=1024, use_linear_projection=True, addition_time_embed_dim=None,
                pooled_projection_dim=None),
    "sdxl": dict(sample_size=128, in_channels=4, block_out_channels=(320, 640, 1280), layers_per_block=2,
                 attn=(False, True, True), transformer_depth=(1, 2, 10), heads=(5, 10, 20),
                 cross_attention_dim=2048, use_linear_projection=True, addition_time_embed_dim=256,
                 pooled_projection_dim=1280),
}


# Reduced-width variants of the same architectures (same block structure, attention placement, projection kind, head
# dim 64, added-condition path): small enough that the fp32 CPU oracle runs them in seconds, so the parity tests and
# bench.py's parity leg can put the REAL module code (not a stand-in) inside an oracle-checked denoising loop.
SMALL_UNET_CONFIGS = {
    "sd15": dict(sample_size=64, in_channels=4, block_out_channels=(64, 128, 256, 256), layers_per_block=2,
                 attn=(True, True, True, False), transformer_depth=(1, 1, 1, 1), heads=(1, 2, 4, 4),
                 cross_attention_dim=64, use_linear_projection=False, addition_time_embed_dim=None,
                 pooled_projection_dim=None),
    "sdxl": dict(sample_size=128, in_channels=4, block_out_channels=(64, 128, 256), layers_per_block=2,
                 attn=(False, True, True), transformer_depth=(1, 1, 2), heads=(1, 2, 4),
                 cross_attention_dim=64, use_linear_projection=True, addition_time_embed_dim=16,
                 pooled_projection_dim=32),
}
SMALL_VAE_CHANNELS = (32, 64, 64, 64)


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, layers, attn, heads, depth, cross, linear, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb) for i in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, depth, cross, linear) for _ in range(layers)]) if attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if downsample else None

    def forward(self, x, temb, ctx, kv=None):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx, kv)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class _MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, depth, cross, linear):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb), ResnetBlock2D(ch, ch, temb)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, heads, depth, cross, linear)])

    def forward(self, x, temb, ctx, kv=None):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, temb), ctx, kv), temb)


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, layers, attn, heads, depth, cross, linear, upsample):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rs.append(ResnetBlock2D((cprev if i == 0 else cout) + skip, cout, temb))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, depth, cross, linear) for _ in range(layers)]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if upsample else None

    def forward(self, x, skips, temb, ctx, kv=None):
        for i, r in enumerate(self.resnets):
            x = r.forward_cat(x, skips.pop(), temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx, kv)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNet2DConditionModel(nn.Module):
    def __init__(self, sample_size, in_channels, block_out_channels, layers_per_block, attn, transformer_depth, heads,
                 cross_attention_dim, use_linear_projection, addition_time_embed_dim, pooled_projection_dim,
                 out_channels=4):
        super().__init__()
        boc = tuple(block_out_channels)
        temb = boc[0] * 4
        self.config = SimpleNamespace(sample_size=sample_size, in_channels=in_channels, block_out_channels=boc,
                                      cross_attention_dim=cross_attention_dim,
                                      addition_time_embed_dim=addition_time_embed_dim,
                                      pooled_projection_dim=pooled_projection_dim)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        if addition_time_embed_dim:
            self.add_embedding = TimestepEmbedding(addition_time_embed_dim * 6 + pooled_projection_dim, temb)
        n = len(boc)
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i in range(n):
            self.down_blocks.append(_DownBlock(ch, boc[i], temb, layers_per_block, attn[i], heads[i], transformer_depth[i],
                                               cross_attention_dim, use_linear_projection, downsample=(i < n - 1)))
            ch = boc[i]
        self.mid_block = _MidBlock(boc[-1], temb, heads[-1], transformer_depth[-1], cross_attention_dim, use_linear_projection)
        self.up_blocks = nn.ModuleList()
        rev, rattn, rheads, rdepth = boc[::-1], attn[::-1], heads[::-1], transformer_depth[::-1]
        prev = rev[0]
        for i in range(n):
            cout, cin = rev[i], rev[min(i + 1, n - 1)]
            self.up_blocks.append(_UpBlock(cin, cout, prev, temb, layers_per_block + 1, rattn[i], rheads[i], rdepth[i],
                                           cross_attention_dim, use_linear_projection, upsample=(i < n - 1)))
            prev = cout
        self.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def embed(self, sample, timestep, added_cond_kwargs):
        B = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(B)
        emb = self.time_embedding(timestep_embedding(t, self.config.block_out_channels[0]).to(sample.dtype))
        if self.config.addition_time_embed_dim:
            ids = added_cond_kwargs["time_ids"]
            tid = timestep_embedding(ids.reshape(-1).float(), self.config.addition_time_embed_dim).reshape(B, -1)
            add = torch.cat([added_cond_kwargs["text_embeds"].to(sample.dtype), tid.to(sample.dtype)], dim=-1)
            emb = emb + self.add_embedding(add)
        return emb

    def cross_attention_kv(self, encoder_hidden_states, into=None):
        """{id(attention module): k|v [B, 77, 2 inner]} for every cross-attention of the model: functions of the text rows only,
        so the pipeline computes them once per image (per hipGraph batch shape) instead of once per forward -- 70 projections
        of 77 tokens per SDXL forward, ~100 forwards per image.  ``into``: a dict from an earlier call whose tensors are
        overwritten in place (a captured hipGraph reads them at fixed addresses)."""
        ctx = encoder_hidden_states.to(self.dtype)
        out = {} if into is None else into
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock):
                out[id(m.attn2)] = m.attn2.project_kv(ctx, None if into is None else into[id(m.attn2)])
        return out

    def forward(self, sample, timestep, encoder_hidden_states=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, cross_kv=None, **_):
        x = sample.to(self.dtype)
        if CHANNELS_LAST and x.is_cuda and x.dtype != torch.float32:
            x = x.contiguous(memory_format=torch.channels_last)
        ctx = encoder_hidden_states.to(self.dtype)
        emb = self.embed(x, timestep, added_cond_kwargs)
        x = self.conv_in(x)
        if getattr(self, "residual_fp32", False) and x.dtype != torch.float32:
            x = x.float()      # from here to conv_norm_out the residual stream is fp32 (see _stream32)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx, cross_kv)
            skips.extend(outs)
        if down_block_additional_residuals is not None:
            skips = [s + r for s, r in zip(skips, down_block_additional_residuals)]
        x = self.mid_block(x, emb, ctx, cross_kv)
        if mid_block_additional_residual is not None:
            x = x + mid_block_additional_residual
        for blk in self.up_blocks:
            x = blk(x, skips, emb, ctx, cross_kv)
        x = self.conv_out(group_norm_act(self.conv_norm_out, x, silu=True))
        return ModelOutput(sample=x)


# ---------------------------------------------------------------------------------------------------
# ControlNet (diffusers ControlNetModel): the UNet encoder + zero-convs, conditioned on a pixel-space image
# ---------------------------------------------------------------------------------------------------
class _CondEmbedding(nn.Module):
    def __init__(self, out_ch, chans=(16, 32, 96, 256)):
        super().__init__()
        self.conv_in = nn.Conv2d(3, chans[0], 3, padding=1)
        blocks = []
        for a, b in zip(chans[:-1], chans[1:]):
            blocks += [nn.Conv2d(a, a, 3, padding=1), nn.Conv2d(a, b, 3, padding=1, stride=2)]
        self.blocks = nn.ModuleList(blocks)
        self.conv_out = nn.Conv2d(chans[-1], out_ch, 3, padding=1)

    def forward(self, c):
        h = F.silu(self.conv_in(c))
        for b in self.blocks:
            h = F.silu(b(h))
        return self.conv_out(h)


class ControlNetModel(nn.Module):
    def __init__(self, unet_cfg):
        super().__init__()
        c = dict(unet_cfg)
        boc = tuple(c["block_out_channels"])
        temb = boc[0] * 4
        self.config = SimpleNamespace(**{k: c[k] for k in ("sample_size", "in_channels", "cross_attention_dim",
                                                           "addition_time_embed_dim", "pooled_projection_dim")},
                                      block_out_channels=boc)
        self.conv_in = nn.Conv2d(c["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        if c["addition_time_embed_dim"]:
            self.add_embedding = TimestepEmbedding(c["addition_time_embed_dim"] * 6 + c["pooled_projection_dim"], temb)
        self.controlnet_cond_embedding = _CondEmbedding(boc[0])
        n = len(boc)
        self.down_blocks = nn.ModuleList()
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(boc[0], boc[0], 1)])
        ch = boc[0]
        for i in range(n):
            self.down_blocks.append(_DownBlock(ch, boc[i], temb, c["layers_per_block"], c["attn"][i], c["heads"][i],
                                               c["transformer_depth"][i], c["cross_attention_dim"],
                                               c["use_linear_projection"], downsample=(i < n - 1)))
            ch = boc[i]
            for _ in range(c["layers_per_block"] + (1 if i < n - 1 else 0)):
                self.controlnet_down_blocks.append(nn.Conv2d(ch, ch, 1))
        self.mid_block = _MidBlock(boc[-1], temb, c["heads"][-1], c["transformer_depth"][-1], c["cross_attention_dim"],
                                   c["use_linear_projection"])
        self.controlnet_mid_block = nn.Conv2d(boc[-1], boc[-1], 1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    embed = UNet2DConditionModel.embed
    cross_attention_kv = UNet2DConditionModel.cross_attention_kv

    def forward(self, sample, timestep, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=1.0,
                guess_mode=False, return_dict=False, added_cond_kwargs=None, cross_kv=None, **_):
        x = sample.to(self.dtype)
        cond = controlnet_cond.to(self.dtype)
        if CHANNELS_LAST and x.is_cuda and x.dtype != torch.float32:
            x = x.contiguous(memory_format=torch.channels_last)
            cond = cond.contiguous(memory_format=torch.channels_last)
        ctx = encoder_hidden_states.to(self.dtype)
        emb = self.embed(x, timestep, added_cond_kwargs)
        x = self.conv_in(x) + self.controlnet_cond_embedding(cond)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx, cross_kv)
            skips.extend(outs)
        x = self.mid_block(x, emb, ctx, cross_kv)
        down = [conv(s) * conditioning_scale for conv, s in zip(self.controlnet_down_blocks, skips)]
        mid = self.controlnet_mid_block(x) * conditioning_scale
        return down, mid


# ---------------------------------------------------------------------------------------------------
# AutoencoderKL
# ---------------------------------------------------------------------------------------------------
class _VaeAttention(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.group_norm = nn.GroupNorm(32, ch, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = group_norm_act(self.group_norm, x)
        cl = not h.is_contiguous() and h.is_contiguous(memory_format=torch.channels_last)   # channels-last VAE (VAE_SPLIT_CONV)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C) if cl else h.view(B, C, H * W).transpose(1, 2)
        if VAE_HIP_ATTENTION and h.is_cuda and h.dtype == torch.float32 and (H * W) % 4 == 0:
            # two fp32 library GEMMs around ed_softmax_rows: no AOTriton (Triton) kernel on the path
            from . import ops
            o = ops.vae_attention(self.to_q(h), self.to_k(h), self.to_v(h))
        else:
            q, k, v = (f(h).unsqueeze(1) for f in (self.to_q, self.to_k, self.to_v))
            o = F.scaled_dot_product_attention(q, k, v).squeeze(1)
        a = self.to_out[0](o)
        a = a.view(B, H, W, C).permute(0, 3, 1, 2) if cl else a.transpose(1, 2).reshape(B, C, H, W)   # (cl: a view, no copy)
        return x + a if VAE_NCHW_RESIDUAL else a + x


class _VaeMid(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, eps=1e-6), ResnetBlock2D(ch, ch, None, eps=1e-6)])
        self.attentions = nn.ModuleList([_VaeAttention(ch)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](_to_nchw(self.resnets[0](x))))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, layers, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.downsamplers[0](x) if self.downsamplers is not None else x


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, layers, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout, vae=True)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if self.upsamplers is not None else x


class _Encoder(nn.Module):
    def __init__(self, boc, latent_channels, layers=2):
        super().__init__()
        self.conv_in = nn.Conv2d(3, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, c in enumerate(boc):
            self.down_blocks.append(_EncBlock(ch, c, layers, down=(i < len(boc) - 1)))
            ch = c
        self.mid_block = _VaeMid(ch)
        self.conv_norm_out = nn.GroupNorm(32, ch, eps=1e-6)
        self.conv_out = nn.Conv2d(ch, 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(group_norm_act(self.conv_norm_out, _to_nchw(self.mid_block(x)), silu=True))


class _Decoder(nn.Module):
    def __init__(self, boc, latent_channels, layers=2):
        super().__init__()
        rev = boc[::-1]
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _VaeMid(rev[0])
        self.up_blocks = nn.ModuleList()
        ch = rev[0]
        for i, c in enumerate(rev):
            self.up_blocks.append(_DecBlock(ch, c, layers + 1, up=(i < len(boc) - 1)))
            ch = c
        self.conv_norm_out = nn.GroupNorm(32, ch, eps=1e-6)
        self.conv_out = nn.Conv2d(ch, 3, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(group_norm_act(self.conv_norm_out, _to_nchw(x), silu=True))


class DiagonalGaussian:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise


class AutoencoderKL(nn.Module):
    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4, scaling_factor=0.18215,
                 force_upcast=False):
        super().__init__()
        self.config = SimpleNamespace(block_out_channels=tuple(block_out_channels), scaling_factor=scaling_factor,
                                      force_upcast=force_upcast, latent_channels=latent_channels)
        self.encoder = _Encoder(tuple(block_out_channels), latent_channels)
        self.decoder = _Decoder(tuple(block_out_channels), latent_channels)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    @property
    def device(self):
        return self.quant_conv.weight.device

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussian(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))


# ---------------------------------------------------------------------------------------------------
def family(sd_version):
    if sd_version.startswith("XL"):
        return "sdxl"
    if sd_version in ("2.0", "2.1"):
        return "sd2"
    if sd_version in ("1.4", "1.5"):
        return "sd15"
    raise ValueError(f"no built-in architecture for sd_version={sd_version!r}; pass unet=/vae= explicitly")


def _seeded_init(module, seed):
    """Deterministic synthetic weights (no checkpoints exist offline): PyTorch's default ``reset_parameters`` init under
    a private seed (measured: the random-init SDXL loop stays finite over 50 steps in bf16 without any rescaling).
    ``torch.random.fork_rng`` restores the CPU generator and the generator of the device being initialised."""
    devs = sorted({p.device.index for p in module.parameters() if p.is_cuda and p.device.index is not None})
    with torch.random.fork_rng(devices=devs):
        torch.manual_seed(seed)
        for m in module.modules():
            if hasattr(m, "reset_parameters"):
                m.reset_parameters()


def load_weights(module, path):
    """Load an HF-layout ``diffusion_pytorch_model.safetensors`` (same parameter names as diffusers)."""
    from safetensors.torch import load_file
    sd = load_file(path)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"{path}: {len(missing)} missing / {len(unexpected)} unexpected keys, e.g. {missing[:3]} {unexpected[:3]}")
    return module


# The 16-bit dtype of the UNet / ControlNet when the caller does not choose one.  fp16 is what the reference's own GPU path
# runs the UNet in (``torch.autocast('cuda')``, ED:1012), and measured on the MI355X (profiles/r3_precision.json) its loop
# drift against the fp32 reference path is 8x smaller than bf16's (5e-3 vs 4e-2 rel-L2 on the SDXL geometry: 10 vs 7
# mantissa bits at the same MFMA rate); it stays finite over 50 steps at full width with random-init weights.
DEFAULT_MODEL_DTYPE = torch.float16


def build_models(sd_version, device="cuda", dtype=None, weights=None, vae_dtype=torch.float32, controlnet=False, seed=0,
                 small=False):
    """(unet, vae[, controlnet]) for the reference's ``sd_version`` keys (ED:128-141).  UNet/ControlNet run in
    DEFAULT_MODEL_DTYPE (fp16) unless ``dtype`` says otherwise; the VAE stays fp32 like the reference (its decode runs
    outside autocast, ED:1080-1121, and the encoder is explicitly kept out of autocast, ED:328).  ``small=True`` builds the
    reduced-width variants (parity checks)."""
    fam = family(sd_version)
    cfg = (SMALL_UNET_CONFIGS if small else UNET_CONFIGS)[fam]
    dtype = dtype or DEFAULT_MODEL_DTYPE
    with torch.device("meta"):
        unet = UNet2DConditionModel(**cfg)
        vae_kw = dict(block_out_channels=SMALL_VAE_CHANNELS) if small else {}
        vae = AutoencoderKL(scaling_factor=0.13025 if fam == "sdxl" else 0.18215, force_upcast=(fam == "sdxl"), **vae_kw)
        cn = ControlNetModel(cfg) if controlnet else None
    out = []
    for k, (m, dt, sub) in enumerate([(unet, dtype, "unet"), (vae, vae_dtype, "vae"), (cn, dtype, "controlnet")]):
        if m is None:
            continue
        m = m.to_empty(device=device)
        f = os.path.join(weights, sub, "diffusion_pytorch_model.safetensors") if weights else None
        if f and os.path.isfile(f):
            load_weights(m, f)
        else:
            _seeded_init(m, seed + k)
        m = m.to(dtype=dt).eval().requires_grad_(False)
        if CHANNELS_LAST and sub != "vae" and dt != torch.float32:  # 16-bit UNet / ControlNet; the fp32 VAE stays NCHW
            m = m.to(memory_format=torch.channels_last)
        out.append(m)
    return tuple(out)
