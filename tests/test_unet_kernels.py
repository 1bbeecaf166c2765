"""-m gpu: the fused HIP kernels inside the UNet (csrc/unet_kernels.hip) against the torch ops they replace.
fp reference = the same op sequence in torch on the same 16-bit tensors; bar: every element within 2 units in the
last place of the 16-bit type (the kernels round at the same points as torch; the residual is libm / FMA detail),
and >= 99 % of elements bit-identical."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ulp_report(got, want):
    eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[got.dtype]
    g, w = got.float(), want.float()
    tol = 2 * eps * w.abs().clamp_min(2.0 ** -14) * 2
    bad = ((g - w).abs() > tol).sum().item()
    same = (got == want).float().mean().item()
    return bad, same


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,I", [(4096, 2560), (2048, 5120), (77, 64)])
def test_geglu(dtype, M, I):
    from elasticdiffusion_official_amd import ops
    x = (torch.randn(M, 2 * I, device=DEV) * 2).to(dtype)
    h, gate = x.chunk(2, dim=-1)
    want = h * F.gelu(gate)
    got = ops.geglu(x, I)
    bad, same = ulp_report(got, want)
    assert bad == 0 and same > 0.99, (bad, same)
    got3 = ops.geglu(x.view(M // 1, 1, 2 * I), I)
    assert got3.shape == (M, 1, I) and torch.equal(got3.view(M, I), got)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W", [(3, 320, 32, 32), (2, 640, 16, 16), (2, 1280, 8, 8), (1, 960, 64, 64), (2, 32, 4, 2)])
@pytest.mark.parametrize("silu,tokens", [(True, False), (False, False), (False, True), (True, True)])
def test_groupnorm(dtype, N, C, H, W, silu, tokens):
    from elasticdiffusion_official_amd import ops
    x = (torch.randn(N, C, H, W, device=DEV) * 1.7 + 0.3).to(dtype)
    w = (1 + 0.2 * torch.randn(C, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(C, device=DEV)).to(dtype)
    if tokens and (C // 32) % 4:
        pytest.skip("token layout needs channels-per-group % 4 == 0")
    want = F.group_norm(x, 32, w, b, 1e-5)
    if silu:
        want = F.silu(want)
    if tokens:
        want = want.permute(0, 2, 3, 1).reshape(N, H * W, C)
    got = ops.groupnorm(x, w, b, 32, 1e-5, silu=silu, tokens=tokens)
    assert got.shape == want.shape
    # Reference = the same op in fp32 (plain PyTorch) on the same 16-bit inputs.  Both torch's 16-bit kernel and ours
    # round the normalised value to 16 bit, apply SiLU in fp32 and round again, so against the fp32 result each may be
    # off by ~1.5 ulp of the 16-bit type; group statistics are accumulated in a different order (sum/sumsq + Chan merge
    # vs torch's Welford), which moves a value by far less than that.
    ref = F.group_norm(x.float(), 32, w.float(), b.float(), 1e-5)
    if silu:
        ref = F.silu(ref)
    if tokens:
        ref = ref.permute(0, 2, 3, 1).reshape(N, H * W, C)
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    err = (got.float() - ref).abs()
    assert bool((err <= 2.0 * ulp * ref.abs() + 4 * ulp).all()), float(err.max())
    rel = float((got.float() - ref).norm() / ref.norm())
    rel_torch = float((want.float() - ref).norm() / ref.norm())
    assert rel < 1.5 * rel_torch + 1e-6, (rel, rel_torch)  # as accurate as the torch kernel it replaces
    # Bit-identity with torch's own 16-bit kernel is NOT claimed: measured on the MI355X 59-75 % of the elements are
    # identical and the rest differ by one 16-bit ulp (different statistics order and a*x+b vs (x-mean)*rstd*w+b
    # association); what is claimed is the bar above -- within 2 ulp of the fp32 result and as accurate as torch's.
    same = float((got == want).float().mean())
    print(f"groupnorm {dtype} {N}x{C}x{H}x{W} silu={silu} tokens={tokens}: bit-identical to torch {same:.4f}")
    assert same > 0.5, same


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,D", [(4096, 640), (1000, 1280), (77, 2048), (5, 8), (3, 320)])
def test_layernorm(dtype, M, D):
    from elasticdiffusion_official_amd import ops
    x = (torch.randn(M, D, device=DEV) * 2.1 + 0.4).to(dtype)
    w = (1 + 0.2 * torch.randn(D, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(D, device=DEV)).to(dtype)
    got = ops.layernorm(x, w, b, 1e-5)
    ref = F.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5)
    want = F.layer_norm(x, (D,), w, b, 1e-5)
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    err = (got.float() - ref).abs()
    assert bool((err <= 1.0 * ulp * ref.abs() + 2 * ulp).all()), float(err.max())  # single rounding: within 1 ulp of fp32
    rel = float((got.float() - ref).norm() / ref.norm())
    rel_torch = float((want.float() - ref).norm() / ref.norm())
    assert rel < 1.2 * rel_torch + 1e-6, (rel, rel_torch)
    got3 = ops.layernorm(x.view(1, M, D), w, b, 1e-5)
    assert torch.equal(got3.view(M, D), got)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W", [(3, 320, 32, 32), (2, 640, 16, 16), (2, 1280, 8, 8), (1, 960, 64, 64), (2, 1920, 16, 16),
                                     (1, 2560, 8, 8), (2, 256, 4, 2)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_channels_last(dtype, N, C, H, W, silu):
    """ed_groupnorm_nhwc: channels-last in / out, vectors straddling group boundaries (cpg = 10, 20, 30, 60, 1)."""
    from elasticdiffusion_official_amd import ops
    x = (torch.randn(N, C, H, W, device=DEV) * 1.7 + 0.3).to(dtype).contiguous(memory_format=torch.channels_last)
    w = (1 + 0.2 * torch.randn(C, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(C, device=DEV)).to(dtype)
    got = ops.groupnorm_nhwc(x, w, b, 32, 1e-5, silu=silu)
    assert got.shape == x.shape and got.is_contiguous(memory_format=torch.channels_last)
    ref = F.group_norm(x.float(), 32, w.float(), b.float(), 1e-5)
    want = F.group_norm(x.contiguous(), 32, w, b, 1e-5)
    if silu:
        ref, want = F.silu(ref), F.silu(want)
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    err = (got.float() - ref).abs()
    assert bool((err <= 2.0 * ulp * ref.abs() + 4 * ulp).all()), float(err.max())
    rel = float((got.float() - ref).norm() / ref.norm())
    rel_torch = float((want.float() - ref).norm() / ref.norm())
    assert rel < 1.5 * rel_torch + 1e-6, (rel, rel_torch)
    # token view of the result == permute of the NCHW result
    tok = got.permute(0, 2, 3, 1).reshape(N, H * W, C)
    assert tok.data_ptr() == got.data_ptr()
    # folded convolution bias + time-embedding add: exactly the kernel applied to the pre-added tensor
    kb, cb = torch.randn(C, device=DEV).to(dtype), torch.randn(N, C, device=DEV).to(dtype)
    pre = ((x + kb[None, :, None, None]) + cb[:, :, None, None]).contiguous(memory_format=torch.channels_last)
    assert torch.equal(ops.groupnorm_nhwc(x, w, b, 32, 1e-5, silu=silu, chan_bias=cb, conv_bias=kb),
                       ops.groupnorm_nhwc(pre, w, b, 32, 1e-5, silu=silu))


@pytest.mark.parametrize("N,C,H,W", [(20, 1280, 32, 32), (6, 640, 64, 64), (3, 320, 128, 128), (5, 2560, 32, 32)])
def test_groupnorm_channels_last_at_the_unet_shapes_many_chunks(N, C, H, W):
    """Round 6 launch plan (gn_plan): block size = the column count rounded to whole waves (C = 1280: 192 threads), ~2048 blocks per launch,
    statistics finalised by a separate N-block launch -- many chunks per sample, chunk boundaries inside image rows, two columns per thread
    (C = 2560).  Against fp32 torch, and launch-to-launch bit-identical (fixed summation order, no atomics)."""
    from elasticdiffusion_official_amd import ops
    g = torch.Generator().manual_seed(C + N)
    x = (torch.randn(N, C, H, W, generator=g) * 2.1 - 0.4).to(DEV, torch.float16).contiguous(memory_format=torch.channels_last)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV, torch.float16)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV, torch.float16)
    got = ops.groupnorm_nhwc(x, w, b, 32, 1e-5, silu=True)
    ref = F.silu(F.group_norm(x.float(), 32, w.float(), b.float(), 1e-5))
    ulp = 2.0 ** -11
    err = (got.float() - ref).abs()
    assert bool((err <= 2.0 * ulp * ref.abs() + 4 * ulp).all()), float(err.max())
    for _ in range(3):
        assert torch.equal(got, ops.groupnorm_nhwc(x, w, b, 32, 1e-5, silu=True))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C1,C2,H,W", [(20, 1280, 1280, 32, 32), (6, 1280, 640, 32, 32), (3, 1280, 640, 64, 64), (2, 640, 320, 64, 64),
                                         (2, 320, 320, 128, 128), (5, 64, 32, 7, 5), (1, 8, 248, 3, 3)])
def test_groupnorm_of_a_concatenation_that_is_never_written(dtype, N, C1, C2, H, W):
    """ed_groupnorm_nhwc_cat (round 6): GroupNorm + SiLU of cat([x1, x2], 1) reading the two channels-last sources in place -- the up
    blocks' ResnetBlock2D(cat([hidden, skip])).  Groups straddle the seam (C = 1920, 960: 60 / 30 channels per group, seam at 21.33
    groups).  Bit-identical to ed_groupnorm_nhwc on the materialised concatenation; both activations; wrapper refusals."""
    from elasticdiffusion_official_amd import ops
    g = torch.Generator().manual_seed(N * C1 + C2 + H)
    cl = torch.channels_last
    x1 = (torch.randn(N, C1, H, W, generator=g) * 1.7 + 0.3).to(DEV, dtype).contiguous(memory_format=cl)
    x2 = (torch.randn(N, C2, H, W, generator=g) * 0.6 - 1.1).to(DEV, dtype).contiguous(memory_format=cl)
    C = C1 + C2
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV, dtype)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV, dtype)
    G = 32 if C % 32 == 0 and C // 32 >= 8 else 8
    cat = torch.cat([x1, x2], dim=1).contiguous(memory_format=cl)
    for silu in (True, False):
        got = ops.groupnorm_nhwc_cat(x1, x2, w, b, G, 1e-5, silu=silu)
        assert got.shape == cat.shape and got.is_contiguous(memory_format=cl)
        assert torch.equal(got, ops.groupnorm_nhwc(cat, w, b, G, 1e-5, silu=silu))
    ref = F.silu(F.group_norm(cat.float(), G, w.float(), b.float(), 1e-5))
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    err = (ops.groupnorm_nhwc_cat(x1, x2, w, b, G, 1e-5, silu=True).float() - ref).abs()
    assert bool((err <= 2.0 * ulp * ref.abs() + 4 * ulp).all()), float(err.max())
    assert not ops.groupnorm_nhwc_cat_ok(x1.contiguous(), x2, G)                       # NCHW memory
    assert not ops.groupnorm_nhwc_cat_ok(x1, x2.float(), G) and not ops.groupnorm_nhwc_cat_ok(x1, x2[:, :, :-1], G)
    with pytest.raises(RuntimeError):
        ops.groupnorm_nhwc_cat(x1, x2.to(torch.float32), w, b, G, 1e-5)


@pytest.mark.gpu
def test_resnet_block_on_a_skip_concatenation_matches_the_cat_path():
    """ResnetBlock2D.forward_cat (the up blocks' call): norm1 through ed_groupnorm_nhwc_cat, the 1x1 shortcut split along K -- against the same
    block on torch.cat([x, skip]).  The only arithmetic difference is one 16-bit rounding of the shortcut's first partial sum; the fallback
    (switch off, fp32 inputs) IS the cat path."""
    from elasticdiffusion_official_amd import models as M, ops
    torch.manual_seed(5)
    cl = torch.channels_last
    for (B, C1, C2, cout, S) in [(6, 1280, 640, 1280, 32), (2, 640, 320, 640, 64), (1, 320, 320, 320, 128)]:
        blk = M.ResnetBlock2D(C1 + C2, cout, 1280).to(DEV, torch.float16).eval().requires_grad_(False).to(memory_format=cl)
        x = torch.randn(B, C1, S, S, device=DEV).half().contiguous(memory_format=cl)
        skip = (torch.randn(B, C2, S, S, device=DEV) * 0.7).half().contiguous(memory_format=cl)
        temb = torch.randn(B, 1280, device=DEV).half()
        ops.TIMER.start()
        got = blk.forward_cat(x, skip, temb)
        launched = set(ops.TIMER.stop())
        assert "ed_groupnorm_nhwc_cat" in launched and "ed_conv3x3_nhwc" in launched, launched
        want = blk(torch.cat([x, skip], dim=1), temb)
        ref = blk.float()(torch.cat([x, skip], dim=1).float(), temb.float())
        blk.half()
        e_got = float((got.float() - ref).norm() / ref.norm())
        e_want = float((want.float() - ref).norm() / ref.norm())
        assert got.shape == want.shape and got.is_contiguous(memory_format=cl)
        assert e_got < 1.15 * e_want + 2e-5, (e_got, e_want)
        keep = M.FUSED_SKIP_CAT
        M.FUSED_SKIP_CAT = False
        try:
            assert torch.equal(blk.forward_cat(x, skip, temb), want)
        finally:
            M.FUSED_SKIP_CAT = keep


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,D", [(300, 320), (77, 640), (1024, 1280), (5, 2048)])
def test_layernorm_and_add_layernorm_on_the_fp32_residual_stream(dtype, M, D):
    """ed_layernorm_s32 / ed_add_layernorm_s32 (round 6, the fp32-residual-stream tolerance mode): the stream is fp32, the branch result
    and the normalised output 16-bit.  Against fp32 torch: the sum is EXACT (a widened + b), the LayerNorm within one 16-bit rounding."""
    from elasticdiffusion_official_amd import ops
    g = torch.Generator().manual_seed(M + D)
    x = (torch.randn(M, D, generator=g) * 3 + 0.5).to(DEV)
    a = torch.randn(M, D, generator=g).to(DEV, dtype)
    w = (1 + 0.2 * torch.randn(D, generator=g)).to(DEV, dtype)
    b = (0.1 * torch.randn(D, generator=g)).to(DEV, dtype)
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    got = ops.layernorm_s32(x, w, b, 1e-5)
    ref = F.layer_norm(x, (D,), w.float(), b.float(), 1e-5)
    assert got.dtype == dtype
    err = (got.float() - ref).abs()
    assert bool((err <= 1.0 * ulp * ref.abs() + 2 * ulp).all()), float(err.max())
    s, ln = ops.add_layernorm_s32(a, x, w, b, 1e-5)
    assert s.dtype == torch.float32 and ln.dtype == dtype
    assert torch.equal(s, a.float() + x)
    ref = F.layer_norm(a.float() + x, (D,), w.float(), b.float(), 1e-5)
    err = (ln.float() - ref).abs()
    assert bool((err <= 1.0 * ulp * ref.abs() + 2 * ulp).all()), float(err.max())
    assert torch.equal(ln, ops.layernorm_s32(s, w, b, 1e-5))
    s2, ln2 = ops.add_layernorm_s32(a, x, w, b, 1e-5)
    assert torch.equal(s2, s) and torch.equal(ln2, ln)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(3, 320, 32, 32), (2, 1280, 8, 8), (1, 960, 64, 64), (1, 2560, 8, 8), (6, 640, 64, 64)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_channels_last_on_the_fp32_residual_stream(dtype, N, C, H, W, silu):
    """ed_groupnorm_nhwc_s32: fp32 channels-last stream in, GroupNorm (+SiLU) in the model dtype out (the operand of the next GEMM)."""
    from elasticdiffusion_official_amd import ops
    g = torch.Generator().manual_seed(N * C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV, dtype)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV, dtype)
    got = ops.groupnorm_nhwc_s32(x, w, b, 32, 1e-5, silu=silu)
    assert got.dtype == dtype and got.shape == x.shape and got.is_contiguous(memory_format=torch.channels_last)
    ref = F.group_norm(x, 32, w.float(), b.float(), 1e-5)
    if silu:
        ref = F.silu(ref)
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    err = (got.float() - ref).abs()
    assert bool((err <= 2.0 * ulp * ref.abs() + 4 * ulp).all()), float(err.max())
    assert torch.equal(got, ops.groupnorm_nhwc_s32(x, w, b, 32, 1e-5, silu=silu))
    # the 16-bit kernel on the rounded stream differs from this one only by that input rounding
    x16 = x.to(dtype).contiguous(memory_format=torch.channels_last)
    near = ops.groupnorm_nhwc(x16, w, b, 32, 1e-5, silu=silu)
    assert float((near.float() - got.float()).norm() / got.float().norm()) < 8 * ulp


def test_unet_channels_last_path_close():
    """The whole (small) UNet with channels-last activations + NHWC GroupNorm vs the default NCHW path."""
    from elasticdiffusion_official_amd import models as M
    cfg = dict(M.UNET_CONFIGS["sdxl"])
    cfg.update(block_out_channels=(256, 320, 512), heads=(4, 5, 8), transformer_depth=(1, 1, 2), cross_attention_dim=64,
               addition_time_embed_dim=8, pooled_projection_dim=16, sample_size=32)
    torch.manual_seed(0)
    u = M.UNet2DConditionModel(**cfg).to(DEV, torch.bfloat16).eval()
    x = torch.randn(3, 4, 32, 32, device=DEV, dtype=torch.bfloat16)
    e = torch.randn(3, 77, 64, device=DEV, dtype=torch.bfloat16)
    kw = {"text_embeds": torch.randn(3, 16, device=DEV, dtype=torch.bfloat16), "time_ids": torch.zeros(3, 6, device=DEV)}
    t = torch.tensor(500, device=DEV)
    saved = M.CHANNELS_LAST
    with torch.no_grad():
        try:
            M.CHANNELS_LAST = False
            a = u(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"].float()
            M.CHANNELS_LAST = True
            ucl = u.to(memory_format=torch.channels_last)
            b = ucl(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"].float()
            ref = u.float().to(memory_format=torch.contiguous_format)(
                x.float(), t, encoder_hidden_states=e.float(), added_cond_kwargs={k: v.float() for k, v in kw.items()})["sample"]
        finally:
            M.CHANNELS_LAST = saved
    rel = float((a - b).norm() / a.norm())
    ra, rb = float((a - ref).norm() / ref.norm()), float((b - ref).norm() / ref.norm())
    print(f"UNet bf16 vs fp32: NCHW {ra:.3e}, channels_last {rb:.3e}; NCHW vs channels_last {rel:.3e}")
    assert rel < 2e-2 and rb < 1.5 * ra + 1e-3, (rel, ra, rb)
    with pytest.raises(RuntimeError):  # channels-per-group < 8 is rejected by the C ABI, never silently wrong
        from elasticdiffusion_official_amd import ops
        xs = torch.randn(1, 32, 4, 4, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ops.groupnorm_nhwc(xs, torch.ones(32, device=DEV, dtype=torch.bfloat16), torch.zeros(32, device=DEV, dtype=torch.bfloat16), 32, 1e-5)


def test_unet_fused_vs_unfused_close():
    """Whole (small) UNet in bf16 with and without the fused kernels: same output up to bf16 noise."""
    from elasticdiffusion_official_amd import models as M
    cfg = dict(M.UNET_CONFIGS["sdxl"])
    cfg.update(block_out_channels=(64, 128, 256), heads=(1, 2, 4), transformer_depth=(1, 1, 2), cross_attention_dim=64,
               addition_time_embed_dim=8, pooled_projection_dim=16, sample_size=32)
    torch.manual_seed(0)
    u = M.UNet2DConditionModel(**cfg).to(DEV, torch.bfloat16).eval()
    x = torch.randn(3, 4, 32, 32, device=DEV, dtype=torch.bfloat16)
    e = torch.randn(3, 77, 64, device=DEV, dtype=torch.bfloat16)
    kw = {"text_embeds": torch.randn(3, 16, device=DEV, dtype=torch.bfloat16), "time_ids": torch.zeros(3, 6, device=DEV)}
    t = torch.tensor(500, device=DEV)
    saved = (M.FUSED_KERNELS, M.SHORTCUT_AS_GEMM)
    try:
        with torch.no_grad():
            M.FUSED_KERNELS = M.SHORTCUT_AS_GEMM = True
            a = u(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"].float()
            M.FUSED_KERNELS = M.SHORTCUT_AS_GEMM = False
            b = u(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"].float()
    finally:
        M.FUSED_KERNELS, M.SHORTCUT_AS_GEMM = saved  # module switches must not leak into later tests
    rel = float((a - b).norm() / b.norm())
    assert a.is_contiguous() and rel < 2e-2, rel


# ---------------------------------------------------------------------------------------------------
# round 2: flash attention, fused residual adds
# ---------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, H, scale=0.125):
    """fp32 PyTorch reference of the op: softmax(q k^T / sqrt(64)) v per head, on the same 16-bit inputs."""
    B, Nq, HD = q.shape
    qf, kf, vf = (t.float().view(B, -1, H, 64).transpose(1, 2) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * scale
    return (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).reshape(B, Nq, HD)


LN2 = 0.6931471805599453


def _exp2_q(q):
    """Exponent-domain queries for v_path 6: q * (softmax scale * log2 e), rounded once to the 16-bit type (what
    models.Attention folds into the query projection weights).  The op on them is softmax base 2 = softmax(ln 2 * q' k^T)."""
    return (q.float() * (0.125 * 1.4426950408889634)).to(q.dtype)


def _flash(ops, q, k, v, H, v_path):
    """ops.flash_attention for any variant, plus the fp32 reference and SDPA on the inputs that variant really sees."""
    if v_path == 6:
        q = _exp2_q(q)
    got = ops.flash_attention(q, k, v, H, v_path=v_path, prescaled=v_path == 6)
    B, Nq = q.shape[:2]
    scale = LN2 if v_path == 6 else 0.125
    sdpa = F.scaled_dot_product_attention(*(t.reshape(B, -1, H, 64).transpose(1, 2) for t in (q, k, v)), scale=scale)
    return got, _attn_ref(q, k, v, H, scale), sdpa.transpose(1, 2).reshape(B, Nq, H * 64)


@pytest.mark.parametrize("v_path", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])  # bit 0: V staging; bit 1: 64 rows/wave; 4-7, 9, 10: pipelined; 8: small-KV
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 10, 4096, 4096), (3, 20, 1024, 1024), (2, 20, 1024, 77), (1, 10, 4096, 77),
                                       (2, 2, 256, 256), (1, 4, 64, 64), (1, 1, 100, 77), (2, 3, 200, 333), (1, 2, 1, 1)])
def test_flash_attention(dtype, B, H, Nq, Nk, v_path):
    """ed_flash_attention vs the fp32 reference; it must be as accurate as the library kernel it replaces (SDPA)."""
    from elasticdiffusion_official_amd import ops
    if v_path == 8 and Nk > 96:
        pytest.skip("the small-KV kernel takes Nk <= 96")
    if v_path == 6 and Nk < 128:
        pytest.skip("the exponent-domain kernel takes Nk >= 128 (ops.flash_prescale)")
    if v_path in (7, 9, 10) and Nk < 64:
        pytest.skip("nothing to pipeline below one full tile")
    g = torch.Generator(device=DEV).manual_seed(Nq * 7 + Nk)
    q, k, v = (torch.randn(B, n, H * 64, device=DEV, generator=g).mul(s).to(dtype) for n, s in ((Nq, 1.5), (Nk, 1.5), (Nk, 1.0)))
    got, ref, sdpa = _flash(ops, q, k, v, H, v_path)
    assert got.shape == (B, Nq, H * 64) and got.is_contiguous()
    err = float((got.float() - ref).abs().max())
    err_sdpa = float((sdpa.float() - ref).abs().max())
    rel = float((got.float() - ref).norm() / ref.norm())
    rel_sdpa = float((sdpa.float() - ref).norm() / ref.norm())
    print(f"flash {dtype} B{B} H{H} Nq{Nq} Nk{Nk} path{v_path}: max|err| {err:.2e} (sdpa {err_sdpa:.2e})  rel {rel:.2e} (sdpa {rel_sdpa:.2e})")
    assert bool(torch.isfinite(got).all())
    assert rel < 1.5 * rel_sdpa + 1e-4, (rel, rel_sdpa)
    assert err < 3.0 * err_sdpa + 4e-3, (err, err_sdpa)


def test_flash_attention_strided_inputs_and_outlier_rows():
    """q/k/v as column slices of one fused projection output; one query row with a huge score (online-softmax rescale
    path: the running max jumps by > 100 in a late tile) and one all-equal row."""
    from elasticdiffusion_official_amd import ops
    B, H, N = 2, 5, 640
    g = torch.Generator(device=DEV).manual_seed(3)
    qkv = torch.randn(B, N, 3 * H * 64, device=DEV, generator=g).to(torch.bfloat16)
    q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
    k[0, 600, :64] = 12.0 * q[0, 5, :64].sign()   # query 5 of head 0 matches key 600 (10th tile) with score >> others
    q[1, 7] = 0                                    # uniform attention for query 7 of batch 1
    got = ops.flash_attention(q, k, v, H)
    ref = _attn_ref(q, k, v, H)
    assert float((got.float() - ref).abs().max()) < 3e-2
    assert float((got[0, 5, :64].float() - v[0, 600, :64].float()).abs().max()) < 3e-2   # ~one-hot row
    assert float((got[1, 7].float() - v[1].float().mean(0)).abs().max()) < 2e-2          # ~mean of V
    base = ops.flash_attention(q, k, v, H, v_path=0)
    for path in (0, 1, 2, 3):
        again = ops.flash_attention(q.contiguous(), k.contiguous(), v.contiguous(), H, v_path=path)
        assert torch.equal(again, base)  # strides, the V staging path and the rows-per-wave variant do not change a bit
    # the pipelined kernel (deferred rescale: the outlier row takes the rescale branch in a late tile) on strided and on
    # contiguous inputs: identical to itself, and within the rounding of P of the others
    for path in (4, 5, 6, 7, 9, 10):
        qq = _exp2_q(q) if path == 6 else q
        kw = dict(v_path=path, prescaled=path == 6)
        piped = ops.flash_attention(qq, k, v, H, **kw)
        assert torch.equal(piped, ops.flash_attention(qq.contiguous(), k.contiguous(), v.contiguous(), H, **kw))
        # v_path 6 is judged on the queries it is given (in the model the factor lives in the projection weights; here the
        # bf16 rounding of q * c alone moves these large scores by several percent)
        ref_p = _attn_ref(qq, k, v, H, LN2) if path == 6 else ref
        assert float((piped.float() - ref_p).abs().max()) < 3e-2
        assert float((piped[0, 5, :64].float() - v[0, 600, :64].float()).abs().max()) < 3e-2
        assert float((piped[1, 7].float() - v[1].float().mean(0)).abs().max()) < 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hd", [40, 80, 160])
@pytest.mark.parametrize("B,H,Nq,Nk", [(2, 8, 4096, 4096), (3, 8, 1024, 1024), (2, 8, 256, 256), (2, 8, 64, 64), (2, 8, 1024, 77), (1, 3, 100, 333), (1, 2, 1, 1)])
def test_flash_attention_sd1_head_dims(dtype, hd, B, H, Nq, Nk):
    """ed_flash_attention at SD 1.x's head dimensions (8 heads of 40 / 80 / 160: k_flash_attn_gen, the head dimension
    zero-padded to a multiple of 32 inside the kernel) vs the fp32 reference; as accurate as SDPA, which it replaces there
    (AOTriton's attn_fwd was 25 % of the SD1.5 workload's GPU time); q / k / v as column slices of one fused projection."""
    from elasticdiffusion_official_amd import ops
    g = torch.Generator(device=DEV).manual_seed(hd * 7 + Nk)
    if Nq == Nk:
        qkv = torch.randn(B, Nq, 3 * H * hd, device=DEV, generator=g).mul(1.5).to(dtype)
        q, k, v = qkv[..., :H * hd], qkv[..., H * hd:2 * H * hd], qkv[..., 2 * H * hd:]
    else:
        q, k, v = (torch.randn(B, n, H * hd, device=DEV, generator=g).mul(s_).to(dtype) for n, s_ in ((Nq, 1.5), (Nk, 1.5), (Nk, 1.0)))
    got = ops.flash_attention(q, k, v, H)
    assert got.shape == (B, Nq, H * hd) and got.is_contiguous() and bool(torch.isfinite(got).all())
    qf, kf, vf = (t.float().reshape(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, dim=-1) @ vf).transpose(1, 2).reshape(B, Nq, H * hd)
    sdpa = F.scaled_dot_product_attention(*(t.reshape(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))).transpose(1, 2).reshape(B, Nq, H * hd)
    err, err_sdpa = float((got.float() - ref).abs().max()), float((sdpa.float() - ref).abs().max())
    rel, rel_sdpa = float((got.float() - ref).norm() / ref.norm()), float((sdpa.float() - ref).norm() / ref.norm())
    assert rel < 1.5 * rel_sdpa + 1e-4, (rel, rel_sdpa)
    assert err < 3.0 * err_sdpa + 4e-3, (err, err_sdpa)
    with pytest.raises(RuntimeError):
        ops.flash_attention(q, k, v, H, v_path=4)     # the variants are head_dim 64 only


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Nk", [130, 333, 192, 4096])
def test_flash_attention_never_reads_past_the_keys(dtype, Nk):
    """the pipelined kernels issue the loads of tiles t+1 / t+2 unconditionally and leave the rows past Nk of
    a ragged tile to the buffer range check -- so K and V sit at the END of an allocation whose tail is poisoned with NaN /
    Inf: a load that escaped the range check (an offset the check does not cover) would turn the output non-finite or wrong."""
    from elasticdiffusion_official_amd import ops
    B, H, Nq = 2, 3, 256
    g = torch.Generator(device=DEV).manual_seed(Nk)
    tail = 3 * 64 * H * 64                      # three tiles' worth of poison behind the last key row of the last batch
    pool = torch.empty(2, B * Nk * H * 64 + tail, device=DEV, dtype=dtype)
    pool[:, B * Nk * H * 64:] = float("nan")
    pool[:, B * Nk * H * 64::2] = float("inf")
    k, v = (pool[i, :B * Nk * H * 64].view(B, Nk, H * 64) for i in range(2))
    k.copy_(torch.randn(B, Nk, H * 64, device=DEV, generator=g))
    v.copy_(torch.randn(B, Nk, H * 64, device=DEV, generator=g))
    q = torch.randn(B, Nq, H * 64, device=DEV, generator=g).to(dtype)
    for path in (4, 5, 6, 7, 9, 10, 0):
        got, ref, _ = _flash(ops, q, k, v, H, path)
        assert bool(torch.isfinite(got).all()), path
        assert float((got.float() - ref).abs().max()) < (3e-2 if dtype == torch.bfloat16 else 6e-3), path


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_flash_attention_deferred_rescale_branches(dtype):
    """The pipelined kernel moves a row's reference maximum only when a tile's maximum exceeds it by more than 2^6 (exp2
    domain): force (a) rows whose maximum creeps up by less than the threshold every tile (never rescaled after the first
    tile: P grows up to 2^6), (b) rows that jump far above it in a late tile, (c) rows whose first tile is the largest --
    against the fp32 reference on the full tensor (CDNA guide 5.4 rule 26: a rare data-dependent branch needs its own test)."""
    from elasticdiffusion_official_amd import ops
    B, H, N = 1, 2, 1024
    g = torch.Generator(device=DEV).manual_seed(11)
    q = torch.randn(B, N, H * 64, device=DEV, generator=g).to(dtype)
    k = (0.3 * torch.randn(B, N, H * 64, device=DEV, generator=g)).to(dtype)
    v = torch.randn(B, N, H * 64, device=DEV, generator=g).to(dtype)
    qh = q.view(B, N, H, 64)
    kh = k.view(B, N, H, 64)
    for tile in range(1, 16):      # (a) head 0, query 3: the score of key 64*tile grows by 0.35 (0.5 in log2 units) per tile
        kh[0, 64 * tile, 0] = (qh[0, 3, 0].float() * (0.35 * tile * 8.0 / float(qh[0, 3, 0].float().pow(2).sum()))).to(dtype)
    kh[0, 900, 1] = (qh[0, 9, 1].float().sign() * 6.0).to(dtype)   # (b) head 1, query 9: a huge score in tile 14
    kh[0, 5, 1] = (qh[0, 17, 1].float().sign() * 6.0).to(dtype)     # (c) head 1, query 17: the largest score in tile 0
    for path in (4, 5, 6, 7, 9, 10, 0):
        got, ref, _ = _flash(ops, q, k, v, H, path)
        err = float((got.float() - ref).abs().max())
        assert err < (2e-2 if dtype == torch.bfloat16 else 4e-3), (path, err)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,D", [(4096, 640), (1000, 1280), (77, 2048), (5, 8), (3, 320)])
def test_add_layernorm(dtype, M, D):
    from elasticdiffusion_official_amd import ops
    a = (torch.randn(M, D, device=DEV) * 2.1 + 0.4).to(dtype)
    b = (torch.randn(M, D, device=DEV) * 0.7).to(dtype)
    w = (1 + 0.2 * torch.randn(D, device=DEV)).to(dtype)
    bb = (0.1 * torch.randn(D, device=DEV)).to(dtype)
    s, out = ops.add_layernorm(a, b, w, bb, 1e-5)
    assert torch.equal(s, a + b)                                   # torch's 16-bit add, bit for bit
    assert torch.equal(out, ops.layernorm(a + b, w, bb, 1e-5))      # and exactly ed_layernorm of that sum


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W", [(3, 320, 32, 32), (2, 640, 64, 64), (2, 1280, 8, 8), (1, 64, 8, 16)])
def test_tokens_add_nchw(dtype, N, C, H, W):
    from elasticdiffusion_official_amd import ops
    x = torch.randn(N, C, H, W, device=DEV).to(dtype)
    tok = torch.randn(N, H * W, C, device=DEV).to(dtype)
    got = ops.tokens_add_nchw(x, tok)
    want = x + tok.view(N, H, W, C).permute(0, 3, 1, 2)
    assert got.is_contiguous() and torch.equal(got, want.contiguous())


@pytest.mark.parametrize("N,C,H,W,tokens", [(2, 320, 128, 128, False), (2, 640, 64, 64, True), (1, 960, 128, 128, False),
                                            (3, 1920, 64, 64, False)])
def test_groupnorm_split_path_for_large_groups(N, C, H, W, tokens):
    """Groups above 64 K elements take the two-launch (chunked Welford partials + apply) path: same accuracy bar as the
    single-launch kernel, and the two agree to within one 16-bit ulp (the statistics are merged in a different order)."""
    from elasticdiffusion_official_amd import ops
    dtype = torch.bfloat16
    x = (torch.randn(N, C, H, W, device=DEV) * 1.7 + 0.3).to(dtype)
    w = (1 + 0.2 * torch.randn(C, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(C, device=DEV)).to(dtype)
    cb = torch.randn(N, C, device=DEV).to(dtype)
    kb = torch.randn(C, device=DEV).to(dtype)
    kw = dict(silu=True, tokens=tokens) if tokens else dict(silu=True, chan_bias=cb, conv_bias=kb)
    assert ops._hip.lib().ed_groupnorm_workspace(N, C, H * W, 32) > 0
    got = ops.groupnorm(x, w, b, 32, 1e-5, **kw)
    ops.GROUPNORM_SPLIT = False
    try:
        one = ops.groupnorm(x, w, b, 32, 1e-5, **kw)
    finally:
        ops.GROUPNORM_SPLIT = True
    xin = x.float() if tokens else (x + kb[None, :, None, None] + cb[:, :, None, None]).float()
    ref = F.silu(F.group_norm(xin, 32, w.float(), b.float(), 1e-5))
    if tokens:
        ref = ref.permute(0, 2, 3, 1).reshape(N, H * W, C)
    ulp = 2.0 ** -8
    for y in (got, one):
        assert bool(((y.float() - ref).abs() <= 2.0 * ulp * ref.abs() + 4 * ulp).all())
    assert float((got.float() - one.float()).abs().max()) <= float((2 * ulp * ref.abs() + 2 * ulp).max())
    print(f"groupnorm split vs single launch {N}x{C}x{H}x{W}: identical {float((got == one).float().mean()):.4f}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W", [(3, 320, 32, 32), (2, 640, 16, 16), (2, 64, 4, 2)])
def test_groupnorm_with_channel_bias(dtype, N, C, H, W):
    """ed_groupnorm(chan_bias=temb) == ed_groupnorm(x + temb[:, :, None, None]) bit for bit."""
    from elasticdiffusion_official_amd import ops
    x = (torch.randn(N, C, H, W, device=DEV) * 1.7 + 0.3).to(dtype)
    cb = torch.randn(N, C, device=DEV).to(dtype)
    w = (1 + 0.2 * torch.randn(C, device=DEV)).to(dtype)
    b = (0.1 * torch.randn(C, device=DEV)).to(dtype)
    got = ops.groupnorm(x, w, b, 32, 1e-5, silu=True, chan_bias=cb)
    want = ops.groupnorm(x + cb[:, :, None, None], w, b, 32, 1e-5, silu=True)
    assert torch.equal(got, want)
    kb = torch.randn(C, device=DEV).to(dtype)  # + the producing convolution's bias, added (and rounded) first
    got = ops.groupnorm(x, w, b, 32, 1e-5, silu=True, chan_bias=cb, conv_bias=kb)
    want = ops.groupnorm((x + kb[None, :, None, None]) + cb[:, :, None, None], w, b, 32, 1e-5, silu=True)
    assert torch.equal(got, want)
    got = ops.groupnorm(x, w, b, 32, 1e-5, conv_bias=kb)
    assert torch.equal(got, ops.groupnorm(x + kb[None, :, None, None], w, b, 32, 1e-5))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,C,H,W", [(3, 320, 32, 32), (2, 1280, 8, 8), (1, 64, 2, 4)])
def test_bias_residual_add(dtype, N, C, H, W):
    from elasticdiffusion_official_amd import ops
    h, res = (torch.randn(N, C, H, W, device=DEV).to(dtype) for _ in range(2))
    hb, rb = (torch.randn(C, device=DEV).to(dtype) for _ in range(2))
    bc = lambda v: v[None, :, None, None]
    assert torch.equal(ops.bias_residual_add(h, hb, res), res + (h + bc(hb)))
    assert torch.equal(ops.bias_residual_add(h, hb, res, rb), (res + bc(rb)) + (h + bc(hb)))
    assert torch.equal(ops.bias_residual_add(h, None, res), res + h)
    if C % 8 == 0:  # channels-last operands: same values, channels-last result
        hc, rc = (t.contiguous(memory_format=torch.channels_last) for t in (h, res))
        got = ops.bias_residual_add(hc, hb, rc, rb)
        assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, (res + bc(rb)) + (h + bc(hb)))


def test_unet_round2_fusions_close():
    """Whole (small, head_dim 64) UNet in bf16: flash attention + fused QKV + fused adds vs all of them off."""
    from elasticdiffusion_official_amd import models as M
    torch.manual_seed(0)
    u = M.UNet2DConditionModel(**M.SMALL_UNET_CONFIGS["sdxl"]).to(DEV, torch.bfloat16).eval()
    x = torch.randn(3, 4, 64, 64, device=DEV, dtype=torch.bfloat16)
    e = torch.randn(3, 77, 64, device=DEV, dtype=torch.bfloat16)
    kw = {"text_embeds": torch.randn(3, 32, device=DEV, dtype=torch.bfloat16), "time_ids": torch.zeros(3, 6, device=DEV)}
    t = torch.tensor(500, device=DEV)
    names = ("FLASH_ATTENTION", "FUSED_QKV", "FUSED_ADD_LAYERNORM", "FUSED_TOKENS_ADD", "FUSED_TEMB_ADD", "FUSED_CONV_BIAS")
    saved = {n: getattr(M, n) for n in names}
    try:
        with torch.no_grad():
            a = u(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"].float()
            for n in names:
                setattr(M, n, False)
            b = u(x, t, encoder_hidden_states=e, added_cond_kwargs=kw)["sample"].float()
            ref = u.float()(x.float(), t, encoder_hidden_states=e.float(),
                            added_cond_kwargs={k: v.float() for k, v in kw.items()})["sample"]
    finally:
        for n, v in saved.items():
            setattr(M, n, v)
    ra, rb = float((a - ref).norm() / ref.norm()), float((b - ref).norm() / ref.norm())
    print(f"small SDXL UNet bf16 vs fp32: fused {ra:.3e}, unfused {rb:.3e}")
    assert ra < 1.5 * rb + 1e-3, (ra, rb)


@pytest.mark.parametrize("rows,cols,scale", [(7, 4096, 0.0442), (3, 32768, 0.0442), (64, 256, 1.0), (1, 4, 2.0), (5, 1028, 0.3)])
def test_softmax_rows(rows, cols, scale):
    from elasticdiffusion_official_amd import ops
    x = torch.randn(rows, cols, device=DEV) * 3.0
    x[0, 3] = 40.0  # a dominant entry
    want = torch.softmax(x.double() * scale, dim=-1)
    got = ops.softmax_rows_(x.clone(), scale)
    assert float((got.double() - want).abs().max()) < 2e-6
    assert float((got.sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("B,N,C", [(2, 1024, 512), (1, 4096, 64), (3, 260, 32)])
def test_vae_attention_matches_sdpa_math(B, N, C):
    """ops.vae_attention (fp32 GEMM -> ed_softmax_rows -> fp32 GEMM, chunked over query rows) vs the fp64 formula, and as
    close to it as torch's own SDPA on the same fp32 inputs."""
    from elasticdiffusion_official_amd import ops
    q, k, v = (torch.randn(B, N, C, device=DEV) for _ in range(3))
    ref = torch.softmax(q.double() @ k.double().transpose(1, 2) * C ** -0.5, dim=-1) @ v.double()
    saved = ops.VAE_ATTENTION_CHUNK_BYTES
    try:
        for chunk in (saved, 4 * N * B * 300):  # one block of rows, and several ragged ones
            ops.VAE_ATTENTION_CHUNK_BYTES = chunk
            got = ops.vae_attention(q, k, v)
            err = float((got.double() - ref).abs().max())
            sdpa = F.scaled_dot_product_attention(q.unsqueeze(1), k.unsqueeze(1), v.unsqueeze(1)).squeeze(1)
            err_sdpa = float((sdpa.double() - ref).abs().max())
            assert err < max(2 * err_sdpa, 2e-5), (err, err_sdpa)
    finally:
        ops.VAE_ATTENTION_CHUNK_BYTES = saved


@pytest.mark.parametrize("N,C,H,W", [(5, 128, 64, 256), (2, 256, 32, 32), (1, 512, 16, 20), (3, 64, 2, 2), (1, 128, 300, 260)])
@pytest.mark.parametrize("silu", [True, False])
def test_groupnorm_f32(N, C, H, W, silu):
    """ed_groupnorm_f32 (the fp32 VAE's GroupNorm [+SiLU]) vs torch's fp32 op and an fp64 reference: as accurate as torch's."""
    from elasticdiffusion_official_amd import ops
    x = torch.randn(N, C, H, W, device=DEV) * 1.7 + 0.6
    w = 1 + 0.2 * torch.randn(C, device=DEV)
    b = 0.1 * torch.randn(C, device=DEV)
    got = ops.groupnorm_f32(x, w, b, 32, 1e-6, silu=silu)
    want = F.group_norm(x, 32, w, b, 1e-6)
    ref = F.group_norm(x.double(), 32, w.double(), b.double(), 1e-6)
    if silu:
        want, ref = F.silu(want), F.silu(ref)
    err, err_torch = float((got.double() - ref).abs().max()), float((want.double() - ref).abs().max())
    assert err < 2.0 * err_torch + 2e-6, (err, err_torch)
    assert got.shape == x.shape and got.is_contiguous()


def test_fp32_vae_uses_hip_groupnorm_and_matches_torch():
    """The fp32 VAE (reduced width) with ed_groupnorm_f32 + ed_softmax_rows vs the same module on plain torch ops."""
    from elasticdiffusion_official_amd import models as M, ops
    _, vae = M.build_models("XL1.0", device=DEV, small=True)[:2]
    x = torch.rand(2, 3, 64, 128, device=DEV) * 2 - 1
    z = torch.randn(2, 4, 16, 32, device=DEV)
    seen = set()
    orig = ops._call

    def spy(name, *a):
        seen.add(name)
        return orig(name, *a)

    ops._call = spy
    try:
        with torch.no_grad():
            enc, dec = vae.encode(x).latent_dist.mean, vae.decode(z).sample
    finally:
        ops._call = orig
    assert {"ed_groupnorm_f32", "ed_softmax_rows"} <= seen, seen
    saved = (M.VAE_HIP_GROUPNORM, M.VAE_HIP_ATTENTION)
    try:
        M.VAE_HIP_GROUPNORM = M.VAE_HIP_ATTENTION = False
        with torch.no_grad():
            enc0, dec0 = vae.encode(x).latent_dist.mean, vae.decode(z).sample
    finally:
        M.VAE_HIP_GROUPNORM, M.VAE_HIP_ATTENTION = saved
    for a, b in ((enc, enc0), (dec, dec0)):
        assert float((a - b).norm() / b.norm()) < 2e-5


# ---------------------------------------------------------------------------------------------------
# round 4: the 8-phase MFMA main loop behind ed_geglu_gemm / ed_linear / ed_conv3x3_nhwc (csrc/gemm_kernels.hip)
# ---------------------------------------------------------------------------------------------------
def _asym(shape, g, scale=1.0):
    """uniform [-1, 1) data: asymmetric by construction, so a transposed operand or store cannot pass"""
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.fixture
def any_grid(monkeypatch):
    """the wrappers refuse grids too small to fill the chip (the model then keeps the library call); tests want every shape"""
    from elasticdiffusion_official_amd import ops
    monkeypatch.setattr(ops, "GEMM_MIN_BLOCKS", 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,I", [(256, 64, 128), (300, 192, 256), (1000, 320, 1280), (4099, 640, 2560), (2048, 1280, 5120), (1, 64, 128)])
def test_geglu_gemm(dtype, M, K, I, any_grid):
    """ed_geglu_gemm vs the fp32 reference of GEGLU.forward on the same 16-bit inputs: ragged M, one K tile, odd tile counts,
    the SDXL shapes' K / I; at least as accurate as the pair it replaces (16-bit projection output + ed_geglu), and 8 launches
    bit-identical (no atomics, no split-K: any difference between launches would be an LDS race)."""
    from elasticdiffusion_official_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = _asym((M, K), g).to(DEV, dtype)
    w = _asym((2 * I, K), g, K ** -0.5).to(DEV, dtype)
    b = _asym((2 * I,), g).to(DEV, dtype)
    got = ops.geglu_gemm(x, w, b)
    y = x.float() @ w.float().t() + b.float()
    ref = y[:, :I] * F.gelu(y[:, I:])
    unfused = ops.geglu(F.linear(x, w, b), I) if I % 8 == 0 else None
    rel = float((got.float() - ref).norm() / ref.norm())
    rel_unfused = float((unfused.float() - ref).norm() / ref.norm())
    assert got.shape == (M, I) and bool(torch.isfinite(got).all())
    assert rel < 1.1 * rel_unfused + 1e-5, (rel, rel_unfused)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert bool(((got.float() - ref).abs() <= 1.0 * ulp * ref.abs() + 2e-5).all())   # one rounding of an fp32 result
    for _ in range(8):
        assert torch.equal(ops.geglu_gemm(x, w, b), got)
    assert torch.equal(ops.geglu_gemm(x.view(1, M, K), w, b).view(M, I), got)
    nob = ops.geglu_gemm(x, w, None)
    y0 = x.float() @ w.float().t()
    assert float((nob.float() - y0[:, :I] * F.gelu(y0[:, I:])).abs().max()) < 8 * ulp


@pytest.fixture(params=["0", "1"], ids=["tiles256", "tiles128"])
def tile_rows(request, monkeypatch):
    """round 6: the launcher picks 128-row tiles (tile_phases_rows) for under-filled grids -- which every small test shape is -- so the
    GEMM tests pin the mode (ED_GEMM_ROWS is read at every launch) and run BOTH loops on every shape"""
    monkeypatch.setenv("ED_GEMM_ROWS", request.param)
    return request.param


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(256, 64, 256), (300, 128, 200), (1000, 640, 1920), (4099, 640, 640), (2000, 2560, 640), (513, 320, 8)])
def test_linear_hip(dtype, M, K, N, any_grid, tile_rows):
    """ed_linear (+bias, +residual) vs fp32: a single rounding of the fp32 result; ragged M and ragged / partial column blocks."""
    from elasticdiffusion_official_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = _asym((M, K), g).to(DEV, dtype)
    w = _asym((N, K), g, K ** -0.5).to(DEV, dtype)
    b = _asym((N,), g).to(DEV, dtype)
    r = _asym((M, N), g).to(DEV, dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    acc = x.float() @ w.float().t()
    for bias, res in ((b, None), (None, None), (b, r), (None, r)):
        ref = acc + (0 if bias is None else bias.float()) + (0 if res is None else res.float())
        got = ops.linear(x, w, bias, res)
        assert got.shape == (M, N)
        # fp32 summation order differs from torch's matmul: allow the accumulation noise on top of the single rounding
        assert bool(((got.float() - ref).abs() <= 1.0 * ulp * ref.abs() + 1e-4).all()), float((got.float() - ref).abs().max())
        for _ in range(4):
            assert torch.equal(ops.linear(x, w, bias, res), got)
    lib = F.linear(x, w, b)
    ref = acc + b.float()
    assert float((ops.linear(x, w, b).float() - ref).norm()) <= 1.05 * float((lib.float() - ref).norm()) + 1e-6


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,N", [(2, 12, 20, 64, 200), (3, 16, 16, 128, 320), (1, 32, 32, 320, 320), (2, 8, 8, 1280, 640), (5, 7, 9, 192, 72)])
def test_conv3x3_nhwc(dtype, B, H, W, Cin, N, any_grid, tile_rows):
    """ed_conv3x3_nhwc vs F.conv2d in fp32 (+ bias, + per-sample channel bias, + residual): image borders, batch seams inside a
    256-row tile (B H W not a multiple of 256), 1..20 K tiles per tap, partial column blocks; bit-identical over launches."""
    from elasticdiffusion_official_amd import ops
    cl = torch.channels_last
    g = torch.Generator().manual_seed(B * H + Cin)
    x = _asym((B, Cin, H, W), g).to(DEV, dtype).contiguous(memory_format=cl)
    w = _asym((N, Cin, 3, 3), g, (9 * Cin) ** -0.5).to(DEV, dtype).contiguous(memory_format=cl)
    b = _asym((N,), g).to(DEV, dtype)
    sb = _asym((B, N), g).to(DEV, dtype)
    r = _asym((B, N, H, W), g).to(DEV, dtype).contiguous(memory_format=cl)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    acc = F.conv2d(x.float(), w.float(), None, padding=1)
    for bias, sbias, res in ((b, None, None), (None, None, None), (b, sb, None), (b, None, r), (b, sb, r)):
        ref = acc + (0 if bias is None else bias.float()[None, :, None, None]) + (0 if sbias is None else sbias.float()[:, :, None, None]) \
            + (0 if res is None else res.float())
        got = ops.conv3x3_nhwc(x, w, bias, sbias, res)
        assert got.shape == (B, N, H, W) and got.is_contiguous(memory_format=cl)
        assert bool(((got.float() - ref).abs() <= 1.0 * ulp * ref.abs() + 1e-4).all()), float((got.float() - ref).abs().max())
        for _ in range(4):
            assert torch.equal(ops.conv3x3_nhwc(x, w, bias, sbias, res), got)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Hs,Ws,Cin,N", [(2, 6, 10, 64, 200), (3, 8, 8, 128, 320), (1, 16, 16, 320, 320), (2, 4, 4, 1280, 640), (5, 7, 9, 192, 72),
                                           (2, 32, 32, 1280, 1280), (1, 64, 64, 640, 640)])
def test_upsampler_convolution_reads_the_source_in_place(dtype, B, Hs, Ws, Cin, N, any_grid):
    """ed_conv3x3_nhwc_up2x (round 6): Upsample2D = nearest 2x + conv 3x3 as one launch whose A operand is gathered from the low-resolution
    source (output pixel (y, x), tap (dy, dx) -> source ((y + dy) >> 1, (x + dx) >> 1)): image borders, odd source sizes, batch seams inside a
    tile, both K loops (Cin = 64 / 128: the 8-phase loop; >= 192: the long-K loop).  Bit-identical to ed_conv3x3_nhwc on the materialised
    upsampling (same products, same order), within one rounding of fp32 torch, launch-to-launch bit-identical."""
    from elasticdiffusion_official_amd import ops
    cl = torch.channels_last
    g = torch.Generator().manual_seed(B * Hs + Cin)
    x = _asym((B, Cin, Hs, Ws), g).to(DEV, dtype).contiguous(memory_format=cl)
    w = _asym((N, Cin, 3, 3), g, (9 * Cin) ** -0.5).to(DEV, dtype).contiguous(memory_format=cl)
    b = _asym((N,), g).to(DEV, dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    up = F.interpolate(x, scale_factor=2.0, mode="nearest").contiguous(memory_format=cl)
    for bias in (b, None):
        got = ops.conv3x3_nhwc_up2x(x, w, bias)
        assert got.shape == (B, N, 2 * Hs, 2 * Ws) and got.is_contiguous(memory_format=cl)
        ref = F.conv2d(up.float(), w.float(), None if bias is None else bias.float(), padding=1)
        assert bool(((got.float() - ref).abs() <= 1.0 * ulp * ref.abs() + 1e-4).all()), float((got.float() - ref).abs().max())
        os.environ["ED_GEMM_ROWS"] = "0"          # (the fused kernel has 256-row tiles only; compare with the same tile height)
        try:
            assert torch.equal(got, ops.conv3x3_nhwc(up, w, bias))
        finally:
            os.environ.pop("ED_GEMM_ROWS", None)
        for _ in range(3):
            assert torch.equal(ops.conv3x3_nhwc_up2x(x, w, bias), got)
    with pytest.raises(RuntimeError):
        ops.conv3x3_nhwc_up2x(x.contiguous(), w, b)        # NCHW memory


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,Cin,N", [(2, 6, 10, 64, 200), (3, 8, 8, 128, 320), (1, 16, 16, 320, 320), (5, 7, 9, 192, 72),
                                         (2, 64, 64, 320, 320), (3, 32, 32, 640, 640)])
def test_downsampler_convolution_stride_2(dtype, B, H, W, Cin, N, any_grid, monkeypatch):
    """ed_conv3x3_nhwc_s2 (round 6): Downsample2D's conv 3x3, stride 2, padding 1 on the MFMA main loop (H, W = the OUTPUT size; output pixel
    (y, x) at tap (dy, dx) reads input (2 y + dy, 2 x + dx)): top / left padding, odd output sizes, batch seams inside a tile, both K loops.
    Within one rounding of fp32 torch; launch-to-launch bit-identical; the module takes it on channels-last activations."""
    from elasticdiffusion_official_amd import models as M, ops
    cl = torch.channels_last
    g = torch.Generator().manual_seed(B * H + Cin + 1)
    x = _asym((B, Cin, 2 * H, 2 * W), g).to(DEV, dtype).contiguous(memory_format=cl)
    w = _asym((N, Cin, 3, 3), g, (9 * Cin) ** -0.5).to(DEV, dtype).contiguous(memory_format=cl)
    b = _asym((N,), g).to(DEV, dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for bias in (b, None):
        got = ops.conv3x3_nhwc_s2(x, w, bias)
        assert got.shape == (B, N, H, W) and got.is_contiguous(memory_format=cl)
        ref = F.conv2d(x.float(), w.float(), None if bias is None else bias.float(), stride=2, padding=1)
        assert bool(((got.float() - ref).abs() <= 1.0 * ulp * ref.abs() + 1e-4).all()), float((got.float() - ref).abs().max())
        for _ in range(3):
            assert torch.equal(ops.conv3x3_nhwc_s2(x, w, bias), got)
    with pytest.raises(RuntimeError):
        ops.conv3x3_nhwc_s2(x[:, :, :-1], w, b)            # odd input height
    if Cin == N:
        monkeypatch.setattr(ops, "gemm_rows_mode", lambda *k: False)     # (the policy keeps under-filled grids with the library; the test wants the kernel)
        monkeypatch.setattr(M, "HIP_DOWNSAMPLE_CONV", True)               # (off by default: a measured tie with the library call)
        down = M.Downsample2D(Cin).to(DEV, dtype).eval().requires_grad_(False).to(memory_format=cl)
        ops.TIMER.start()
        y = down(x)
        assert "ed_conv3x3_nhwc_s2" in set(ops.TIMER.stop())
        ref = F.conv2d(x.float(), down.conv.weight.float(), down.conv.bias.float(), stride=2, padding=1)
        assert bool(((y.float() - ref).abs() <= 1.0 * ulp * ref.abs() + 1e-4).all())


def test_transformer_block_residual_adds_inside_the_projections(monkeypatch):
    """BasicTransformerBlock at the SDXL 640-channel level (where ed_linear runs to_out and the feed-forward's second projection): with
    FUSED_RESIDUAL_LINEAR the three `x = x + branch` adds ride in the projections' epilogues -- no ed_add_layernorm launch, plain
    ed_layernorm instead -- and the block's output stays as close to the fp32 block as the unfused path's."""
    from elasticdiffusion_official_amd import models as M, ops
    torch.manual_seed(11)
    blk = M.BasicTransformerBlock(640, 10, 64, 2048).to(DEV, torch.float16).eval().requires_grad_(False)
    x = torch.randn(20, 4096, 640, device=DEV).half()       # 81920 rows: a grid ed_linear takes (ops.linear_wins)
    ctx = torch.randn(20, 77, 2048, device=DEV).half()
    monkeypatch.setattr(M, "FUSED_RESIDUAL_LINEAR", True)   # (off by default: measured no gain in the forward)

    def run():
        ops.TIMER.start()
        pend, h = blk(x, ctx)
        names = set(ops.TIMER.stop())
        return (h if pend is None else pend + h), names, pend is None

    got, names, summed = run()
    assert summed and "ed_add_layernorm" not in names and "ed_layernorm" in names and "ed_linear" in names, names
    monkeypatch.setattr(M, "FUSED_RESIDUAL_LINEAR", False)
    want, names0, summed0 = run()
    assert not summed0 and "ed_add_layernorm" in names0
    keep_all = M.FUSED_KERNELS
    M.FUSED_KERNELS = False
    try:
        blk.float()
        pend, h = blk(x.float(), ctx.float())
        ref = pend + h
    finally:
        M.FUSED_KERNELS = keep_all
        blk.half()
    e_got = float((got.float() - ref).norm() / ref.norm())
    e_want = float((want.float() - ref).norm() / ref.norm())
    assert e_got < 1.1 * e_want + 1e-5, (e_got, e_want)


def test_upsample2d_module_takes_the_fused_kernel():
    """models.Upsample2D on a channels-last 16-bit activation launches ed_conv3x3_nhwc_up2x (no torch upsample kernel) and equals the
    unfused path bit for bit; the switch restores the latter."""
    from elasticdiffusion_official_amd import models as M, ops
    torch.manual_seed(3)
    cl = torch.channels_last
    up = M.Upsample2D(640).to(DEV, torch.float16).eval().requires_grad_(False).to(memory_format=cl)
    x = torch.randn(6, 640, 32, 32, device=DEV).half().contiguous(memory_format=cl)
    ops.TIMER.start()
    got = up(x)
    launched = set(ops.TIMER.stop())
    assert "ed_conv3x3_nhwc_up2x" in launched, launched
    keep = M.FUSED_UPSAMPLE_CONV
    M.FUSED_UPSAMPLE_CONV = False
    try:
        ops.TIMER.start()
        want = up(x)
        assert "ed_conv3x3_nhwc_up2x" not in set(ops.TIMER.stop())
    finally:
        M.FUSED_UPSAMPLE_CONV = keep
    assert got.shape == (6, 640, 64, 64) and torch.equal(got, want)


@pytest.mark.parametrize("B,H,W,Cin,N", [(40, 32, 32, 1280, 1280), (40, 32, 32, 640, 1280), (80, 32, 32, 320, 1280)])
def test_convolution_batch_split_changes_no_bit(B, H, W, Cin, N, monkeypatch):
    """ops.conv3x3_batch_split (round 6): a batch whose grid ends in a nearly empty round of 256-row tiles runs as two launches split at a sample
    boundary (the tail as 128-row tiles) -- same pixel geometry, same sums in the same order: bit-identical to the single launch, with every
    epilogue operand (bias, per-sample bias, residual) following the split."""
    from elasticdiffusion_official_amd import ops
    cl = torch.channels_last
    BA = ops.conv3x3_batch_split(B, H, W, N)
    assert BA is not None and 0 < BA < B
    g = torch.Generator().manual_seed(B + H)
    x = _asym((B, Cin, H, W), g).to(DEV, torch.float16).contiguous(memory_format=cl)
    w = _asym((N, Cin, 3, 3), g, (9 * Cin) ** -0.5).to(DEV, torch.float16).contiguous(memory_format=cl)
    b = _asym((N,), g).to(DEV, torch.float16)
    sb = _asym((B, N), g).to(DEV, torch.float16)
    r = _asym((B, N, H, W), g).to(DEV, torch.float16).contiguous(memory_format=cl)
    for bias, sbias, res in ((b, None, None), (b, sb, None), (b, None, r), (b, sb, r)):
        monkeypatch.setattr(ops, "CONV_BATCH_SPLIT", True)
        ops.TIMER.start()
        got = ops.conv3x3_nhwc(x, w, bias, sbias, res)
        assert ops.TIMER.stop()["ed_conv3x3_nhwc"][0] == 2          # two launches
        monkeypatch.setattr(ops, "CONV_BATCH_SPLIT", False)
        assert torch.equal(got, ops.conv3x3_nhwc(x, w, bias, sbias, res))
    assert ops.conv3x3_batch_split(20, 32, 32, 1280) is None and ops.conv3x3_batch_split(40, 128, 128, 320) is None
    assert ops.conv3x3_batch_split(12, 64, 64, 640) is None          # 2 full rounds + 64 tiles: measured slower split, stays one launch


def test_gemm_tile_height_is_chosen_by_round_count_and_both_heights_agree(monkeypatch, any_grid):
    """The launcher's own choice (no ED_GEMM_ROWS): 120 full tiles on 256 CUs (the batch-6 forward's 32 x 32 convolutions) run as 240
    128-row tiles.  Whatever it picks, the two tile heights compute the same sums in the same order per output element: bit-identical."""
    from elasticdiffusion_official_amd import ops
    cl = torch.channels_last
    g = torch.Generator().manual_seed(7)
    x = _asym((6, 1280, 32, 32), g).to(DEV, torch.float16).contiguous(memory_format=cl)
    w = _asym((1280, 1280, 3, 3), g, (9 * 1280) ** -0.5).to(DEV, torch.float16).contiguous(memory_format=cl)
    b = _asym((1280,), g).to(DEV, torch.float16)
    monkeypatch.delenv("ED_GEMM_ROWS", raising=False)
    auto = ops.conv3x3_nhwc(x, w, b)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("ED_GEMM_ROWS", mode)
        outs[mode] = ops.conv3x3_nhwc(x, w, b)
    assert torch.equal(outs["0"], outs["1"]) and torch.equal(auto, outs["1"])
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    assert bool(((auto.float() - ref).abs() <= 2.0 ** -11 * ref.abs() + 1e-4).all())
    xl, wl = _asym((3072, 1280), g).to(DEV, torch.float16), _asym((1280, 1280), g, 1280 ** -0.5).to(DEV, torch.float16)
    for mode in ("0", "1"):
        monkeypatch.setenv("ED_GEMM_ROWS", mode)
        outs[mode] = ops.linear(xl, wl, b)
    assert torch.equal(outs["0"], outs["1"])


def test_gemm_wrappers_refuse_what_the_kernel_does_not_take():
    from elasticdiffusion_official_amd import ops
    x = torch.zeros(512, 96, device=DEV, dtype=torch.float16)       # K % 64 != 0
    w = torch.zeros(256, 96, device=DEV, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        ops.geglu_gemm(x, w)
    with pytest.raises(RuntimeError):
        ops.linear(x, w)
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(512, 128), torch.zeros(256, 128))   # CPU tensors: no CPU fallback
    # where the model uses them (measured policy, profiles/r4_s2_probe_gemm_*.jsonl)
    assert not ops.linear_wins(20480, 1280, 3840) and ops.linear_wins(81920, 640, 1920) and ops.linear_wins(81920, 2560, 640)
    assert not ops.linear_wins(24576, 640, 640) and ops.linear_wins(24576, 640, 1920)      # 288 tiles: 2 rounds, 56 % full
    assert not ops.conv3x3_ok(20, 128, 128, 4, 320) and ops.conv3x3_wins(20, 32, 32, 1280, 1280) and ops.conv3x3_wins(6, 32, 32, 1280, 1280)
    assert ops.conv3x3_wins(6, 64, 64, 640, 640) and not ops.conv3x3_wins(20, 8, 8, 1280, 1280) and ops.conv3x3_ok(20, 8, 8, 1280, 1280)
    # round 6: under-filled grids run as 128-row tiles (profiles/r6_s4_gemm_rows_mode.jsonl) -- the 1- / 3-row per-rank forwards' shapes
    assert ops.gemm_rows_mode(6144, 5) and not ops.gemm_rows_mode(24576, 3) and not ops.gemm_rows_mode(20480, 5)
    assert ops.conv3x3_wins(3, 32, 32, 1280, 1280) and ops.conv3x3_wins(1, 64, 64, 640, 640)
    # projections: only the batch-6 grids (240 half tiles); the 3- / 1-row forwards' 40-120-tile grids won alone and lost in the forward
    # (profiles/r6_s7_policy_split.jsonl)
    assert ops.linear_wins(6144, 1280, 1280) and not ops.linear_wins(3072, 1280, 1280) and not ops.linear_wins(3072, 5120, 1280)
    assert not ops.linear_wins(4096, 640, 640) and not ops.linear_wins(1024, 1280, 1280)
    assert not ops.linear_wins(12288, 2560, 640) and not ops.linear_wins(20480, 1280, 1280)
