"""The fp32 VAE's ResnetBlock convolutions on split fp16 operands (csrc/vae_kernels.hip, the GEMM kernel's fp32-output epilogue;
models.VAE_SPLIT_CONV): the reference runs the VAE in fp32 (elastic_diffusion.py:267-310, :327-364 -- the encoder explicitly outside
autocast, :328), so this path is held to fp32 accuracy, not to a 16-bit bar.

CPU: the operand algebra ([xh | xl | xh] against [wh | wh | wl], power-of-two pre-scaling) reproduces the fp64 convolution to < 2e-7.
-m gpu, through the C ABI: GroupNorm(+SiLU) -> split operand and the fp32-output convolution against fp64 torch, as accurate as torch's
own fp32 ops; ragged shapes, half / full column tiles, image borders inside a tile; launch-to-launch bit identity; the full-width VAE
with the switch on against the switch off (MIOpen fp32)."""
import pytest
import torch
import torch.nn.functional as F


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def test_split_operand_algebra_on_the_cpu():
    from elasticdiffusion_official_amd import ops
    g = torch.Generator().manual_seed(0)
    for scale in (1e-3, 0.04, 7.0):           # weights far below / around / above 1: the pre-scaling must keep wl normal
        x = torch.randn(2, 64, 9, 11, generator=g) * 3
        w = (torch.rand(40, 64, 3, 3, generator=g) * 2 - 1) * scale
        ws, sc = ops.split_conv_weight(w)
        assert ws.dtype == torch.float16 and tuple(ws.shape) == (40, 192, 3, 3) and ws.is_contiguous(memory_format=torch.channels_last)
        assert 8192 <= float(ws.float().abs().max()) < 16385 and sc == 2.0 ** round(__import__("math").log2(sc))
        hi = x.half()
        lo = (x - hi.float()).half()
        got = F.conv2d(torch.cat([hi, lo, hi], 1).double(), ws.double(), padding=1) * sc
        want = F.conv2d(x.double(), w.double(), padding=1)
        assert _rel(got, want) < 2e-7
        assert _rel(F.conv2d(hi.double(), w.half().double(), padding=1), want) > 1e-4     # what one fp16 pass would cost


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W,groups,silu", [(2, 128, 24, 40, 32, True), (1, 256, 17, 9, 32, True), (3, 512, 8, 8, 32, False),
                                                 (1, 128, 128, 256, 32, True), (2, 64, 5, 7, 16, True)])
def test_groupnorm_nhwc_f32_plain_and_split(N, C, H, W, groups, silu):
    from elasticdiffusion_official_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x = (torch.randn(N, C, H, W, generator=g) * 2.5 + 0.7).to(dev).contiguous(memory_format=torch.channels_last)
    gamma, beta = (torch.randn(C, generator=g) * 0.5 + 1).to(dev), (torch.randn(C, generator=g) * 0.3).to(dev)
    want = F.group_norm(x.double(), groups, gamma.double(), beta.double(), 1e-6)
    if silu:
        want = F.silu(want)
    ref32 = F.group_norm(x, groups, gamma, beta, 1e-6)
    ref32 = F.silu(ref32) if silu else ref32
    y = ops.groupnorm_nhwc_f32(x, gamma, beta, groups, 1e-6, silu=silu)
    assert y.dtype == torch.float32 and y.is_contiguous(memory_format=torch.channels_last) and y.shape == x.shape
    assert _rel(y, want) <= max(2.0 * _rel(ref32, want), 3e-7)
    s = ops.groupnorm_nhwc_f32(x, gamma, beta, groups, 1e-6, silu=silu, split=True)
    assert s.dtype == torch.float16 and tuple(s.shape) == (N, 3 * C, H, W) and s.is_contiguous(memory_format=torch.channels_last)
    hi, lo, hi2 = s[:, :C], s[:, C:2 * C], s[:, 2 * C:]
    assert torch.equal(hi, hi2) and torch.equal(hi, y.half())
    # hi + lo carries y to 2^-22 of its magnitude (lo in fp16's subnormal range for tiny y: absolute 2^-25)
    err = (hi.double() + lo.double() - y.double()).abs()
    assert bool((err <= y.double().abs() * 2.0 ** -21 + 2.0 ** -24).all())
    assert torch.equal(s, ops.groupnorm_nhwc_f32(x, gamma, beta, groups, 1e-6, silu=silu, split=True))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,Cin,N,bias,res", [(2, 24, 40, 128, 128, True, True), (1, 17, 9, 128, 256, True, False),
                                                  (2, 16, 16, 256, 512, False, True), (1, 33, 20, 64, 104, True, True),
                                                  (1, 128, 256, 128, 128, True, True), (3, 8, 8, 512, 512, True, True)])
def test_conv3x3_f32out_on_split_operands_is_fp32_accurate(B, H, W, Cin, N, bias, res):
    from elasticdiffusion_official_amd import ops
    dev = "cuda:0"
    cl = torch.channels_last
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + H)
    x = F.silu(torch.randn(B, Cin, H, W, generator=g) * 2).to(dev).contiguous(memory_format=cl)
    w = ((torch.rand(N, Cin, 3, 3, generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev) if bias else None
    r = torch.randn(B, N, H, W, generator=g).to(dev).contiguous(memory_format=cl) if res else None
    hi = x.half()
    lo = (x - hi.float()).half()
    a = torch.cat([hi, lo, hi], 1).contiguous(memory_format=cl)
    ws, sc = ops.split_conv_weight(w)
    want = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if r is not None:
        want = want + r.double()
    prev = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False          # an independent fp32 arithmetic (im2col + rocBLAS), not MIOpen
    try:
        ref32 = F.conv2d(x.contiguous(), w, b, padding=1)
    finally:
        torch.backends.cudnn.enabled = prev
    if r is not None:
        ref32 = ref32 + r
    got = ops.conv3x3_f32out(a, ws, b, r, sc)
    assert got.dtype == torch.float32 and got.is_contiguous(memory_format=cl) and tuple(got.shape) == (B, N, H, W)
    e, e32 = _rel(got, want), _rel(ref32, want)
    assert e <= max(2.0 * e32, 6e-7), (e, e32)
    for _ in range(3):                             # the LDS race screen: launch-to-launch bit identity
        assert torch.equal(got, ops.conv3x3_f32out(a, ws, b, r, sc))


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout", [(128, 128), (128, 256), (512, 512)])
def test_vae_resnet_block_split_vs_library(cin, cout):
    from elasticdiffusion_official_amd import models as M
    dev = "cuda:0"
    torch.manual_seed(cin + cout)
    blk = M.ResnetBlock2D(cin, cout, None, eps=1e-6).to(dev).eval().requires_grad_(False)
    x = torch.randn(2, cin, 24, 40, device=dev) * 1.5
    ref64 = M.ResnetBlock2D(cin, cout, None, eps=1e-6).to(dev).double().eval().requires_grad_(False)
    ref64.load_state_dict({k: v.double() for k, v in blk.state_dict().items()})
    saved = M.VAE_SPLIT_CONV
    try:
        M.VAE_SPLIT_CONV = False
        want64 = ref64(x.double())
        lib = blk(x)
        M.VAE_SPLIT_CONV = True
        got = blk(x.contiguous(memory_format=torch.channels_last))
    finally:
        M.VAE_SPLIT_CONV = saved
    assert got.is_contiguous(memory_format=torch.channels_last) and got.dtype == torch.float32
    e, elib = _rel(got, want64), _rel(lib, want64)
    assert e <= max(2.0 * elib, 1e-6), (e, elib)


@pytest.mark.gpu
def test_full_width_vae_split_vs_library_encode_and_decode():
    """The real AutoencoderKL widths (128, 256, 512, 512): encode a 5-strip batch and decode a latent tile with the switch on
    (ResnetBlock / upsampler convolutions on the MFMA pipe) and off (MIOpen fp32): rel-L2 < 5e-5 between the two fp32 paths; the split
    path must actually have launched."""
    from elasticdiffusion_official_amd import models as M, ops
    dev = "cuda:0"
    with torch.device("meta"):
        vae = M.AutoencoderKL(scaling_factor=0.13025, force_upcast=True)
    vae = vae.to_empty(device=dev)
    M._seeded_init(vae, 1)
    vae = vae.eval().requires_grad_(False)
    g = torch.Generator(device="cpu").manual_seed(3)
    img = (torch.rand(5, 3, 64, 256, generator=g) * 2 - 1).to(dev)
    z = torch.randn(2, 4, 24, 32, generator=g).to(dev)
    saved = M.VAE_SPLIT_CONV
    try:
        M.VAE_SPLIT_CONV = False
        with torch.no_grad():
            enc0, dec0 = vae.encode(img).latent_dist.mean, vae.decode(z).sample
        M.VAE_SPLIT_CONV = True
        ops.TIMER.start()
        with torch.no_grad():
            enc1, dec1 = vae.encode(img).latent_dist.mean, vae.decode(z).sample
        kt = ops.TIMER.stop()
    finally:
        M.VAE_SPLIT_CONV = saved
    assert kt.get("ed_conv3x3_nhwc_f32out", (0,))[0] >= 40 and kt.get("ed_groupnorm_nhwc_f32", (0,))[0] >= 40, kt
    assert dec1.is_contiguous() and dec1.shape == dec0.shape and enc1.stride() == enc0.stride()
    # two fp32 implementations of a 30-layer network: measured 7e-6 ... 2.1e-5 between the paths over the workloads' shapes
    # (profiles/r5_s4_vae_split_with_upsampler_vs_library.jsonl); the per-op tests above hold each path to fp64
    assert _rel(enc1, enc0) < 5e-5 and _rel(dec1, dec0) < 5e-5, (_rel(enc1, enc0), _rel(dec1, dec0))


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,H,W,up", [(2, 128, 12, 20, True), (1, 256, 9, 7, False), (3, 64, 16, 16, True)])
def test_split_f32_raw_stream_with_saturating_hi_and_upsampling(N, C, H, W, up):
    from elasticdiffusion_official_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(N * 100 + C)
    x = torch.randn(N, C, H, W, generator=g) * 30
    x[0, :8, 0, 0] = torch.tensor([7.0e4, -9.9e4, 65504.0, -65504.0, 1.2e5, 3e-6, -2e-7, 0.0])   # beyond fp16's range, tiny, zero
    x = x.to(dev).contiguous(memory_format=torch.channels_last)
    s = ops.split_f32(x, upsample2x=up)
    u = 2 if up else 1
    assert s.dtype == torch.float16 and tuple(s.shape) == (N, 3 * C, u * H, u * W) and s.is_contiguous(memory_format=torch.channels_last)
    assert bool(torch.isfinite(s.float()).all())
    hi, lo, hi2 = s[:, :C], s[:, C:2 * C], s[:, 2 * C:]
    assert torch.equal(hi, hi2)
    want = F.interpolate(x, scale_factor=2.0, mode="nearest") if up else x
    rec = hi.double() + lo.double()
    err = (rec - want.double()).abs()
    small = want.abs() <= 65504
    assert bool((err[small] <= want.double().abs()[small] * 2.0 ** -21 + 2.0 ** -24).all())
    assert bool((err[~small] <= want.double().abs()[~small] * 2.0 ** -10).all())     # above the range: lo's 11 bits carry the excess
    assert torch.equal(hi[small], want.clamp(-65504, 65504).half()[small])


@pytest.mark.gpu
def test_vae_upsampler_split_vs_library():
    from elasticdiffusion_official_amd import models as M
    dev = "cuda:0"
    torch.manual_seed(5)
    up = M.Upsample2D(128, vae=True).to(dev).eval().requires_grad_(False)
    x = torch.randn(3, 128, 20, 28, device=dev) * 4
    want = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), up.conv.weight.double(), up.conv.bias.double(), padding=1)
    saved = M.VAE_SPLIT_CONV
    try:
        M.VAE_SPLIT_CONV = False
        lib = up(x)
        M.VAE_SPLIT_CONV = True
        got = up(x)
    finally:
        M.VAE_SPLIT_CONV = saved
    assert got.shape == lib.shape and got.is_contiguous(memory_format=torch.channels_last)
    e, elib = _rel(got, want), _rel(lib, want)
    assert e <= max(2.0 * elib, 6e-7), (e, elib)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,cout,H,W", [(3, 128, 128, 20, 28), (1, 256, 256, 18, 6), (2, 512, 512, 8, 8), (5, 128, 128, 64, 256)])
def test_vae_downsampler_split_vs_library(B, C, cout, H, W):
    """The VAE encoder's Downsample2D -- F.pad(x, (0, 1, 0, 1)) + conv 3x3, stride 2 -- on the split-operand main loop (round 6,
    ed_conv3x3_nhwc_f32out_s2: template value CONV = 4; bottom / right taps past the edge read zeros): as accurate against fp64 as the
    library's fp32 convolution, channels-last out, launch-to-launch bit-identical; values beyond fp16's range go through the absmax scale."""
    from elasticdiffusion_official_amd import models as M, ops
    dev = "cuda:0"
    torch.manual_seed(B + C)
    down = M.Downsample2D(C, padding=0).to(dev).eval().requires_grad_(False)
    x = torch.randn(B, C, H, W, device=dev) * 4
    x[0, :, H // 2, W // 2] *= 3.0e4          # a stream that leaves fp16's range (the real VAE's does)
    want = F.conv2d(F.pad(x.double(), (0, 1, 0, 1)), down.conv.weight.double(), down.conv.bias.double(), stride=2)
    saved = M.VAE_SPLIT_DOWNSAMPLE
    try:
        M.VAE_SPLIT_DOWNSAMPLE = False
        lib = down(x)
        M.VAE_SPLIT_DOWNSAMPLE = True
        ops.TIMER.start()
        got = down(x)
        assert "ed_conv3x3_nhwc_f32out_s2" in ops.TIMER.stop()
        again = down(x)
    finally:
        M.VAE_SPLIT_DOWNSAMPLE = saved
    assert got.shape == lib.shape == (B, cout, H // 2, W // 2) and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, again)
    e, elib = _rel(got, want), _rel(lib, want)
    assert e <= max(2.0 * elib, 6e-7), (e, elib)


# ---- round 6: the raw-stream split is exact over the WHOLE fp32 range ----------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("peak", [3.0e4, 7.0e4, 1.4e5, 2.5e6, 3.0e9, 1.0e30])
def test_split_f32_with_absmax_scale_is_exact_beyond_fp16_range(peak):
    """|v| far beyond 1.3e5 (where round 5's saturating hi ran out of bits, and beyond 131 008 produced inf - inf): with the per-tensor
    power-of-two scale hi + lo = 2^-e v to 2^-22 for every element, nothing saturates, and the convolution's 2^e brings the result back:
    finite and as accurate as the library's fp32 convolution against fp64."""
    import math
    from elasticdiffusion_official_amd import ops
    dev = "cuda:0"
    cl = torch.channels_last
    g = torch.Generator(device="cpu").manual_seed(int(math.log2(peak)))
    N, C, H, W, Nout = 2, 128, 10, 14, 128
    x = torch.randn(N, C, H, W, generator=g) * (peak / 6)
    x[0, 3, 2, 5], x[1, 7, 9, 13] = peak, -peak * 0.75
    x = x.to(dev).contiguous(memory_format=cl)
    am = ops.absmax_f32(x)
    assert float(am) == float(x.abs().max()) and torch.equal(am, ops.absmax_f32(x))
    e = max(0, math.frexp(float(am))[1] - 1 - 14)
    s = ops.split_f32(x, upsample2x=True, absmax=am)
    assert bool(torch.isfinite(s.float()).all())
    hi, lo = s[:, :C].double(), s[:, C:2 * C].double()
    want = F.interpolate(x, scale_factor=2.0, mode="nearest").double() * 2.0 ** -e
    assert float(hi.abs().max()) < 32768.0 * 1.0001                      # hi never reaches the clamp
    assert bool(((hi + lo - want).abs() <= want.abs() * 2.0 ** -21 + 2.0 ** -24).all())
    w = ((torch.rand(Nout, C, 3, 3, generator=g) * 2 - 1) / (9 * C) ** 0.5).to(dev)
    b = (torch.randn(Nout, generator=g) * peak * 0.01).to(dev)
    ws, sc = ops.split_conv_weight(w)
    got = ops.conv3x3_f32out(s, ws, b, None, sc, act_absmax=am)
    ref64 = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), b.double(), padding=1)
    prev = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False
    try:
        ref32 = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest").contiguous(), w, b, padding=1)
    finally:
        torch.backends.cudnn.enabled = prev
    assert bool(torch.isfinite(got).all())
    e_got, e_lib = _rel(got, ref64), _rel(ref32, ref64)
    assert e_got <= max(2.0 * e_lib, 6e-7), (peak, e_got, e_lib)
    assert torch.equal(got, ops.conv3x3_f32out(s, ws, b, None, sc, act_absmax=am))


@pytest.mark.gpu
def test_vae_upsampler_on_a_stream_beyond_fp16_range_matches_the_library():
    """The product path itself (models.Upsample2D, vae=True): the real SDXL decoder stream exceeds fp16's range; the split path must stay
    finite and fp32-accurate there."""
    from elasticdiffusion_official_amd import models as M
    dev = "cuda:0"
    torch.manual_seed(11)
    up = M.Upsample2D(128, vae=True).to(dev).eval().requires_grad_(False)
    x = torch.randn(2, 128, 16, 24, device=dev) * 5.0e4
    x[0, 0, 0, 0], x[1, 5, 3, 3] = 4.0e5, -2.9e5
    want = F.conv2d(F.interpolate(x.double(), scale_factor=2.0, mode="nearest"), up.conv.weight.double(), up.conv.bias.double(), padding=1)
    saved = M.VAE_SPLIT_CONV
    try:
        M.VAE_SPLIT_CONV = False
        lib = up(x)
        M.VAE_SPLIT_CONV = True
        got = up(x)
    finally:
        M.VAE_SPLIT_CONV = saved
    assert bool(torch.isfinite(got).all())
    e, elib = _rel(got, want), _rel(lib, want)
    assert e <= max(2.0 * elib, 6e-7), (e, elib)


@pytest.mark.gpu
def test_groupnorm_split_output_saturates_instead_of_overflowing():
    """GroupNorm + SiLU bounds its output by the affine parameters; a pathological gamma must still give a finite operand (hi and lo
    clamp at +-65504) instead of hi = inf, lo = -inf."""
    from elasticdiffusion_official_amd import ops
    dev = "cuda:0"
    C, G = 128, 32
    x = torch.randn(1, C, 8, 8, device=dev).contiguous(memory_format=torch.channels_last)
    gamma = torch.full((C,), 1.0e5, device=dev)
    beta = torch.zeros(C, device=dev)
    s = ops.groupnorm_nhwc_f32(x, gamma, beta, G, 1e-6, silu=False, split=True)
    assert bool(torch.isfinite(s.float()).all())
    y = ops.groupnorm_nhwc_f32(x, gamma, beta, G, 1e-6, silu=False, split=False)
    hi, lo = s[:, :C].double(), s[:, C:2 * C].double()
    inside = y.abs() <= 65504
    assert bool(((hi + lo - y.double()).abs()[inside] <= y.double().abs()[inside] * 2.0 ** -21 + 2.0 ** -24).all())
    assert bool(((hi + lo).abs()[~inside] >= 65504).all())


def test_split_weight_cache_key_tells_a_replaced_parameter_apart():
    """a fresh Parameter at a recycled address (same data_ptr, version 0) must not hit the cached split weights."""
    from elasticdiffusion_official_amd import models as M
    conv = torch.nn.Conv2d(64, 64, 3, padding=1)
    w0 = conv.weight
    k0 = (id(w0), w0.data_ptr(), w0._version, str(w0.device))
    conv.weight = torch.nn.Parameter(w0.detach().clone())
    w1 = conv.weight
    assert (id(w1), w1.data_ptr(), w1._version, str(w1.device)) != k0
    import inspect
    assert "id(w)" in inspect.getsource(M._split_weight)
    assert M.prepare_vae_split(torch.nn.Sequential(conv)) == 0     # CPU weights: nothing to pre-split, no error
