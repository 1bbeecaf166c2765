"""Round 6 (r5_pmc_kernels.py + the 64 x 64 x 640 convolution, the batch-6 32 x 32 convolution and the channels-last GroupNorm at three shapes).
Workload for `rocprofv3 --pmc` passes over this repo's MFMA kernels, at the headline workload's
largest shapes (batch 20): the default attention variant (v_path 5, loads / LDS writes inside the MFMA region) at N = 4096 and 1024, the
persistent GEGLU GEMM, ed_linear, ed_conv3x3_nhwc (long-K loop), and the fp32 VAE's split-operand path at the pad-strip encode's first
level ([5, 128, 256, 1024]: ed_groupnorm_nhwc_f32 -> ed_conv3x3_nhwc_f32out).  Each launch mix is repeated; tools/pmc_by_kernel.py
averages the later half per (kernel, grid)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from elasticdiffusion_official_amd import ops

g = torch.Generator(device="cuda").manual_seed(0)
dt = torch.float16
cl = torch.channels_last
with torch.no_grad():
    for (H, N) in ((10, 4096), (20, 1024)):
        qkv = torch.randn(20, N, 3 * H * 64, device="cuda", generator=g).to(dt)
        q, k, v = qkv[..., :H * 64], qkv[..., H * 64:2 * H * 64], qkv[..., 2 * H * 64:]
        for _ in range(6):
            ops.flash_attention(q, k, v, H, v_path=5)
    for (M, K, I) in ((20480, 1280, 5120), (81920, 640, 2560)):
        x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(dt)
        w = ((torch.rand(2 * I, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).to(dt)
        b = (torch.rand(2 * I, device="cuda", generator=g) * 2 - 1).to(dt)
        for _ in range(6):
            ops.geglu_gemm(x, w, b)
    for (M, K, N) in ((81920, 640, 1920), (81920, 2560, 640)):
        x = (torch.rand(M, K, device="cuda", generator=g) * 2 - 1).to(dt)
        w = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) / K ** 0.5).to(dt)
        for _ in range(6):
            ops.linear(x, w, None)
    for (B, Hh, W, Cin, N) in ((20, 32, 32, 1280, 1280), (20, 128, 128, 320, 320), (20, 64, 64, 640, 640), (6, 32, 32, 1280, 1280)):
        x = (torch.rand(B, Cin, Hh, W, device="cuda", generator=g) * 2 - 1).to(dt).contiguous(memory_format=cl)
        w = ((torch.rand(N, Cin, 3, 3, device="cuda", generator=g) * 2 - 1) / (9 * Cin) ** 0.5).to(dt).contiguous(memory_format=cl)
        b = (torch.rand(N, device="cuda", generator=g) * 2 - 1).to(dt)
        for _ in range(6):
            ops.conv3x3_nhwc(x, w, b)
    # round 6: the upsampler convolution that gathers its A operand from the low-resolution source, the 77-key cross attention (output rows as
    # dwordx4), GroupNorm of a concatenation that is never written
    x = (torch.rand(20, 1280, 32, 32, device="cuda", generator=g) * 2 - 1).to(dt).contiguous(memory_format=cl)
    w = ((torch.rand(1280, 1280, 3, 3, device="cuda", generator=g) * 2 - 1) / (9 * 1280) ** 0.5).to(dt).contiguous(memory_format=cl)
    b = (torch.rand(1280, device="cuda", generator=g) * 2 - 1).to(dt)
    for _ in range(6):
        ops.conv3x3_nhwc_up2x(x, w, b)
    q = torch.randn(20, 4096, 640, device="cuda", generator=g).to(dt)
    k, v = (torch.randn(20, 77, 640, device="cuda", generator=g).to(dt) for _ in range(2))
    for _ in range(6):
        ops.flash_attention(q, k, v, 10)
    x1 = torch.randn(20, 1280, 32, 32, device="cuda", generator=g).to(dt).contiguous(memory_format=cl)
    x2 = torch.randn(20, 640, 32, 32, device="cuda", generator=g).to(dt).contiguous(memory_format=cl)
    gw, gb = torch.ones(1920, device="cuda", dtype=dt), torch.zeros(1920, device="cuda", dtype=dt)
    for _ in range(6):
        ops.groupnorm_nhwc_cat(x1, x2, gw, gb, 32, 1e-5, silu=True)
    for shape in ((20, 320, 128, 128), (20, 640, 64, 64), (20, 1280, 32, 32)):
        x = torch.randn(*shape, device="cuda", generator=g).to(dt).contiguous(memory_format=cl)
        gw, gb = torch.ones(shape[1], device="cuda", dtype=dt), torch.zeros(shape[1], device="cuda", dtype=dt)
        for _ in range(6):
            ops.groupnorm_nhwc(x, gw, gb, 32, 1e-5, silu=True)
    # the VAE encoder's first level on a 5-strip batch
    x32 = torch.randn(5, 128, 256, 1024, device="cuda", generator=g).contiguous(memory_format=cl)
    gamma, beta = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    w32 = (torch.rand(128, 128, 3, 3, device="cuda", generator=g) * 2 - 1) / (9 * 128) ** 0.5
    ws, sc = ops.split_conv_weight(w32)
    bias = torch.zeros(128, device="cuda")
    for _ in range(6):
        a = ops.groupnorm_nhwc_f32(x32, gamma, beta, 32, 1e-6, silu=True, split=True)
        ops.conv3x3_f32out(a, ws, bias, x32, sc)
torch.cuda.synchronize()
print("done")
