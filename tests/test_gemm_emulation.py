"""CPU: the 8-phase GEMM kernel (csrc/gemm_kernels.hip) keeps compiling for gfx950 within its register / LDS budget, and its
lane-level CPU replay (tools/emulate_gemm_kernel.py: every address, the LDS-DMA / fragment-read hazard intervals under the two
adversarial timings) stays exact -- the check that let the kernel run correctly on its first GPU launch."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gemm_kernel_replay_is_exact_and_catches_broken_schedules():
    from tests import replays     # one background pool for every replay of the suite (tests/replays.py)
    ok = replays.result("gemm_quick")
    assert ok.returncode == 0 and "WRONG" not in ok.stdout, ok.stdout + ok.stderr
    for brk in ("war", "raw", "lgkm", "early"):   # early: the first K tile's counted waits one DMA pair too weak
        out = replays.result("gemm_quick_" + brk)
        assert out.returncode == 0 and "caught the deliberately broken schedule" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_gemm_kernels_compile_for_gfx950_without_spills(tmp_path):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-c",
                          os.path.join(ROOT, "elasticdiffusion_official_amd", "csrc", "gemm_kernels.hip"), "-o", str(tmp_path / "gemm.o"),
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=str(tmp_path), timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    rep = out.stderr
    # bf16 / f16 x (with, without epilogue addends) x (plain projection | 3x3 convolution x (8-phase, long-K loop)) = 4 + 8, + the f16
    # convolution's fp32-output epilogue (the VAE's split operands) x addends x the two loops = 4, + round 6's 128-row tile mode for the
    # plain projection and the convolution x dtype x addends = 8, + the upsampler (CONV = 2: A operand gathered from the low-resolution source)
    # and the stride-2 downsampler (CONV = 3) convolutions, bias only, x dtype x the two loops = 8, + the VAE encoder's pad-(0, 1, 0, 1) stride-2
    # convolution on split operands (CONV = 4, fp32 epilogue, f16 only) x the two loops = 2; GEGLU runs the persistent loop
    assert len(re.findall(r"Function Name: .*k_gemm_8phase", rep)) == 34
    assert len(re.findall(r"Function Name: .*k_geglu_persist", rep)) == 2
    assert set(re.findall(r"VGPRs Spill: (\d+)", rep)) == {"0"} and set(re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", rep)) == {"0"}
    assert all(int(v) <= 256 for v in re.findall(r" VGPRs: (\d+)", rep))         # 2 waves per SIMD
    assert set(re.findall(r"LDS Size \[bytes/block\]: (\d+)", rep)) == {"131072"}
