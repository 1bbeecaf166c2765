"""CPU: the not-yet-validated kernels under experiments/ keep compiling for gfx950 and keep passing their CPU replays.
Nothing here (or anywhere in the product) launches them; see experiments/README.md."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "experiments", "geglu_gemm")


def test_geglu_gemm_replay_is_exact_and_catches_broken_schedules():
    run = lambda *a: subprocess.run([sys.executable, os.path.join(EXP, "emulate_geglu_gemm.py"), "--quick", *a],
                                    capture_output=True, text=True, cwd=ROOT)
    ok = run()
    assert ok.returncode == 0 and "WRONG" not in ok.stdout, ok.stdout + ok.stderr
    for brk in ("war", "raw", "lgkm"):
        out = run("--break", brk)
        assert out.returncode == 0 and "caught the deliberately broken schedule" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_geglu_gemm_compiles_for_gfx950_without_spills(tmp_path):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                          os.path.join(EXP, "geglu_gemm.hip"), "-o", str(tmp_path / "libgeglu_gemm.so"),
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    rep = out.stderr
    assert len(re.findall(r"Function Name: .*k_geglu_gemm", rep)) == 12         # bf16 / f16 x GEGLU / plain / 3x3 convolution x (normal, SAFE diagnosis build)
    assert set(re.findall(r"VGPRs Spill: (\d+)", rep)) == {"0"} and set(re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", rep)) == {"0"}
    assert all(int(v) <= 256 for v in re.findall(r" VGPRs: (\d+)", rep))         # 2 waves per SIMD
    assert set(re.findall(r"LDS Size \[bytes/block\]: (\d+)", rep)) == {"131072"}
