"""The lane-level CPU replays (tools/emulate_*.py) as ONE pool of background subprocesses: the first test that asks for a result starts
all of them at once (15 independent single-threaded numpy scripts, ~170 s of CPU work back to back, ~35 s side by side on the 8-core
builder container), every later test only collects its own.  Each job has a deadline and is killed on expiry (tests/procs.py's rule:
a hang is a named failure).  VERDICT r5 item 8: the CPU suite has to stay a few-minute check."""
import os
import subprocess
import sys
import tempfile
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

JOBS = {
    "gemm_quick": ("emulate_gemm_kernel.py", "--quick"),
    "gemm_quick_war": ("emulate_gemm_kernel.py", "--quick", "--break", "war"),
    "gemm_quick_raw": ("emulate_gemm_kernel.py", "--quick", "--break", "raw"),
    "gemm_quick_lgkm": ("emulate_gemm_kernel.py", "--quick", "--break", "lgkm"),
    "gemm_quick_early": ("emulate_gemm_kernel.py", "--quick", "--break", "early"),
    "persist2": ("emulate_gemm_kernel.py", "--persist2"),
    "break_pf": ("emulate_gemm_kernel.py", "--break", "pf"),
    "sched_two_read": ("emulate_gemm_kernel.py", "--sched", "two_read"),
    "sched_r4": ("emulate_gemm_kernel.py", "--sched", "r4"),
    "sched_bad_early_b": ("emulate_gemm_kernel.py", "--sched", "bad_early_b"),
    "half": ("emulate_gemm_kernel.py", "--half"),
    "break_half_raw": ("emulate_gemm_kernel.py", "--break", "half_raw"),
    "rows": ("emulate_gemm_kernel.py", "--rows"),
    "break_rows_raw": ("emulate_gemm_kernel.py", "--break", "rows_raw"),
    "flash_attention": ("emulate_flash_attention.py",),
}
_RUNNING = {}
_DONE = {}
_T0 = [0.0]


def _start_all():
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    _T0[0] = time.monotonic()
    for name, (script, *args) in JOBS.items():
        out, err = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", script), *args], stdout=out, stderr=err, cwd=ROOT, env=env)
        _RUNNING[name] = (p, out, err)


def result(name, timeout=900):
    """-> SimpleNamespace(returncode, stdout, stderr) of replay ``name`` (see JOBS); ``timeout`` counts from the pool's start."""
    if name in _DONE:
        return _DONE[name]
    if not _RUNNING:
        _start_all()
    p, out, err = _RUNNING[name]
    left = max(1.0, timeout - (time.monotonic() - _T0[0]))
    try:
        p.wait(left)
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait(10)
        out.seek(0)
        raise AssertionError(f"replay {name} ({' '.join(JOBS[name])}) did not finish within {timeout} s; stdout tail: {out.read()[-1500:]!r}") from None
    out.seek(0), err.seek(0)
    _DONE[name] = SimpleNamespace(returncode=p.returncode, stdout=out.read(), stderr=err.read())
    out.close(), err.close()
    return _DONE[name]


def kill_all():
    for p, _, _ in _RUNNING.values():
        if p.poll() is None:
            p.kill()
