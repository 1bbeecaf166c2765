"""Process helpers for the multi-process tests: every child is joined with a deadline, and a child that is still alive after it is
terminated (then killed) before the test fails -- a hang becomes a named failure instead of a pytest run that never ends (a live
non-daemon child would otherwise be joined forever at interpreter exit).  VERDICT r4, "possible flaky hang in the CPU suite"."""
import subprocess
import sys


def join_all(procs, timeout, what="worker"):
    """Join every process within ``timeout`` seconds overall-per-process; kill stragglers; assert all exited with 0."""
    hung = []
    for i, p in enumerate(procs):
        p.join(timeout)
        if p.is_alive():
            hung.append(i)
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(10)
            if p.is_alive():
                p.kill()
                p.join(10)
    assert not hung, f"{what}: rank(s) {hung} still running after {timeout} s (terminated)"
    codes = [p.exitcode for p in procs]
    assert all(c == 0 for c in codes), f"{what}: exit codes {codes}"


def run(cmd, timeout, **kw):
    """subprocess.run(capture_output, text) with a deadline; on expiry the child is killed and the test fails by name."""
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, **kw)
    except subprocess.TimeoutExpired as e:
        tail = (e.stdout or b"")[-1500:] if isinstance(e.stdout, (bytes, bytearray)) else (e.stdout or "")[-1500:]
        raise AssertionError(f"{' '.join(map(str, cmd))[:200]} did not finish within {timeout} s; stdout tail: {tail!r}") from None


def single_thread():
    """First call of every spawned worker: ONE intra-op thread.  N workers x (all cores) OpenMP threads that spin at their barriers on
    an 8-core box is the one way these tiny CPU workloads can take minutes instead of milliseconds (the CPU suite was once seen to take
    1 200 s instead of 200 s with nothing failing -- cf. VERDICT r4's run that "spun at 450 % CPU")."""
    import os
    os.environ["OMP_NUM_THREADS"] = "1"
    import torch
    torch.set_num_threads(1)


PY = sys.executable
