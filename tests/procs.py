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


PY = sys.executable
