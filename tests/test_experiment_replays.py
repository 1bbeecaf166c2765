"""CPU: the lane-level replays that stood behind the round-4/5 experiments (tools/gemm_persist, tools/gemm_sched, tools/attn16 -- folded into the
product in round 5 and removed; git history up to 24deb02) keep
passing -- they are what let those kernels run correctly on their first GPU launch, and what the patches in tools/r5_patches cite."""
from tests import replays

_run = replays.result      # every replay of the suite runs in ONE background pool (tests/replays.py)


def test_persistent_gemm_prefetch_replay_and_its_broken_variant():
    ok = _run("persist2")            # next tile's K tile 0 staged during the last K tile
    assert ok.returncode == 0 and "WRONG" not in ok.stdout and ok.stdout.count("exact") >= 12, ok.stdout + ok.stderr
    bad = _run("break_pf")         # first prefetch aimed at the buffer being read
    assert bad.returncode == 0 and "caught the deliberately broken schedule" in bad.stdout, bad.stdout + bad.stderr


def test_schedule_descriptor_replays_including_the_convolution():
    for name in ("two_read", "r4"):
        ok = _run("sched_" + name)
        assert ok.returncode == 0 and "WRONG" not in ok.stdout and "exact" in ok.stdout, ok.stdout + ok.stderr
    assert "conv3x3" in _run("sched_two_read").stdout
    bad = _run("sched_bad_early_b")
    assert bad.returncode == 0 and "caught the deliberately broken schedule" in bad.stdout, bad.stdout + bad.stderr


def test_half_tile_mode_replay_and_its_broken_variant():
    ok = _run("half")                 # value half only where a tile's gate half lies beyond N
    assert ok.returncode == 0 and "WRONG" not in ok.stdout and ok.stdout.count("exact") >= 31, ok.stdout + ok.stderr
    bad = _run("break_half_raw")
    assert bad.returncode == 0 and "caught the deliberately broken schedule" in bad.stdout, bad.stdout + bad.stderr


def test_128_row_tile_mode_replay_and_its_broken_variant():
    ok = _run("rows")                 # round 6: a 128-row tile, each wave row runs its m-half 0 only (tile_phases_rows)
    assert ok.returncode == 0 and "WRONG" not in ok.stdout and ok.stdout.count("exact") >= 22, ok.stdout + ok.stderr
    bad = _run("break_rows_raw")
    assert bad.returncode == 0 and "caught the deliberately broken schedule" in bad.stdout, bad.stdout + bad.stderr


def test_attention_index_math_including_the_16x16x32_layout():
    ok = _run("flash_attention")
    assert ok.returncode == 0 and "index math OK" in ok.stdout and ok.stdout.count("16x16x32 experiment") == 3, ok.stdout + ok.stderr
