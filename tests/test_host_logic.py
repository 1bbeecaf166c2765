"""CPU: the product's host-side logic (integer tables, RNG replay, schedule scalars) against the oracle and the
golden fixtures written by the real reference.  Everything here is integer / index work or scalar fp32: exact."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from elasticdiffusion_official_amd import geometry, host_rng, schedule
from oracle import elastic_oracle as eo
from oracle.ddim import DDIMOracle
from tests.fakes import FakeUNet, FakeVAE
from tests.golden import cases


def test_view_plan_matches_reference_views_and_crops(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, "g1_views.json")))
    for r in rows:
        H, W, sample, patch = r["H"], r["W"], r["sample"], r["patch"]
        ws = patch if patch is not None else sample // 2
        vp = geometry.ViewPlan(H // 8, W // 8, ws, ws, sample - ws)
        assert [list(v) for v in vp.views] == r["views"]
        assert [list(c) for c in vp.ctx] == [c["n4"] for c in r["crops"]]
        Wl = W // 8
        for v in range(vp.V):  # contiguous window: first / last element of an index image
            assert [vp.Sh, vp.Sw] == r["crops"][v]["shape"]
            assert int(vp.win_y0[v]) * Wl + int(vp.win_x0[v]) == r["crops"][v]["first"]
            assert (int(vp.win_y0[v]) + vp.Sh - 1) * Wl + int(vp.win_x0[v]) + vp.Sw - 1 == r["crops"][v]["last"]
        assert list(geometry.reduced_size(H, W, "1.5")) == r["downsample_sd"]
        assert list(geometry.reduced_size(H, W, "XL1.0")) == r["downsample_xl"]


def test_cover_tables_reproduce_first_writer_wins():
    for (Hl, Wl, ws, ctx) in [(67, 97, 32, 32), (135, 240, 32, 32), (128, 256, 64, 64), (96, 96, 48, 16)]:
        vp = geometry.ViewPlan(Hl, Wl, ws, ws, ctx)
        rb, rs, cb, cs = vp.cover_tables()
        owner = np.full((Hl, Wl), -1)
        for k, (h0, h1, w0, w1) in reversed(list(enumerate(vp.views))):
            owner[h0:h1, w0:w1] = k  # lowest index wins
        got = rb.reshape(Hl, 2)[:, 0][:, None] * vp.n_col_blocks + cb.reshape(Wl, 2)[:, 0][None, :]
        np.testing.assert_array_equal(got, owner)


@pytest.mark.parametrize("name", list(cases.G2_CASES))
def test_pick_plan_matches_reference_tables_and_masks(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g2_downsample.npz"))
    Hl, Wl, h, w, seed = cases.G2_CASES[name]
    try:
        pp = geometry.PickPlan(Hl, Wl, h, w)
    except ValueError:
        assert name == "r65_97"  # mask folds to 67 lines > 65: the reference's fill raises for this size too
        return
    nr, nc = len(g[f"{name}/table_row_indices"]), len(g[f"{name}/table_col_indices"])
    np.testing.assert_array_equal(pp.src_row[:nr], g[f"{name}/table_row_indices"] // 2)
    np.testing.assert_array_equal(pp.src_col[:nc], g[f"{name}/table_col_indices"] // 2)
    torch.manual_seed(seed)
    x = torch.randn(1, 4, Hl, Wl)
    for step in range(3):
        idx = g[f"{name}/idx{step}"].astype(np.int64).reshape(h, w)
        ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        low = x[0, :, pp.src_row[2 * ii + idx // 2], pp.src_col[2 * jj + idx % 2]]
        np.testing.assert_array_equal(low.numpy(), g[f"{name}/low{step}"][0])
        shape = tuple(g[f"{name}/mask_shape{step}"])
        want = np.unpackbits(g[f"{name}/mask{step}"])[: shape[0] * shape[1]].reshape(shape).astype(bool)
        mask = np.zeros((Hl, Wl), dtype=bool)
        inv_r, inv_c = pp.inv_row.reshape(Hl, 2), pp.inv_col.reshape(Wl, 2)
        for Y in range(Hl):
            for X in range(Wl):
                for r in inv_r[Y]:
                    for c in inv_c[X]:
                        if r >= 0 and c >= 0 and idx[r // 2, c // 2] == (r % 2) * 2 + (c % 2):
                            mask[Y, X] = True
        np.testing.assert_array_equal(mask, want)


def test_nearest_index_maps_are_torch_nearest():
    for n_in, n_out in [(36, 135), (64, 240), (135, 36), (44, 67), (97, 64), (128, 256), (7, 5)]:
        m = geometry.nearest_index_map(n_in, n_out)
        x = torch.randn(1, 1, n_in, 3)
        np.testing.assert_array_equal(F.interpolate(x, size=(n_out, 3), mode="nearest").numpy(), x[:, :, m].numpy())


def test_pad_plan_strip_order_and_sizes():
    p = geometry.PadPlan(32, 64, 64)
    assert (p.PH, p.PW, p.top, p.bottom, p.left, p.right) == (64, 64, 16, 16, 0, 0)
    assert [(s[0], s[1], s[2], s[3]) for s in p.strips] == [(2, 1, 16, 64), (2, 2, 16, 64)]
    q = geometry.PadPlan(33, 50, 64)  # both axes, odd split
    assert (q.top, q.bottom, q.left, q.right) == (15, 16, 7, 7)
    assert [(s[0], s[1], s[2], s[3]) for s in q.strips] == [(3, 1, 33, 7), (3, 2, 33, 7), (2, 1, 15, 64), (2, 2, 16, 64)]
    assert not geometry.PadPlan(64, 64, 64).padded


def test_strip_draws_and_reseed_replay_match_reference_side_effects():
    """make_denoised_background through the oracle vs the product's private-generator draws + replay."""
    orc = eo.ElasticOracle(FakeUNet(64), FakeVAE(), DDIMOracle())
    orc.scheduler.set_timesteps(50)
    t = orc.scheduler.timesteps[4]
    host_rng.seed_everything(3)
    want = orc.make_denoised_background((16, 64), t, "2_1")
    tail_t, tail_n = torch.rand(3), np.random.randint(1000)
    host_rng.seed_everything(3)
    colour, post, fwd = host_rng.strip_draws(2, 1, 16, 64, t)
    host_rng.replay_strip_reseeds(1)
    assert torch.equal(torch.rand(3), tail_t) and np.random.randint(1000) == tail_n
    vae = FakeVAE()
    dist = vae.encode(colour[:, :, None, None].repeat(1, 1, 128, 512)).latent_dist
    enc = (dist.mean + dist.std * post) * vae.config.scaling_factor
    sa, so = schedule.DDIMSchedule().add_noise_coefficients(t)
    assert torch.equal(np.float32(sa) * enc + np.float32(so) * fwd, want)


def test_pick_sampler_replays_reference_rng_trace(golden_dir):
    """Full host-side replay for cfg2: same (fn, shape) event sequence as the reference run."""
    from tests.golden.make_golden import RngTrace
    want = json.load(open(os.path.join(golden_dir, "g9_rng_trace.json")))["cfg2_sd_512x1024"]
    # events the product draws on the global generators: everything except the md5-seeded strip internals
    # (private generator) and the re-seed calls themselves (default_generator.manual_seed is not a hooked entry point;
    # that the re-seeding happens is proven by the identical draws that follow and by the rng_tail checks)
    skip, reseed_next = 0, False
    filtered = []
    for ev in want:
        if skip:
            skip -= 1
            continue
        if ev[0] == "manual_seed":
            if not reseed_next:
                skip = 3  # rand(1,3), randn (posterior), randn_like inside the md5-seeded section
            reseed_next = False
            continue
        reseed_next = ev[0] == "np_randint"
        filtered.append(ev)
    c = cases.E2E_CASES["cfg2_sd_512x1024"]
    pp = geometry.PickPlan(64, 128, 32, 64)
    gpad = geometry.PadPlan(32, 64, 64)
    sampler = host_rng.PickSampler(pp.N)
    sch = schedule.DDIMSchedule()
    ts = sch.set_timesteps(c["steps"])
    host_rng.seed_everything(c["seed"])
    with RngTrace() as tr:
        torch.randn(1, 4, 64, 128)
        for i in range(len(ts)):
            sampler.draw(c["R"] + 1, 0.7, lambda: host_rng.replay_strip_reseeds(len(gpad.strips)))
            if i < len(ts) - 1:
                for _ in range(1000 // c["steps"]):
                    torch.randn(1, 4, 64, 128)
                sampler.draw(1, 0.7, lambda: host_rng.replay_strip_reseeds(len(gpad.strips)))
    got = [e for e in tr.events]
    assert got == filtered


@pytest.mark.parametrize("steps", [50, 10, 4, 20])
def test_schedule_scalars_match_oracle_ddim(steps):
    sch, orc = schedule.DDIMSchedule(), DDIMOracle()
    ts = sch.set_timesteps(steps)
    orc.set_timesteps(steps)
    assert torch.equal(ts, orc.timesteps) and str(ts[0]) == f"tensor({int(ts[0])})"
    x, e = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    for t in ts:
        sb, sa, sp, sd = (np.float32(v) for v in sch.step_coefficients(t))
        out = orc.step(e, t, x)
        x0 = (x - sb * e) / sa
        assert torch.equal(x0, out["pred_original_sample"])
        assert torch.equal(sp * x0 + sd * e, out["prev_sample"])
    coef = sch.undo_coefficients(ts[1])
    assert coef.shape == (1000 // steps, 2)
    b = orc.betas[int(ts[1])]
    assert coef[0, 0] == (1 - b) ** 0.5 and coef[0, 1] == b ** 0.5


def test_rrg_schedulers_match_oracle():
    for i in range(60):
        assert schedule.CosineScheduler(40, 10.0, 1000)(i) == eo.CosineScheduler(40, 10.0, 1000)(i)
        assert schedule.LinearScheduler(40, 1000, 0)(i) == eo.LinearScheduler(40, 1000, 0)(i)
        assert schedule.ConstScheduler(40, 1000, 0)(i) == eo.ConstScheduler(40, 1000, 0)(i)


def test_tile_plan_matches_reference_tiling():
    for (Hl, Wl, sample, low_vram) in [(16, 24, 32, False), (20, 28, 32, True), (256, 256, 128, False), (135, 240, 64, True)]:
        tp = geometry.TilePlan(Hl, Wl, sample, 8, low_vram)
        core = sample // 4
        stride = core // 2 if low_vram else core
        views = eo.get_views(Hl * 8, Wl * 8, core, core, stride)
        assert len(views) == tp.T
        pad = core if low_vram else sample // 8 * 3
        assert [(int(y) + pad, int(x) + pad) for y, x in zip(tp.tile_y0, tp.tile_x0)] == [(v[0], v[2]) for v in views]
        rt, rs, ct, cs = tp.pixel_tables()
        count = np.zeros((Hl * 8, Wl * 8))
        for (h0, h1, w0, w1) in views:
            count[h0 * 8:h1 * 8, w0 * 8:w1 * 8] += 1
        got = (rt.reshape(-1, 4) >= 0).sum(1)[:, None] * (ct.reshape(-1, 4) >= 0).sum(1)[None, :]
        np.testing.assert_array_equal(got, count)


@pytest.mark.parametrize("h,w,K,seed", [(32, 64, 4, 0), (64, 128, 8, 1), (8, 8, 12, 2)])
def test_pick_sampler_draws_equal_oracle_chain(h, w, K, seed):
    """Same picks as the oracle's random_downsample chain (incl. exhausted-choice fallback at K > 4), same generator
    end state, and a stamp table that reproduces 'last step that picked q'."""
    orc = eo.ElasticOracle(FakeUNet(64), FakeVAE(), DDIMOracle())
    N = h * w
    x = torch.zeros(1, 1, 2 * h, 2 * w)
    torch.manual_seed(seed)
    prev, exclude, want = None, None, []
    for k in range(K):
        _, _, prev = orc.random_downsample(x, exclude, prev, drop_p=0.7, nearest=(k == 0))
        if exclude is None:
            exclude = torch.zeros(N, 4, dtype=torch.bool)
        exclude[torch.arange(N), prev] = True
        want.append(prev.clone())
    tail = torch.rand(3)
    torch.manual_seed(seed)
    stamp = torch.empty(N, 4, dtype=torch.int8)
    got = host_rng.PickSampler(N).draw(K, 0.7, lambda: None, stamp=stamp)
    assert torch.equal(torch.rand(3), tail)
    assert torch.equal(got.long(), torch.stack(want))
    ref = torch.full((N, 4), -1, dtype=torch.int8)
    for k in range(K):
        ref[torch.arange(N), want[k]] = k
    assert torch.equal(stamp, ref)


def test_make_grid_matches_torchvision_layout():
    """pipeline._make_grid == torchvision.utils.make_grid(imgs, nrow=8, padding=2, pad_value=0) (what ED:1124 calls)."""
    import torch
    from elasticdiffusion_official_amd.pipeline import _make_grid
    one = torch.rand(1, 3, 5, 7)
    assert torch.equal(_make_grid(one), one[0])                       # a single image passes through
    imgs = torch.rand(10, 3, 4, 6)
    g = _make_grid(imgs)
    assert g.shape == (3, 2 * (4 + 2) + 2, 8 * (6 + 2) + 2)           # 8 per row, 2 rows, 2-px frame and separators
    assert torch.equal(g[:, 2:6, 2:8], imgs[0]) and torch.equal(g[:, 2:6, 10:16], imgs[1])
    assert torch.equal(g[:, 8:12, 2:8], imgs[8]) and float(g[:, :2].abs().sum()) == 0.0
    assert float(g[:, 8:12, 18:].abs().sum()) == 0.0                  # the unused cells of the last row stay zero


def test_flash_variant_heuristic():
    from elasticdiffusion_official_amd import ops
    # round 3 (profiles/r3_s2_probe_attn.jsonl): the pipelined kernel wins or ties on every self-attention shape, the
    # small-KV kernel on the 77-token cross attention
    pipe = ops.FLASH_DEFAULT_PIPE
    assert pipe in (4, 5)
    assert ops._flash_variant(20, 10, 4096, 4096) == pipe   # SDXL level-2 self attention, natural-domain q
    assert ops._flash_variant(20, 20, 1024, 1024) == pipe   # level 3
    assert ops._flash_variant(20, 10, 4096, 4096, prescaled=True) == 6   # exponent-domain q (models.Attention, self-attention)
    assert ops._flash_variant(20, 10, 4096, 77) == 8     # cross attention (Nk <= 96)
    assert ops._flash_variant(1, 10, 4096, 100) == 0     # neither: one full tile + a ragged one
    class _Strided:   # stand-in for a k / v tensor with a huge token stride: 32-bit K offsets would overflow -> round-2 kernel
        def stride(self, dim):
            return 2 ** 20
    assert ops._flash_variant(20, 10, 4096, 4096, _Strided(), _Strided()) == 2
    # which shapes get exponent-domain queries: self-attention with >= 2 full key tiles and 32-bit addressable K / V
    c = 0.125 * 1.4426950408889634
    saved = ops.FLASH_V_PATH, ops.FLASH_EXP2
    try:
        ops.FLASH_EXP2 = False   # the default: no in-situ gain measured (ops.py)
        assert ops.flash_prescale(4096, 4096, 3 * 640) is None
        ops.FLASH_EXP2 = True
        assert ops.flash_prescale(4096, 4096, 3 * 640) == c and ops.flash_prescale(1024, 1024, 1280) == c
        assert ops.flash_prescale(64, 64, 3 * 1280) is None and ops.flash_prescale(4096, 4096, 2 ** 20) is None
        ops.FLASH_V_PATH = 1
        assert ops._flash_variant(20, 10, 4096, 4096) == 1 and ops.flash_prescale(4096, 4096, 3 * 640) is None
        ops.FLASH_V_PATH = 6   # forcing 6 only applies where the caller really pre-scaled q
        assert ops.flash_prescale(4096, 4096, 3 * 640) == c and ops._flash_variant(20, 10, 4096, 77) == 8
    finally:
        ops.FLASH_V_PATH, ops.FLASH_EXP2 = saved


def test_miopen_db_derivation_is_idempotent_and_well_formed(tmp_path):
    """tools/miopen_nhwc_from_nchw.py on a copy of the in-tree db: running it twice changes nothing, every NHWC perf-db
    key has its NCHW twin's CK instance, and borrowed records exist for the untuned batch sizes of the UNet shapes."""
    import glob
    import importlib.util
    import os
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    work = os.path.join(tmp_path, "miopen_cache")
    shutil.copytree(os.path.join(root, "miopen_cache"), work)
    spec = importlib.util.spec_from_file_location("nhwc_tool", os.path.join(root, "tools", "miopen_nhwc_from_nchw.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    tool.CACHE = work
    def all_steps():
        tool.main()
        tool.borrow_ck_instances()
        tool.borrow_ck_instances("FP16")
        tool.derive_find_records("FP16")
        tool.derive_find_records("BF16")

    all_steps()
    snap = {f: open(f).read() for f in glob.glob(os.path.join(work, "*.txt"))}
    all_steps()
    assert all(open(f).read() == s for f, s in snap.items())
    # the committed db already is the fixed point (nothing the product needs is produced only by running the tool here)
    assert all(open(os.path.join(root, "miopen_cache", os.path.basename(f))).read() == s for f, s in snap.items())
    udb = tool.read(glob.glob(os.path.join(work, "*.udb.txt"))[0])
    k20 = "2x320x128x128x1x3x3x1x320x20x1x1x0x1x1x0x1x1x0x0x1x{}xBF16xF"
    inst = [r for r in udb[k20.format("NCHW")].split(";") if r.startswith(tool.CK)][0]
    assert inst in udb[k20.format("NHWC")]
    assert inst in udb[k20.format("NHWC").replace("x320x20x", "x320x32x")]      # borrowed for cfg4's batch 32
    assert tool.CK in udb["2x320x64x64x1x3x3x1x320x20x1x1x0x1x1x0x1x1x0x0x1xNHWCxBF16xF"]  # SD1.5 shape
    # round 3: the default fp16 UNet -- real finds at the bench's batches (20, 6) and the per-rank batches of row sharding
    # (10, 3), perf-db + find-db records (borrowed / derived) for every other batch, so no rank count starts with a find
    ufdb = tool.read(glob.glob(os.path.join(work, "*.ufdb.txt"))[0])
    f16 = "320-128-128-3x3-320-128-128-{}-1x1-1x1-1x1-0-NHWC-NHWC-NHWC-FP16-F"
    for n in (20, 6, 10, 3, 32, 18, 5, 1):
        assert ufdb[f16.format(n)].startswith(tool.CK), n
        assert tool.CK in udb[k20.format("NHWC").replace("x320x20x", f"x320x{n}x").replace("BF16", "FP16")], n


def test_flash_attention_index_math_emulation():
    """tools/emulate_flash_attention.py: the attention kernel's LDS addresses, MFMA fragment slots and accumulator
    registers, emulated lane by lane under the documented gfx950 layouts, reproduce softmax(QK^T)V -- ragged Nq / Nk
    (cross attention's 77 keys) and both V staging paths."""
    import importlib.util
    import os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("emu", os.path.join(root, "tools", "emulate_flash_attention.py"))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    rng = np.random.default_rng(1)
    Nq, Nk = 70, 77
    q, k, v = rng.standard_normal((Nq, 64)), rng.standard_normal((Nk, 64)), rng.standard_normal((Nk, 64))
    want = emu.reference(q, k, v, 0.125)
    for tr in (True, False):
        got = emu.run_block(q, k, v, Nq, Nk, 0, 0.125, tr)
        assert sorted(got) == list(range(Nq))
        assert max(np.abs(got[r] - want[r]).max() for r in got) < 1e-12
    # round 4: the generic-head-dimension kernel (SD 1.x: 40 / 80 / 160), LDS poisoned with NaN before the kernel's own writes
    for DH in (40, 80, 160):
        q, k, v = rng.standard_normal((Nq, DH)), rng.standard_normal((Nk, DH)), rng.standard_normal((Nk, DH))
        want = emu.reference(q, k, v, DH ** -0.5)
        got = emu.run_block_gen(q, k, v, Nq, Nk, 0, DH ** -0.5, DH)
        assert sorted(got) == list(range(Nq))
        assert max(np.abs(got[r] - want[r]).max() for r in got) < 1e-12


def test_pipelined_attention_schedule_emulation():
    """tools/emulate_attention_pipeline.py: (1) the pipelined attention kernels' LDS buffer rotation has no interval in
    which a buffer is both read and written and every read finds its tile -- for the exact kernel (2 K buffers) and the
    lazy-maximum one (3), while lazy with 2 K buffers is flagged (the race the GPU accuracy tests caught in round 3);
    (2) the 16-slice softmax, the deferred rescale and the lazy variant's check / slow path reproduce softmax(S) V."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pipe_emu", os.path.join(root, "tools", "emulate_attention_pipeline.py"))
    emu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(emu)
    for n_tiles in range(1, 10):
        for n_full in (n_tiles, n_tiles - 1):
            slow = range(1, max(1, n_full))
            assert emu.lds_hazards(n_tiles, n_full, 2, False) == []
            assert emu.lds_hazards(n_tiles, n_full, 3, True, slow) == []
            if n_full >= 2:
                assert emu.lds_hazards(n_tiles, n_full, 2, True, slow) != []
    rng = np.random.default_rng(3)
    slow_total = 0
    for trial in range(24):
        n = int(rng.integers(1, 10))
        sc = rng.normal(size=(n, 64)) * 3.0
        if trial % 3 == 0 and n > 2:
            sc[n - 2, 40] += 70.0                      # a late tile far above the reference (other lane of the row)
        if trial % 3 == 1:
            sc += np.arange(n)[:, None] * 2.5          # creeping maximum: deferred until it has grown by 2^6
        v = rng.normal(size=(n, 64, 4))
        want = emu.reference(sc, v, 0.18)
        for lazy in (False, True):
            got, slow_tiles = emu.softmax_schedule(sc, v, 0.18, lazy)
            assert np.abs(got - want).max() < 1e-12
            slow_total += slow_tiles
        got, slow6 = emu.exp2_schedule(sc, v)            # v_path 6: reference carried in the S accumulators' initial value
        assert np.abs(got - emu.reference(sc, v, 1.0)).max() < 1e-12
        slow_total += slow6
    assert slow_total > 0   # the slow path was exercised


def test_round5_shape_policies_and_graph_keys():
    """Host-side decisions added in round 5 (ADVICE r4 + the VAE split path), CPU only: the GEGLU GEMM's grid policy, the split-weight
    preparation at its edges, the per-shape gates of the split VAE path, and the hipGraph key that separates fresh side inputs."""
    import math
    import torch
    from elasticdiffusion_official_amd import models as M, ops
    from elasticdiffusion_official_amd.graphs import GraphedForward
    # GEGLU: the fused pair is used only where the grid fills the chip (the same main loop loses 2.4x on a 25-tile grid)
    assert ops.geglu_gemm_wins(20480, 1280, 5120) and ops.geglu_gemm_wins(6144, 1280, 5120)
    assert not ops.geglu_gemm_wins(256, 1280, 5120) and ops.geglu_gemm_ok(256, 1280, 5120)      # 1 x 40 tiles: the kernel takes it, the model does not
    assert not ops.geglu_gemm_wins(20480, 1280, 5000)                                            # I % 128 != 0
    # split weights: scale is a power of two, wl never subnormal for the bulk, all-zero and huge weights survive
    for w in (torch.zeros(8, 64, 3, 3), torch.full((8, 64, 3, 3), 3.0e4), torch.randn(8, 64, 3, 3) * 1e-6):
        ws, sc = ops.split_conv_weight(w)
        assert bool(torch.isfinite(ws.float()).all()) and sc > 0 and math.log2(sc) == round(math.log2(sc))
        hi, _, lo = ws[:, :64].double(), ws[:, 64:128].double(), ws[:, 128:].double()
        assert torch.equal(ws[:, :64], ws[:, 64:128])
        assert float(((hi + lo) * sc - w.double()).abs().max()) <= float(w.abs().max()) * 2.0 ** -20 + 1e-30
    # the fp32 GroupNorm kernel's shapes: 4-channel columns must lie inside one group
    assert ops.groupnorm_nhwc_f32_ok(128, 32) and ops.groupnorm_nhwc_f32_ok(512, 32) and not ops.groupnorm_nhwc_f32_ok(64, 32)
    assert ops.conv3x3_f32out_ok(5, 256, 1024, 384, 128) and not ops.conv3x3_f32out_ok(1, 1024, 2048, 768, 256)   # 3.2 GB operand: 32-bit offsets
    # gates never fire for CPU tensors (the oracle / CPU-baseline copies of the VAE run plain torch)
    blk = M.ResnetBlock2D(128, 128, None, eps=1e-6)
    assert not M._vae_split_ok(torch.zeros(1, 128, 8, 8), blk)
    up = M.Upsample2D(128, vae=True)
    assert not up._split_ok(torch.zeros(1, 128, 8, 8)) and not M.Upsample2D(128)._split_ok(torch.zeros(1, 128, 8, 8))
    x = torch.randn(2, 128, 6, 10)
    y = up(x)                                                                                    # the library path on the CPU: unchanged semantics
    assert tuple(y.shape) == (2, 128, 12, 20)
    assert M._to_nchw(x.contiguous(memory_format=torch.channels_last)).is_contiguous()
    # one hipGraph per (shape, dtype, condition, timestep shape, fresh side inputs)
    k0 = GraphedForward._key((20, 4, 128, 128), torch.float16, None, ())
    k1 = GraphedForward._key((20, 4, 128, 128), torch.float16, None, (), True)
    assert k0 != k1 and k0 == GraphedForward._key((20, 4, 128, 128), torch.float16, None)


def test_round6_host_policies_and_cpu_fallbacks():
    """Host-side decisions added in round 6, CPU only: the convolution batch split (a pure function of the grid), the shape gates of the
    fused up-sampler / down-sampler / concatenation paths, and that on the CPU (the oracle's and the CPU baseline's copies of the modules)
    every new path keeps plain-torch semantics."""
    import torch
    import torch.nn.functional as F
    from elasticdiffusion_official_amd import models as M, ops
    # 800 tiles (3 full rounds of 256 + 32): the last two samples run on their own; grids that end well, or under three rounds, stay one launch
    assert ops.conv3x3_batch_split(40, 32, 32, 1280) == 38 and ops.conv3x3_batch_split(80, 32, 32, 1280) == 76
    for shape in ((20, 32, 32, 1280), (12, 64, 64, 640), (6, 64, 64, 640), (40, 128, 128, 320), (40, 64, 64, 640), (1, 32, 32, 1280), (40, 8, 8, 1280)):
        assert ops.conv3x3_batch_split(*shape) is None, shape
    # the fused up-sampler takes full grids only (no 128-row instantiation); H, W = the OUTPUT size
    assert ops.conv3x3_up2x_wins(20, 64, 64, 1280, 1280) and ops.conv3x3_up2x_wins(6, 128, 128, 640, 640)
    assert not ops.conv3x3_up2x_wins(1, 64, 64, 1280, 1280) and not ops.conv3x3_up2x_wins(20, 63, 64, 1280, 1280)
    assert ops.conv3x3_f32out_s2_ok(5, 256, 1024, 384, 128) and not ops.conv3x3_f32out_s2_ok(5, 255, 1024, 384, 128)
    assert not ops.conv3x3_f32out_s2_ok(16, 256, 1024, 384, 128)                    # 3.2 GB operand: 32-bit offsets -> the model slices the batch
    # CPU tensors never take a HIP path: ResnetBlock2D.forward_cat IS the block on the concatenation, Upsample2D / Downsample2D plain torch
    torch.manual_seed(0)
    blk = M.ResnetBlock2D(24 + 16, 32, 64, groups=8).eval()
    x, skip, temb = torch.randn(2, 24, 6, 5), torch.randn(2, 16, 6, 5), torch.randn(2, 64)
    with torch.no_grad():
        assert torch.equal(blk.forward_cat(x, skip, temb), blk(torch.cat([x, skip], 1), temb))
        down = M.Downsample2D(16, padding=0).eval()
        y = torch.randn(2, 16, 8, 6)
        assert torch.equal(down(y), F.conv2d(F.pad(y, (0, 1, 0, 1)), down.conv.weight, down.conv.bias, stride=2))
    assert not ops.groupnorm_nhwc_cat_ok(x, skip, 8)                                  # CPU tensors
    w1, w2 = M._split_shortcut(blk.conv_shortcut, 24)
    assert torch.equal(torch.cat([w1, w2], 1), blk.conv_shortcut.weight.reshape(32, 40)) and w1.is_contiguous() and w2.is_contiguous()
    # switches exist under the names the A/B tools and ED_DISABLE use; the two measured non-gains are off
    assert M.FUSED_SKIP_CAT and M.FUSED_UPSAMPLE_CONV and M.FUSED_PROJ_OUT_ADD and M.VAE_SPLIT_DOWNSAMPLE
    assert not M.HIP_DOWNSAMPLE_CONV and not M.FUSED_RESIDUAL_LINEAR


def test_precision_attribution_tool_reproduces_its_headline_on_the_reduced_width_model():
    """tools/r5_precision_modes.py (CPU): rounding ONE class of values of the fp32 UNet to fp16 -- the reduced-width forward must show what
    profiles/r5_precision_attribution.json shows at full width: the classes are comparable (6-7e-4 each), attention's operands are
    negligible, all of them together land near the 1.3e-3 the MI355X measured for the fused fp16 forward, and an fp32 residual stream with
    fp32 epilogue adds removes about a third."""
    import importlib.util
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("r5_precision_modes", os.path.join(root, "tools", "r5_precision_modes.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    from elasticdiffusion_official_amd import models as M
    cfg = M.SMALL_UNET_CONFIGS["sdxl"]
    unet = M.UNet2DConditionModel(**cfg)
    M._seeded_init(unet, 0)
    unet = unet.eval().requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 128, 128, generator=g)
    txt = torch.randn(1, 77, cfg["cross_attention_dim"], generator=g)
    kw = {"text_embeds": torch.randn(1, cfg["pooled_projection_dim"], generator=g), "time_ids": torch.zeros(1, 6)}
    t = torch.tensor(500)
    with torch.no_grad():
        ref = unet(x, t, encoder_hidden_states=txt, added_cond_kwargs=kw)["sample"]
        err = {m: tool.rel_l2(tool.Rounded(unet, tool.MODES[m], torch.float16)(x, t, encoder_hidden_states=txt, added_cond_kwargs=kw)["sample"], ref)
               for m in ("w", "act", "attn", "fp16_model", "mixed_out32")}
    assert 3e-4 < err["w"] < 1.2e-3 and 3e-4 < err["act"] < 1.2e-3 and err["attn"] < 1e-4, err
    assert 8e-4 < err["fp16_model"] < 2.5e-3 and 0.5 * err["fp16_model"] < err["mixed_out32"] < 0.9 * err["fp16_model"], err
