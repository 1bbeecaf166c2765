"""CPU: sweep of image sizes (every multiple of 8 the reference could be asked for is too many; a seeded sample of
~150 sizes per model family): the product's index tables against the oracle's tensor-level functions.
  * ViewPlan == get_views + crop_with_context (windows, context split, crop origin);
  * PickPlan: gathering through src_row/src_col == random_nearest_downsample's picked values, and the fold tables
    reproduce its mask exactly; sizes whose mask folds larger than the latent must be rejected by PickPlan exactly when
    the oracle's fill raises (the reference fails there too, ED:637)."""
import numpy as np
import pytest
import torch

from elasticdiffusion_official_amd import geometry
from oracle import elastic_oracle as eo
from oracle.ddim import DDIMOracle
from tests.fakes import FakeUNet, FakeVAE


def sizes(seed, base, n):
    rng = np.random.RandomState(seed)
    out = {(base, base), (base, 2 * base), (2 * base, base), (2 * base, 2 * base)}
    while len(out) < n:
        out.add((int(rng.randint(base // 16, 2 * base // 8 + 1)) * 8, int(rng.randint(base // 16, 2 * base // 8 + 1)) * 8))
    return sorted(out)


@pytest.mark.parametrize("sd,base,sample", [("1.5", 512, 64), ("XL1.0", 1024, 128)])
def test_tables_match_oracle_over_many_sizes(sd, base, sample):
    orc = eo.ElasticOracle(FakeUNet(sample), FakeVAE(), DDIMOracle(), sd_version=sd)
    g = torch.Generator().manual_seed(0)
    n_ok = n_rejected = 0
    for (H, W) in sizes(1, base, 90 if base == 512 else 30):
        Hl, Wl = H // 8, W // 8
        ws = sample // 2
        vp = geometry.ViewPlan(Hl, Wl, ws, ws, sample - ws)
        h_ws = Hl if sample >= Hl else ws
        w_ws = Wl if sample >= Wl else ws
        views = eo.get_views(H, W, h_ws, w_ws, ws)
        assert views == vp.views, (H, W)
        X = torch.arange(Hl * Wl, dtype=torch.float32).view(1, 1, Hl, Wl)
        for k, v in enumerate(views):
            crop, n4 = eo.crop_with_context(X, *v, 1, (sample - ws) // 2)
            assert tuple(n4) == tuple(vp.ctx[k]) and tuple(crop.shape[-2:]) == (vp.Sh, vp.Sw)
            assert int(crop[0, 0, 0, 0]) == int(vp.win_y0[k]) * Wl + int(vp.win_x0[k])
        h, w = geometry.reduced_size(H, W, sd)
        assert (h, w) == eo.get_downsample_size(H, W, sd)
        x = torch.randn(1, 2, Hl, Wl, generator=g)
        torch.manual_seed(H * 4096 + W)
        try:
            low, mask, idx = orc.random_nearest_downsample(x, (h, w), nearest=False)
            # supported = the mask fits the latent (else ED:637 raises) AND the reduced latent really has the requested
            # size (else the RRG branch ED:918 adds a 31-row score to a 32-row direction and raises)
            oracle_ok = mask.shape == (Hl, Wl) and tuple(low.shape[-2:]) == (h, w)
        except Exception:  # noqa: BLE001 -- sizes the reference itself cannot handle
            oracle_ok = False
        try:
            pp = geometry.PickPlan(Hl, Wl, h, w)
        except ValueError:
            assert not oracle_ok, (H, W, "product rejects a size the reference supports")
            n_rejected += 1
            continue
        assert oracle_ok, (H, W, "product accepts a size the reference cannot fill")
        q = idx.view(h, w).numpy()
        ii, jj = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        got_low = x[0][:, pp.src_row[2 * ii + q // 2], pp.src_col[2 * jj + q % 2]]
        assert torch.equal(got_low, low[0]), (H, W)
        m = np.zeros((Hl, Wl), dtype=bool)
        inv_r, inv_c = pp.inv_row.reshape(Hl, 2), pp.inv_col.reshape(Wl, 2)
        rr = 2 * ii + q // 2
        cc = 2 * jj + q % 2
        picked = np.zeros((2 * h, 2 * w), dtype=bool)
        picked[rr, cc] = True
        for a in range(2):
            for b in range(2):
                r_ok, c_ok = inv_r[:, a] >= 0, inv_c[:, b] >= 0
                sub = picked[np.clip(inv_r[:, a], 0, None)][:, np.clip(inv_c[:, b], 0, None)]
                m |= sub & r_ok[:, None] & c_ok[None, :]
        np.testing.assert_array_equal(m, mask.numpy(), err_msg=str((H, W)))
        n_ok += 1
    assert n_ok > 20, (n_ok, n_rejected)
