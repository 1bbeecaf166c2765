"""Host-side integer geometry of the hot path: computed once per ``generate_image`` call, uploaded as small int32
tables, consumed by the HIP kernels.  Nothing here touches the GPU or the RNG.

Each table restates a piece of index logic the reference performs with tensor ops on every step
("ED:n" = /root/reference/elastic_diffusion.py line n):

  * ViewPlan   -- get_views ED:198-229, crop_with_context ED:706-757 (S == 1 => contiguous windows), the edge case
                  ED:820-825, the centre write-back ED:852-861 turned into per-pixel cover lists;
  * PickPlan   -- random_nearest_downsample's cached row/col tables ED:565-613, restore_mask_shape ED:446-465 /
                  ED:622-628 turned into inverse lists, F.interpolate(nearest) index maps ED:636, 688, 922;
  * PadPlan    -- unet_step's pad-to-model-size ED:398-411 and the strip geometry of background_pad ED:366-391;
  * TilePlan   -- tiled_decode ED:275-310.
"""
import math
from dataclasses import dataclass, field
from fractions import Fraction

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------
# windows along one axis
# ---------------------------------------------------------------------------------------------------
def axis_windows(size, ws, stride):
    """[lo,hi) spans of the regular grid along one axis, the last one shifted back inside (ED:206-225)."""
    n = math.ceil((size - ws) / stride) + 1 if stride else 1
    spans = []
    for k in range(int(n)):
        lo = k * stride
        hi = lo + ws
        if hi > size:
            lo = max(0, lo - (hi - size))
            hi = size
        spans.append((lo, hi))
    return spans


def axis_context(lo, hi, size, n):
    """(#context lines before, #after) for a centre [lo,hi) with n wanted per side and unit stride; what is missing
    on one side is given to the other, clipped to the axis (ED:716-744)."""
    if lo - n < 0:
        before = lo
        after = max(0, min(size, hi + (2 * n - before)) - hi)
    else:
        after = max(0, min(size, hi + n) - hi)
        before = lo - max(0, lo - (2 * n - after))
    return before, after


def nearest_index_map(n_in, n_out):
    """Source index per destination index of F.interpolate(mode='nearest'), produced by torch itself so that the
    float32 scale rounding of ATen is reproduced exactly (ED:876)."""
    src = torch.arange(n_in, dtype=torch.float32).view(1, 1, n_in, 1)
    return F.interpolate(src, size=(n_out, 1), mode="nearest").view(-1).to(torch.int64).numpy().astype(np.int32)


# ---------------------------------------------------------------------------------------------------
@dataclass
class PadPlan:
    """Pad a (h,w) model input up to d x d (ED:398-411): offsets, padded size and the strip rectangles."""
    h: int
    w: int
    d: int
    PH: int = 0
    PW: int = 0
    top: int = 0
    bottom: int = 0
    left: int = 0
    right: int = 0
    strips: list = field(default_factory=list)  # [(dim, side, Hs, Ws, y0, x0)] in the reference's RNG order

    def __post_init__(self):
        h_p, w_p = max(self.d - self.h, 0), max(self.d - self.w, 0)
        self.left, self.top = w_p // 2, h_p // 2
        self.right, self.bottom = w_p - self.left, h_p - self.top
        self.PH, self.PW = self.h + h_p, self.w + w_p
        if h_p or w_p:
            # W pair first on the un-widened tensor, then the H pair on the widened one (ED:372-389)
            cand = [(3, 1, self.h, self.left, self.top, 0),
                    (3, 2, self.h, self.right, self.top, self.left + self.w),
                    (2, 1, self.top, self.PW, 0, 0),
                    (2, 2, self.bottom, self.PW, self.top + self.h, 0)]
            self.strips = [s for s in cand if s[2] > 0 and s[3] > 0]  # empty strips never touch the RNG (ED:332-333)

    @property
    def padded(self):
        return self.PH != self.h or self.PW != self.w


# ---------------------------------------------------------------------------------------------------
@dataclass
class ViewPlan:
    H: int
    W: int
    window: int
    stride: int
    context: int
    views: list = None            # (h0,h1,w0,w1) centre rectangles, reference order
    ctx: list = None              # (n_t,n_b,n_l,n_r) per view
    n_row_blocks: int = 0
    n_col_blocks: int = 0
    Sh: int = 0                   # crop size (identical for all views)
    Sw: int = 0
    win_y0: np.ndarray = None     # int32[V] crop origin (context included)
    win_x0: np.ndarray = None

    def __post_init__(self):
        ws, cx = self.window, self.context
        h_ws = self.H if ws + cx >= self.H else ws  # ED:820-825
        w_ws = self.W if ws + cx >= self.W else ws
        rows = axis_windows(self.H, h_ws, self.stride)
        cols = axis_windows(self.W, w_ws, self.stride)
        self.n_row_blocks, self.n_col_blocks = len(rows), len(cols)
        n = cx // 2
        rctx = [axis_context(lo, hi, self.H, n) for lo, hi in rows]
        cctx = [axis_context(lo, hi, self.W, n) for lo, hi in cols]
        self._rows, self._cols, self._rctx, self._cctx = rows, cols, rctx, cctx
        self.views = [(r[0], r[1], c[0], c[1]) for r in rows for c in cols]
        self.ctx = [(rc[0], rc[1], cc[0], cc[1]) for rc in rctx for cc in cctx]
        sh = {b + (hi - lo) + a for (lo, hi), (b, a) in zip(rows, rctx)}
        sw = {b + (hi - lo) + a for (lo, hi), (b, a) in zip(cols, cctx)}
        if len(sh) != 1 or len(sw) != 1:
            raise NotImplementedError(f"views of different crop sizes {sh} x {sw} cannot share one UNet batch")
        self.Sh, self.Sw = sh.pop(), sw.pop()
        self.win_y0 = np.asarray([r[0] - rc[0] for r, rc in zip(rows, rctx) for _ in cols], dtype=np.int32)
        self.win_x0 = np.asarray([c[0] - cc[0] for _ in rows for c, cc in zip(cols, cctx)], dtype=np.int32)

    @property
    def V(self):
        return len(self.views)

    def cover_tables(self, off_y=0, off_x=0):
        """Per latent row Y: the (<=2) row-blocks whose CENTRE contains Y, ascending, and the row of the model
        output that holds it; same per column.  off_* = where the crop sits inside a padded model input."""
        def one(spans, ctxs, size, off):
            blk = np.full((size, 2), -1, dtype=np.int32)
            src = np.zeros((size, 2), dtype=np.int32)
            fill = np.zeros(size, dtype=np.int64)
            for k, ((lo, hi), (before, _)) in enumerate(zip(spans, ctxs)):
                for p in range(lo, hi):
                    if fill[p] >= 2:
                        raise NotImplementedError("more than two overlapping view centres on one line")
                    blk[p, fill[p]] = k
                    src[p, fill[p]] = p - lo + before + off
                    fill[p] += 1
            if (fill == 0).any():
                raise ValueError("views do not cover the latent")
            return blk.reshape(-1), src.reshape(-1)

        rb, rs = one(self._rows, self._rctx, self.H, off_y)
        cb, cs = one(self._cols, self._cctx, self.W, off_x)
        return rb, rs, cb, cs


# ---------------------------------------------------------------------------------------------------
def _even_ratio(f, max_block=32):
    """(keep, block) with both even, approximating f = n_out / n_in (ED:468-476)."""
    fr = Fraction(f).limit_denominator(max_block)
    if fr.numerator % 2 or fr.denominator % 2:
        fr = Fraction(f).limit_denominator(max_block // 2)
    k, b = fr.numerator, fr.denominator
    return (2 * k, 2 * b) if (k % 2 or b % 2) else (k, b)


def _block_pattern(block, n_remove):
    """Which offsets of a block survive when n_remove/2 line pairs are dropped at even spacing, plus the kept-index
    positions that start a 'do not merge' pair when the mask is folded back (ED:478-499)."""
    pairs = n_remove // 2
    step = block // (pairs + 1)
    if step % 2:
        step += 1
    alive = np.ones(block, dtype=bool)
    marks = np.zeros(2 * pairs, dtype=np.int64)
    for p in range(pairs):
        start = (p + 1) * step - 1
        alive[start:start + 2] = False
        marks[2 * p] = start - 1 - 2 * p
        marks[2 * p + 1] = start - 2 * p
    return np.flatnonzero(alive), marks


def _axis_pick_tables(n_in, n_out):
    """One axis of the pick grid (2*n_out lines).

    returns
      src   int32[2*n_out]   latent line behind each grid line (2x nearest upsample => line // 2)       ED:565, 584-613
      fold  int32[2*n_out]   latent-resolution mask line each grid line folds onto                       ED:446-465
      n_mask                 number of mask lines produced by the fold (must be <= n_in, ED:625-628)
    """
    keep, block = _even_ratio(n_out / n_in)
    n_blocks = (2 * n_out) // keep
    if n_blocks * block > 2 * n_in:
        n_blocks -= 1
    covered = n_blocks * block
    offs, marks = _block_pattern(block, block - keep)
    lines = (np.arange(n_blocks)[:, None] * block + offs[None, :]).reshape(-1)
    lines = lines[lines < 2 * n_in]
    missing = 2 * n_out - len(lines)
    if missing > 0:
        lines = np.concatenate([lines, np.arange(2 * n_in)[covered:covered + missing]])
    if len(lines) != 2 * n_out:
        raise ValueError(f"cannot build a {2 * n_out}-line pick grid from {n_in} latent lines "
                         "(the reference fails for this size as well)")
    src = (lines // 2).astype(np.int32)
    all_marks = (np.arange(0, 2 * n_out, keep)[:, None] + marks[None, :]).reshape(-1)
    fold = np.zeros(2 * n_out, dtype=np.int32)
    dst, i, j = 0, 0, 0
    while i < 2 * n_out:
        if j < len(all_marks) and i == all_marks[j]:
            fold[i], fold[i + 1] = dst, dst + 1
            dst += 2
            j += 2
        else:
            fold[i] = fold[i + 1] = dst
            dst += 1
        i += 2
    return src, fold, dst


def _invert_fold(fold, n_in):
    inv = np.full((n_in, 2), -1, dtype=np.int32)
    cnt = np.zeros(n_in, dtype=np.int64)
    for line, dst in enumerate(fold):
        inv[dst, cnt[dst]] = line
        cnt[dst] += 1
    return inv.reshape(-1)


@dataclass
class PickPlan:
    H: int
    W: int
    h: int
    w: int
    src_row: np.ndarray = None
    src_col: np.ndarray = None
    inv_row: np.ndarray = None
    inv_col: np.ndarray = None
    up_row: np.ndarray = None
    up_col: np.ndarray = None
    down_row: np.ndarray = None
    down_col: np.ndarray = None

    def __post_init__(self):
        self.src_row, fold_r, mh = _axis_pick_tables(self.H, self.h)
        self.src_col, fold_c, mw = _axis_pick_tables(self.W, self.w)
        if mh > self.H or mw > self.W:
            # torch.where(mask, ...) raises a shape mismatch in the reference for these sizes (ED:637)
            raise ValueError(f"latent {self.H}x{self.W} -> reduced {self.h}x{self.w}: the pick mask folds to "
                             f"{mh}x{mw}, larger than the latent; the reference does not support this size")
        self.inv_row = _invert_fold(fold_r, self.H)
        self.inv_col = _invert_fold(fold_c, self.W)
        self.up_row = nearest_index_map(self.h, self.H)
        self.up_col = nearest_index_map(self.w, self.W)
        self.down_row = nearest_index_map(self.H, self.h)
        self.down_col = nearest_index_map(self.W, self.w)

    @property
    def N(self):
        return self.h * self.w


def reduced_size(height_px, width_px, sd_version, scale=8):
    """get_downsample_size (ED:943-950): the longer side is brought to the model's training resolution."""
    base = 1024 if "XL" in sd_version else 512
    factor = max(max(height_px, width_px) / base, 1)
    return int((height_px // factor) // scale), int((width_px // factor) // scale)


# ---------------------------------------------------------------------------------------------------
@dataclass
class TilePlan:
    """tiled_decode geometry (ED:275-310) in latent units; pixel tables for the accumulate kernel."""
    H: int
    W: int
    sample_size: int
    scale: int
    low_vram: bool = False
    MAXC: int = 4

    def __post_init__(self):
        self.core = self.sample_size // 4
        self.core_stride = self.core // 2 if self.low_vram else self.core
        self.pad = self.core if self.low_vram else self.sample_size // self.scale * 3
        self.rows = axis_windows(self.H, self.core, self.core_stride)
        self.cols = axis_windows(self.W, self.core, self.core_stride)
        self.n_row_tiles, self.n_col_tiles = len(self.rows), len(self.cols)
        self.Ts = self.core + 2 * self.pad
        self.tile_y0 = np.asarray([r[0] - self.pad for r in self.rows for _ in self.cols], dtype=np.int32)
        self.tile_x0 = np.asarray([c[0] - self.pad for _ in self.rows for c in self.cols], dtype=np.int32)

    @property
    def T(self):
        return self.n_row_tiles * self.n_col_tiles

    def pixel_tables(self):
        s = self.scale

        def one(spans, size):
            tile = np.full((size * s, self.MAXC), -1, dtype=np.int32)
            src = np.zeros((size * s, self.MAXC), dtype=np.int32)
            cnt = np.zeros(size * s, dtype=np.int64)
            for k, (lo, hi) in enumerate(spans):
                for p in range(lo * s, hi * s):
                    if cnt[p] >= self.MAXC:
                        raise NotImplementedError("more than MAXC tiles cover one pixel line")
                    tile[p, cnt[p]] = k
                    src[p, cnt[p]] = p - lo * s + self.pad * s
                    cnt[p] += 1
            return tile.reshape(-1), src.reshape(-1)

        rt, rs = one(self.rows, self.H)
        ct, cs = one(self.cols, self.W)
        return rt, rs, ct, cs
