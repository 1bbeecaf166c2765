"""Torch-tensor wrappers over the C ABI (include/elastic_hip.h).  PyTorch is only plumbing here: it owns the device
buffers and the stream; every wrapper validates its arguments, passes raw device pointers and raises on a HIP
error.  No wrapper has a fallback path -- a tensor that is not on a ROCm device is an error."""
import torch

from . import _hip

ED_F32, ED_F16, ED_BF16 = 0, 1, 2
_DTYPE = {torch.float32: ED_F32, torch.float16: ED_F16, torch.bfloat16: ED_BF16}


# Device of the launch being assembled: every ``_dev`` of one wrapper call must name the same ROCm device; ``_stream``
# (always the last argument evaluated) and ``_call`` then use THAT device's current stream and make it the current
# device around the ctypes launch.  Nothing here depends on the process-wide current device (ADVICE r1:
# ``ElasticDiffusion('cuda:1')`` in a process whose current device is 0 must not launch on device 0).
_LAUNCH = {"device": None}


def _stream_of(t):
    """``_stream`` for wrappers whose tensor arguments are not all checked through ``_dev`` (non-default strides)."""
    if _LAUNCH["device"] is None:
        _LAUNCH["device"] = t.device
    return _stream()


def _stream():
    dev = _LAUNCH["device"]
    if dev is None:
        raise RuntimeError("internal: _stream() before any device tensor argument")
    return torch.cuda.current_stream(dev).cuda_stream


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline leg).
    Disabled by default: zero overhead on the product path."""

    def __init__(self):
        self.enabled = False
        self.events = {}
        self.work = {}

    def start(self):
        self.events, self.work, self.enabled = {}, {}, True

    def stop(self):
        """-> {kernel: (launches, mean_us, total_ms)}; synchronises.  ``self.work[kernel]`` = [flops, bytes] summed over
        the timed launches (ALGORITHMIC work each wrapper declared via ``note_work``)."""
        self.enabled = False
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), 1e3 * sum(ms) / len(ms), sum(ms))
        return out

    def note_work(self, name, flops=0.0, nbytes=0.0):
        if self.enabled and not torch.cuda.is_current_stream_capturing():
            w = self.work.setdefault(name, [0.0, 0.0])
            w[0] += flops
            w[1] += nbytes


TIMER = KernelTimer()


def _call(name, *args):
    fn = getattr(_hip.lib(), name)
    dev, _LAUNCH["device"] = _LAUNCH["device"], None
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):  # a HIP launch runs on the calling thread's current device
            _LAUNCH["device"] = dev
            return _call(name, *args)
    if TIMER.enabled and not torch.cuda.is_current_stream_capturing():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        err = fn(*args)
        b.record()
        TIMER.events.setdefault(name, []).append((a, b))
    else:
        err = fn(*args)
    _hip.check(err, name)


def _reject(msg):
    _LAUNCH["device"] = None  # an abandoned launch must not leak its device into the next one
    raise RuntimeError(msg)


def _dev(t, dtype=None, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        _reject(f"{name} must be a tensor on the MI355X (got {type(t).__name__}"
                f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}); no CPU fallback")
    if not t.is_contiguous():
        _reject(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        _reject(f"{name} must be {dtype}, got {t.dtype}")
    if _LAUNCH["device"] is None:
        _LAUNCH["device"] = t.device
    elif _LAUNCH["device"] != t.device:
        _reject(f"{name} lives on {t.device} but an earlier argument of this launch is on {_LAUNCH['device']}")
    return t.data_ptr()


def _opt(t, dtype=None, name="tensor"):
    return None if t is None else _dev(t, dtype, name)


def _code(t, name):
    try:
        return _DTYPE[t.dtype]
    except KeyError:
        _reject(f"{name}: unsupported dtype {t.dtype}")


def gather_views(latent, out, win_y0, win_x0, Sh, Sw, off_y=0, off_x=0, frame=None, divisor=1.0):
    """latent f32 [B,C,H,W] -> out [(V*B),C,PH,PW] (row = v*B+b).  See ed_gather_views."""
    B, C, H, W = latent.shape
    V = win_y0.numel()
    rows, C2, PH, PW = out.shape
    assert rows == V * B and C2 == C and win_x0.numel() == V
    assert off_y >= 0 and off_x >= 0 and off_y + Sh <= PH and off_x + Sw <= PW
    if frame is not None:
        assert tuple(frame.shape) == (C, PH, PW)
    _call("ed_gather_views", _dev(latent, torch.float32, "latent"), _dev(out, None, "out"), _code(out, "out"),
          B, C, H, W, _dev(win_y0, torch.int32, "win_y0"), _dev(win_x0, torch.int32, "win_x0"),
          V, Sh, Sw, PH, PW, off_y, off_x, _opt(frame, torch.float32, "frame"),
          float(divisor), _stream())
    return out


def scatter_centres(pred, local, n_col_blocks, row_blk, row_src, col_blk, col_src):
    """pred [(V*B),C,PH,PW] -> local f32 [B,C,H,W], first writer (by value) wins.  See ed_scatter_centres."""
    B, C, H, W = local.shape
    rows, C2, PH, PW = pred.shape
    assert C2 == C and rows % B == 0
    assert row_blk.numel() == H * 2 == row_src.numel() and col_blk.numel() == W * 2 == col_src.numel()
    _call("ed_scatter_centres", _dev(pred, None, "pred"), _code(pred, "pred"), _dev(local, torch.float32, "local"),
          B, C, H, W, PH, PW, n_col_blocks,
          _dev(row_blk, torch.int32), _dev(row_src, torch.int32),
          _dev(col_blk, torch.int32), _dev(col_src, torch.int32), _stream())
    return local


def pick_assemble(latent, idx, src_row, src_col, out, h, w, off_y=0, off_x=0, frame=None, low=None):
    """latent f32 [B,C,H,W], idx u8 [K,h*w] -> out [(K*2*B),C,PH,PW], low f32 [K,B,C,h,w].  See ed_pick_assemble."""
    B, C, H, W = latent.shape
    K = idx.shape[0]
    rows, C2, PH, PW = out.shape
    assert rows == K * 2 * B and C2 == C and idx.shape[1] == h * w
    assert src_row.numel() == 2 * h and src_col.numel() == 2 * w
    assert off_y + h <= PH and off_x + w <= PW
    if frame is not None:
        assert tuple(frame.shape) == (C, PH, PW)
    if low is not None:
        assert tuple(low.shape) == (K, B, C, h, w)
    _call("ed_pick_assemble", _dev(latent, torch.float32, "latent"), _dev(idx, torch.uint8, "idx"),
          _dev(src_row, torch.int32), _dev(src_col, torch.int32),
          _opt(frame, torch.float32, "frame"), _dev(out, None, "out"), _code(out, "out"),
          _opt(low, torch.float32, "low"), K, B, C, H, W, h, w, PH, PW, off_y, off_x,
          _stream())
    return out


def assemble_rows(latent, idx, src_row, src_col, g_rows, h, w, g_off_y, g_off_x, gframe, low, v_rows, win_y0, win_x0, Sh,
                  Sw, v_off_y, v_off_x, vframe):
    """pick_assemble + gather_views in one launch (see ed_assemble_rows): g_rows [(K*2*B),C,gPH,gPW] and v_rows
    [(V*B),C,vPH,vPW] are the two parts of one fused model batch (usually slices of one tensor)."""
    B, C, H, W = latent.shape
    K, V = idx.shape[0], win_y0.numel()
    gr, C2, gPH, gPW = g_rows.shape
    vr, C3, vPH, vPW = v_rows.shape
    assert gr == K * 2 * B and vr == V * B and C2 == C == C3 and idx.shape[1] == h * w and g_rows.dtype == v_rows.dtype
    assert src_row.numel() == 2 * h and src_col.numel() == 2 * w and win_x0.numel() == V
    assert g_off_y + h <= gPH and g_off_x + w <= gPW and v_off_y + Sh <= vPH and v_off_x + Sw <= vPW
    if gframe is not None:
        assert tuple(gframe.shape) == (C, gPH, gPW)
    if vframe is not None:
        assert tuple(vframe.shape) == (C, vPH, vPW)
    if low is not None:
        assert tuple(low.shape) == (K, B, C, h, w)
    _call("ed_assemble_rows", _dev(latent, torch.float32, "latent"), B, C, H, W, _dev(idx, torch.uint8, "idx"),
          _dev(src_row, torch.int32), _dev(src_col, torch.int32), _opt(gframe, torch.float32, "gframe"),
          _dev(g_rows, None, "g_rows"), _opt(low, torch.float32, "low"), K, h, w, gPH, gPW, g_off_y, g_off_x,
          _dev(win_y0, torch.int32), _dev(win_x0, torch.int32), _opt(vframe, torch.float32, "vframe"),
          _dev(v_rows, None, "v_rows"), V, Sh, Sw, vPH, vPW, v_off_y, v_off_x, _code(g_rows, "g_rows"), _stream())


def phase_epilogue(g_out, v_out, x, stamp, pick_tables, view_tables, n_col_blocks, g_off, K, h, w, g, coef, prev, x0,
                   low_dir=None, uncond_last=None, direction=None, local=None, x_next=None, low_latent=None,
                   rrg_norm=0.0, rrg_weight=0.0):
    """unpad_direction + fill_directions + scatter_centres + cfg_ddim_step (+ rrg_update when ``x_next`` is given) in
    one launch (see ed_phase_epilogue).  pick_tables = (inv_row, inv_col, up_row, up_col, down_row, down_col);
    view_tables = (row_blk, row_src, col_blk, col_src); coef = DDIMSchedule.step_coefficients(t)."""
    B, C, H, W = x.shape
    gr, C2, gPH, gPW = g_out.shape
    vr, C3, vPH, vPW = v_out.shape
    assert gr == K * 2 * B and C2 == C == C3 and vr % B == 0 and g_out.dtype == v_out.dtype
    assert tuple(stamp.shape) == (h * w, 4) and tuple(prev.shape) == tuple(x.shape) == tuple(x0.shape)
    inv_row, inv_col, up_row, up_col, down_row, down_col = pick_tables
    assert inv_row.numel() == 2 * H and inv_col.numel() == 2 * W and up_row.numel() == H and up_col.numel() == W
    assert down_row.numel() == h and down_col.numel() == w
    row_blk, row_src, col_blk, col_src = view_tables
    assert row_blk.numel() == H * 2 == row_src.numel() and col_blk.numel() == W * 2 == col_src.numel()
    for t_, shp in ((low_dir, (B, C, h, w)), (uncond_last, (B, C, h, w)), (low_latent, (B, C, h, w)),
                    (direction, (B, C, H, W)), (local, (B, C, H, W)), (x_next, (B, C, H, W))):
        assert t_ is None or tuple(t_.shape) == shp
    if x_next is not None and low_latent is None:
        _reject("phase_epilogue: x_next (fused RRG) needs low_latent")
    f32 = torch.float32
    _call("ed_phase_epilogue", _dev(g_out, None, "g_out"), _dev(v_out, None, "v_out"), _code(g_out, "g_out"),
          _dev(x, f32, "x"), _dev(stamp, torch.int8, "stamp"), *(_dev(t_, torch.int32) for t_ in pick_tables),
          *(_dev(t_, torch.int32) for t_ in view_tables), _opt(low_latent, f32, "low_latent"), _dev(prev, f32, "prev"),
          _dev(x0, f32, "x0"), _opt(x_next, f32, "x_next"), _opt(low_dir, f32, "low_dir"),
          _opt(uncond_last, f32, "uncond_last"), _opt(direction, f32, "direction"), _opt(local, f32, "local"),
          K, B, C, H, W, h, w, gPH, gPW, g_off[0], g_off[1], vPH, vPW, n_col_blocks, float(g), *(float(c) for c in coef),
          float(rrg_norm), float(rrg_weight), _stream())
    return prev, x0


def unpad_direction(unet_out, dirs, uncond_last, off_y=0, off_x=0):
    """unet_out [(K*2*B),C,PH,PW] -> dirs f32 [K,B,C,h,w] (= cond - uncond), uncond_last f32 [B,C,h,w]."""
    K, B, C, h, w = dirs.shape
    rows, C2, PH, PW = unet_out.shape
    assert rows == K * 2 * B and C2 == C
    if uncond_last is not None:
        assert tuple(uncond_last.shape) == (B, C, h, w)
    _call("ed_unpad_direction", _dev(unet_out, None, "unet_out"), _code(unet_out, "unet_out"),
          _dev(dirs, torch.float32, "dirs"), _opt(uncond_last, torch.float32, "uncond_last"),
          K, B, C, h, w, PH, PW, off_y, off_x, _stream())
    return dirs


def fill_directions(dirs, stamp, inv_row, inv_col, up_row, up_col, down_row, down_col, target, low_dir=None):
    """dirs f32 [K,B,C,h,w] + stamp i8 [h*w,4] -> target f32 [B,C,H,W] (+ low_dir f32 [B,C,h,w])."""
    K, B, C, h, w = dirs.shape
    B2, C2, H, W = target.shape
    assert (B2, C2) == (B, C) and tuple(stamp.shape) == (h * w, 4)
    assert inv_row.numel() == 2 * H and inv_col.numel() == 2 * W and up_row.numel() == H and up_col.numel() == W
    assert down_row.numel() == h and down_col.numel() == w
    _call("ed_fill_directions", _dev(dirs, torch.float32, "dirs"), _dev(stamp, torch.int8, "stamp"),
          _dev(inv_row, torch.int32), _dev(inv_col, torch.int32),
          _dev(up_row, torch.int32), _dev(up_col, torch.int32),
          _dev(down_row, torch.int32), _dev(down_col, torch.int32),
          _dev(target, torch.float32, "target"), _opt(low_dir, torch.float32, "low_dir"),
          K, B, C, H, W, h, w, _stream())
    return target


def cfg_ddim_step(local, direction, x, prev, x0, g, sqrt_beta_t, sqrt_alpha_t, sqrt_alpha_prev, sqrt_1m_alpha_prev):
    n = x.numel()
    for t in (local, direction, prev, x0):
        assert t.numel() == n
    _call("ed_cfg_ddim_step", _dev(local, torch.float32), _dev(direction, torch.float32), _dev(x, torch.float32),
          _dev(prev, torch.float32), _dev(x0, torch.float32), float(g), float(sqrt_beta_t),
          float(sqrt_alpha_t), float(sqrt_alpha_prev), float(sqrt_1m_alpha_prev), n, _stream())
    return prev, x0


def undo_step(x_in, noise, coef, x_out):
    """noise f32 [n_sub, *x.shape], coef f32 [n_sub,2] on device."""
    n = x_in.numel()
    n_sub = noise.shape[0]
    assert noise.numel() == n_sub * n and coef.numel() == 2 * n_sub and x_out.numel() == n
    _call("ed_undo_step", _dev(x_in, torch.float32), _dev(noise, torch.float32), _dev(coef, torch.float32),
          _dev(x_out, torch.float32), n_sub, n, _stream())
    return x_out


def rrg_update(prev, x0, low_latent, low_uncond, low_dir, up_row, up_col, out, g, sqrt_beta_t, sqrt_alpha_t, norm, weight):
    B, C, H, W = prev.shape
    h, w = low_latent.shape[-2:]
    assert tuple(low_latent.shape) == (B, C, h, w) == tuple(low_uncond.shape) == tuple(low_dir.shape)
    assert up_row.numel() == H and up_col.numel() == W
    _call("ed_rrg_update", _dev(prev, torch.float32), _dev(x0, torch.float32), _dev(low_latent, torch.float32),
          _dev(low_uncond, torch.float32), _dev(low_dir, torch.float32),
          _dev(up_row, torch.int32), _dev(up_col, torch.int32), _dev(out, torch.float32),
          float(g), float(sqrt_beta_t), float(sqrt_alpha_t), float(norm), float(weight),
          B, C, H, W, h, w, _stream())
    return out


def gather2d(inp, out, src_n, rows, cols):
    """out[n,c,i,j] = inp[src_n[n],c,rows[n,i],cols[n,j]] (0 where an index is negative)."""
    _, C, H, W = inp.shape
    N, C2, oh, ow = out.shape
    assert C2 == C and tuple(rows.shape) == (N, oh) and tuple(cols.shape) == (N, ow) and src_n.numel() == N
    _call("ed_gather2d", _dev(inp, None, "inp"), _code(inp, "inp"), _dev(out, None, "out"), _code(out, "out"),
          C, H, W, _dev(src_n, torch.int32), _dev(rows, torch.int32), _dev(cols, torch.int32),
          N, oh, ow, _stream())
    return out


def tile_gather_pad(latent, tiles, tile_y0, tile_x0, scaling_factor):
    """latent f32 [B,C,H,W] -> tiles [(T*B),C,Ts,Ts] = zero-haloed windows / scaling_factor."""
    B, C, H, W = latent.shape
    T = tile_y0.numel()
    rows, C2, Ts, Ts2 = tiles.shape
    assert rows == T * B and C2 == C and Ts == Ts2
    _call("ed_tile_gather_pad", _dev(latent, torch.float32), _dev(tiles, None, "tiles"), _code(tiles, "tiles"),
          B, C, H, W, _dev(tile_y0, torch.int32), _dev(tile_x0, torch.int32), T, Ts,
          float(scaling_factor), _stream())
    return tiles


TILE_MAXC = 4


def tile_accumulate_normalise(decoded, image, n_col_tiles, row_tile, row_src, col_tile, col_src):
    """decoded [(T*B),3,TP,TP] raw VAE output -> image f32 [B,3,HP,WP] = mean over covering tiles of clamp(v/2+.5)."""
    B, Cimg, HP, WP = image.shape
    rows, C2, TP, TP2 = decoded.shape
    assert C2 == Cimg and TP == TP2 and rows % B == 0
    assert row_tile.numel() == HP * TILE_MAXC == row_src.numel() and col_tile.numel() == WP * TILE_MAXC == col_src.numel()
    _call("ed_tile_accumulate_normalise", _dev(decoded, None, "decoded"), _code(decoded, "decoded"),
          _dev(image, torch.float32), B, Cimg, HP, WP, TP, n_col_tiles,
          _dev(row_tile, torch.int32), _dev(row_src, torch.int32),
          _dev(col_tile, torch.int32), _dev(col_src, torch.int32), _stream())
    return image


# ---- fused kernels inside the UNet (csrc/unet_kernels.hip) ---------------------------------------------------------
def geglu(x2, inner):
    """x2 [..., 2*inner] (16-bit, contiguous) -> [..., inner] = x2[..., :inner] * gelu(x2[..., inner:])."""
    assert x2.shape[-1] == 2 * inner
    out = torch.empty(x2.shape[:-1] + (inner,), dtype=x2.dtype, device=x2.device)
    TIMER.note_work("ed_geglu", nbytes=1.5 * x2.numel() * x2.element_size())  # read 2I, write I
    _call("ed_geglu", _dev(x2, None, "x2"), _dev(out, None, "out"), _code(x2, "x2"), x2.numel() // (2 * inner), inner,
          _stream())
    return out


# ---- dense contractions (csrc/gemm_kernels.hip) ---------------------------------------------------------------------
GEMM_BM, GEMM_BN = 256, 128   # rows / value columns of one workgroup tile (256 output columns for the plain epilogue)
GEMM_MIN_BLOCKS = 96          # below this many workgroups the 256-CU chip is mostly idle: the library call stays
GEMM_CUS = 256                # one 128-KiB-LDS workgroup per CU: the grid runs in rounds of 256 tiles
GEMM_MIN_ROUND_FILL = 0.7     # a multi-round grid whose rounds are on average emptier than this loses to the library's stream-K


GEMM_ROWS_TILE_COST = 0.72    # csrc/gemm_kernels.hip ROWS_TILE_COST: a round of 128-row tiles relative to a round of 256-row ones
GEMM_ROWS_MIN_BLOCKS_LINEAR = 200   # ... for projections (short kernels: the in-situ measurement, linear_wins)
GEMM_ROWS_MIN_BLOCKS = 80      # 128-row tiles below which even the half-height grid leaves the chip mostly idle (SD 1.5's 8 x 8 level: 50)


def gemm_rows_mode(M, n_col_blocks):
    """Does the launcher run this grid as 128-row tiles (csrc/gemm_kernels.hip, launch(): fewer effective rounds)?  The mirror of its rule."""
    b256, b128 = -(-M // GEMM_BM) * n_col_blocks, -(-M // (GEMM_BM // 2)) * n_col_blocks
    return -(-b128 // GEMM_CUS) * GEMM_ROWS_TILE_COST < -(-b256 // GEMM_CUS) - 1e-9


def _gemm_grid_ok(blocks, M=None, n_col_blocks=None, min_fill=None):
    """Is a grid of ``blocks`` 256 x 256 tiles worth launching?  Measured (profiles/r4_s2_probe_gemm_*.jsonl): 25 tiles lose
    2.4 x, 100-120 tiles win 1.06-1.18 x (the library under-fills the chip as well), 288 tiles = 2 rounds, the second 12 %
    full, lose 0.83-0.90 x; 400 tiles (78 %) and everything fuller win.  Round 6: an under-filled grid runs as 128-row tiles when that
    takes fewer effective rounds (``gemm_rows_mode``; profiles/r6_s4_gemm_rows_mode.jsonl: 120 tiles 1.35-1.45 x faster than as 256-row
    tiles -- 1.95 x MIOpen on the batch-6 32 x 32 convolutions -- and 20-60 tiles, the 1- / 3-row per-rank forwards of the multi-GPU layout,
    1.24-1.93 x the library call), so such a grid is taken from GEMM_ROWS_MIN_BLOCKS half-height tiles on."""
    if M is not None and n_col_blocks is not None and gemm_rows_mode(M, n_col_blocks):
        return -(-M // (GEMM_BM // 2)) * n_col_blocks >= (GEMM_ROWS_MIN_BLOCKS if GEMM_MIN_BLOCKS > 1 else 1)
    if blocks < GEMM_MIN_BLOCKS:
        return False
    rounds = -(-blocks // GEMM_CUS)
    return rounds == 1 or blocks / (rounds * GEMM_CUS) >= (GEMM_MIN_ROUND_FILL if min_fill is None else min_fill)


def _gemm_x(x, name):
    """A [..., K] activation as the kernel sees it: M rows of K contiguous 16-bit values."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype not in (torch.float16, torch.bfloat16):
        _reject(f"{name} must be a 16-bit tensor on the MI355X; no CPU fallback")
    if not x.is_contiguous():
        _reject(f"{name} must be contiguous")
    return x.numel() // x.shape[-1], x.shape[-1]


def geglu_gemm_ok(M, K, I):
    return K % 64 == 0 and K >= 64 and I % GEMM_BN == 0 and M * K * 2 < 2 ** 31 - 16 and 2 * I * K * 2 < 2 ** 31 - 16


def geglu_gemm_wins(M, K, I):
    """Where the model uses ed_geglu_gemm: every shape the kernel takes whose grid fills the chip.  The fused pair beats hipBLASLt +
    ed_geglu by 1.11-1.91 x on full grids, but it is the same main loop as ed_linear, which loses 2.4 x on a 25-tile grid
    (one 128-KiB-LDS workgroup per CU): a batch-1 / batch-2 forward at the 16 x 16 or 32 x 32 level stays with the library
    (ADVICE r4).  The GEGLU kernel is persistent (one workgroup per CU walking the tiles), so there is no round-fill condition."""
    return geglu_gemm_ok(M, K, I) and -(-M // GEMM_BM) * (I // GEMM_BN) >= GEMM_MIN_BLOCKS


def geglu_gemm(x, w, bias=None):
    """x [..., K], w [2I, K], bias [2I] -> [..., I] = (x w_v^T + b_v) * gelu(x w_g^T + b_g).  See ed_geglu_gemm."""
    M, K = _gemm_x(x, "x")
    I = w.shape[0] // 2
    if tuple(w.shape) != (2 * I, K) or not geglu_gemm_ok(M, K, I):
        _reject(f"geglu_gemm: unsupported shape x {tuple(x.shape)} w {tuple(w.shape)}")
    out = torch.empty(x.shape[:-1] + (I,), dtype=x.dtype, device=x.device)
    TIMER.note_work("ed_geglu_gemm", flops=4.0 * M * K * I, nbytes=2.0 * (M * K + 2 * I * K + M * I))
    _call("ed_geglu_gemm", _dev(x, None, "x"), _dev(w, x.dtype, "w"), _opt(bias, x.dtype, "bias"), _dev(out, None, "out"),
          _code(x, "x"), M, K, I, _stream())
    return out


def linear_ok(M, K, N):
    """what the kernel takes (linear_wins: where the model uses it)"""
    return K % 64 == 0 and K >= 64 and N % 8 == 0 and M * K * 2 < 2 ** 31 - 16 and N * K * 2 < 2 ** 31 - 16


def linear_wins(M, K, N):
    """Where ed_linear measured faster than hipBLASLt on the MI355X in fp16, over every projection shape of the SDXL (batch 20
    and 6) and SD1.5 (batch 20) forwards (profiles/r4_s2_probe_gemm_*.jsonl): hipBLASLt is weak when K <= 640 or N <= 640
    (388-900 TFLOP/s; this kernel 1.09-1.58 x) and strong on the K >= 1280, N >= 1280 projections (930-1320 TFLOP/s; this
    kernel 0.70-0.95 x there), whenever the grid fills the chip (_gemm_grid_ok)."""
    ncb = -(-N // (2 * GEMM_BN))
    if linear_ok(M, K, N) and K <= 1280 and gemm_rows_mode(M, ncb):
        # round 6: an under-filled grid as 128-row tiles beats the library up to K = 1280 whatever N in an isolated loop (6144 x 1280 -> 1280:
        # 1.06 x, 3072 x 1280 -> 1280: 1.27 x, 4096 x 640 -> 640: 1.40 x; K = 5120 does not -- the library splits K) -- but IN the forward only the
        # batch-6 shapes keep that (240 half tiles: + 0.35 %); the 40-120-tile grids of the 3- and 1-row forwards measured 1.27-1.40 x alone
        # cost those forwards 4.0 % and 2.1 % (profiles/r6_s7_policy_split.jsonl: a 128-KiB-LDS workgroup cannot start beside the
        # previous kernel's tail, the library's small tiles can), so projections are taken from GEMM_ROWS_MIN_BLOCKS_LINEAR half tiles on
        return -(-M // (GEMM_BM // 2)) * ncb >= GEMM_ROWS_MIN_BLOCKS_LINEAR
    blocks = -(-M // GEMM_BM) * ncb
    if K > 1280 and blocks < 200:      # long K on a grid well under one round: the library splits K (12288 x 2560 -> 640, 144 tiles: 0.80 x)
        return False
    return linear_ok(M, K, N) and (K <= 640 or N <= 640) and _gemm_grid_ok(blocks)


def linear(x, w, bias=None, residual=None):
    """x [..., K], w [N, K] -> x w^T + bias (+ residual [..., N]).  See ed_linear."""
    M, K = _gemm_x(x, "x")
    N = w.shape[0]
    if tuple(w.shape) != (N, K) or not linear_ok(M, K, N):
        _reject(f"linear: unsupported shape x {tuple(x.shape)} w {tuple(w.shape)}")
    out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    if residual is not None and (residual.shape != out.shape or residual.dtype != x.dtype):
        _reject("linear: residual must match the output")
    TIMER.note_work("ed_linear", flops=2.0 * M * K * N, nbytes=2.0 * (M * K + N * K + M * N * (2 if residual is not None else 1)))
    _call("ed_linear", _dev(x, None, "x"), _dev(w, x.dtype, "w"), _opt(bias, x.dtype, "bias"), _opt(residual, x.dtype, "residual"),
          _dev(out, None, "out"), _code(x, "x"), M, K, N, _stream())
    return out


def conv3x3_ok(B, H, W, Cin, N):
    """what the kernel takes (conv3x3_wins: where the model uses it)"""
    M = B * H * W
    return Cin % 64 == 0 and N % 8 == 0 and M * Cin * 2 < 2 ** 31 - 16 and N * 9 * Cin * 2 < 2 ** 31 - 16


def conv3x3_wins(B, H, W, Cin, N):
    """Where ed_conv3x3_nhwc measured faster than MIOpen's CK kernels (fp16, profiles/r4_s2_probe_gemm_*.jsonl): every
    ResnetBlock / upsampler shape whose grid fills the chip -- 1.10-1.59 x at batch 20, 1.06-1.49 x at batch 6 except the
    288-tile 64 x 64 x 640 shapes (0.83-0.90 x in round 4, 1.17 x since round 6), 0.4 x on SD1.5's 25-tile 8 x 8 level."""
    ncb = -(-N // (2 * GEMM_BN))
    # round 6: with the long-K loop, half column tiles and the channel-block-major walk the 288-tile 64 x 64 x 640 shapes of the batch-6
    # forward now measure 1.17 x MIOpen (profiles/r6_s4_gemm_rows_mode.jsonl; round 4: 0.83-0.90 x): no round-fill condition for convolutions
    return conv3x3_ok(B, H, W, Cin, N) and _gemm_grid_ok(-(-(B * H * W) // GEMM_BM) * ncb, B * H * W, ncb, min_fill=0.5)


def conv3x3_nhwc(x, w, bias=None, sample_bias=None, residual=None):
    """x [B,Cin,H,W] and w [N,Cin,3,3] in torch.channels_last memory format (16-bit) -> conv2d(x, w, stride 1, padding 1)
    + bias[n] + sample_bias[b, n] + residual, channels_last.  See ed_conv3x3_nhwc."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16)):
        _reject("conv3x3_nhwc: x must be a 16-bit [B,C,H,W] tensor on the MI355X; no CPU fallback")
    B, Cin, H, W = x.shape
    N = w.shape[0]
    cl = torch.channels_last
    if tuple(w.shape) != (N, Cin, 3, 3) or w.dtype != x.dtype or not conv3x3_ok(B, H, W, Cin, N):
        _reject(f"conv3x3_nhwc: unsupported shape x {tuple(x.shape)} w {tuple(w.shape)}")
    if not x.is_contiguous(memory_format=cl) or not w.is_contiguous(memory_format=cl):
        _reject("conv3x3_nhwc: x and w must be channels_last")
    out = torch.empty((B, N, H, W), dtype=x.dtype, device=x.device, memory_format=cl)
    if residual is not None and (residual.shape != out.shape or residual.dtype != x.dtype or not residual.is_contiguous(memory_format=cl)):
        _reject("conv3x3_nhwc: residual must be a channels_last tensor of the output's shape")
    if sample_bias is not None and tuple(sample_bias.shape) != (B, N):
        _reject("conv3x3_nhwc: sample_bias must be [B, N]")
    TIMER.note_work("ed_conv3x3_nhwc", flops=2.0 * B * H * W * 9 * Cin * N,
                    nbytes=2.0 * (B * H * W * (Cin + N * (2 if residual is not None else 1)) + 9 * Cin * N))
    BA = conv3x3_batch_split(B, H, W, N) if CONV_BATCH_SPLIT else None
    for b0, nb in (((0, B),) if BA is None else ((0, BA), (BA, B - BA))):
        # (the tail of a batch whose grid ends in a nearly empty round of 256-row tiles is launched on its own: the launcher runs it as 128-row tiles)
        px, po = b0 * H * W * Cin * x.element_size(), b0 * H * W * N * x.element_size()
        _call("ed_conv3x3_nhwc", x.data_ptr() + px, w.data_ptr(), _opt(bias, x.dtype, "bias"),
              None if sample_bias is None else _dev(sample_bias, x.dtype, "sample_bias") + b0 * N * x.element_size(),
              None if residual is None else residual.data_ptr() + po, out.data_ptr() + po, _code(x, "x"), nb, H, W, Cin, N, _stream_of(x))
    return out


CONV_BATCH_SPLIT = True


def conv3x3_batch_split(B, H, W, N):
    """Samples of a batch ed_conv3x3_nhwc should run in a first launch (the rest in a second), or None for one launch.  One 128-KiB-LDS
    workgroup per CU: a grid runs in rounds of 256 tiles, and the SDXL forwards of the default bench end some of them badly -- the 32 x 32
    convolutions at 40 rows are 800 tiles: three full rounds and a fourth 12.5 % full.  Splitting the batch at a SAMPLE boundary (whole images:
    the pixel geometry of either part is unchanged, and so is every output bit) lets the launcher run the short tail as 128-row tiles (0.72 of a
    round): 3.72 rounds instead of 4.  Taken when the launcher's own cost model says the two launches finish at least 0.1 round sooner."""
    HW = H * W
    if HW % GEMM_BM or B < 2:
        return None
    ncb = -(-N // (2 * GEMM_BN))
    tps = HW // GEMM_BM * ncb                      # 256-row tiles per sample
    total = B * tps

    def rounds(tiles):                             # the launcher's choice for a grid of `tiles` full tiles (csrc/gemm_kernels.hip, launch())
        return min(-(-tiles // GEMM_CUS), -(-2 * tiles // GEMM_CUS) * GEMM_ROWS_TILE_COST)

    full = total // GEMM_CUS
    # measured (profiles/r6_s25_*): 800 tiles (3 full rounds + 32 tiles) +5.6...5.9 % per convolution, bit-identical, +0.23 % of the 40-row forward;
    # 576 tiles (2 + 64) -4 % and 288 (1 + 32) +2.6 % alone, nothing in the forward: the model is only trusted from three full rounds on
    if full < 3 or total % GEMM_CUS == 0:
        return None
    BA = full * GEMM_CUS // tps
    if BA <= 0 or BA >= B:
        return None
    return BA if -(-BA * tps // GEMM_CUS) + rounds((B - BA) * tps) < rounds(total) - 0.1 else None


def conv3x3_up2x_wins(B, H, W, Cin, N):
    """Where Upsample2D runs as ONE ed_conv3x3_nhwc_up2x launch (H, W = the OUTPUT size): wherever the convolution on the materialised
    upsampling would run as 256-row tiles of ed_conv3x3_nhwc (the fused kernel has no 128-row instantiation: the upsamplers' grids are
    80+ tiles per row of the batch) -- the upsampled tensor is then never written or read back (4x the source)."""
    ncb = -(-N // (2 * GEMM_BN))
    return (H % 2 == 0 and W % 2 == 0 and conv3x3_wins(B, H, W, Cin, N) and not gemm_rows_mode(B * H * W, ncb)
            and B * (H // 2) * (W // 2) * Cin * 2 < 2 ** 31 - 16)


def conv3x3_nhwc_up2x(x, w, bias=None):
    """x [B,Cin,H/2,W/2] and w [N,Cin,3,3] channels_last 16-bit -> conv2d(nearest_upsample_2x(x), w, stride 1, padding 1) + bias as
    channels_last [B,N,H,W]; the upsampled tensor never exists.  See ed_conv3x3_nhwc_up2x."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16)):
        _reject("conv3x3_nhwc_up2x: x must be a 16-bit [B,C,H,W] tensor on the MI355X; no CPU fallback")
    B, Cin, Hs, Ws = x.shape
    H, W = 2 * Hs, 2 * Ws
    N = w.shape[0]
    cl = torch.channels_last
    if tuple(w.shape) != (N, Cin, 3, 3) or w.dtype != x.dtype or not conv3x3_ok(B, H, W, Cin, N):
        _reject(f"conv3x3_nhwc_up2x: unsupported shape x {tuple(x.shape)} w {tuple(w.shape)}")
    if not x.is_contiguous(memory_format=cl) or not w.is_contiguous(memory_format=cl):
        _reject("conv3x3_nhwc_up2x: x and w must be channels_last")
    out = torch.empty((B, N, H, W), dtype=x.dtype, device=x.device, memory_format=cl)
    TIMER.note_work("ed_conv3x3_nhwc_up2x", flops=2.0 * B * H * W * 9 * Cin * N, nbytes=2.0 * (B * Hs * Ws * Cin + B * H * W * N + 9 * Cin * N))
    _call("ed_conv3x3_nhwc_up2x", x.data_ptr(), w.data_ptr(), _opt(bias, x.dtype, "bias"), out.data_ptr(), _code(x, "x"), B, H, W, Cin, N,
          _stream_of(x))
    return out


def conv3x3_s2_wins(B, H, W, Cin, N):
    """Where Downsample2D runs as ed_conv3x3_nhwc_s2 (H, W = the OUTPUT size): full grids of 256-row tiles, as for the stride-1 kernel
    (no 128-row instantiation)."""
    ncb = -(-N // (2 * GEMM_BN))
    return (conv3x3_wins(B, H, W, Cin, N) and not gemm_rows_mode(B * H * W, ncb) and 4 * B * H * W * Cin * 2 < 2 ** 31 - 16)


def conv3x3_nhwc_s2(x, w, bias=None):
    """x [B,Cin,2H,2W] and w [N,Cin,3,3] channels_last 16-bit -> conv2d(x, w, stride 2, padding 1) + bias as channels_last [B,N,H,W].
    See ed_conv3x3_nhwc_s2."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16)):
        _reject("conv3x3_nhwc_s2: x must be a 16-bit [B,C,H,W] tensor on the MI355X; no CPU fallback")
    B, Cin, Hi, Wi = x.shape
    if Hi % 2 or Wi % 2:
        _reject("conv3x3_nhwc_s2: even input height and width")
    H, W = Hi // 2, Wi // 2
    N = w.shape[0]
    cl = torch.channels_last
    if tuple(w.shape) != (N, Cin, 3, 3) or w.dtype != x.dtype or not conv3x3_ok(B, Hi, Wi, Cin, N):
        _reject(f"conv3x3_nhwc_s2: unsupported shape x {tuple(x.shape)} w {tuple(w.shape)}")
    if not x.is_contiguous(memory_format=cl) or not w.is_contiguous(memory_format=cl):
        _reject("conv3x3_nhwc_s2: x and w must be channels_last")
    out = torch.empty((B, N, H, W), dtype=x.dtype, device=x.device, memory_format=cl)
    TIMER.note_work("ed_conv3x3_nhwc_s2", flops=2.0 * B * H * W * 9 * Cin * N, nbytes=2.0 * (B * Hi * Wi * Cin + B * H * W * N + 9 * Cin * N))
    _call("ed_conv3x3_nhwc_s2", x.data_ptr(), w.data_ptr(), _opt(bias, x.dtype, "bias"), out.data_ptr(), _code(x, "x"), B, H, W, Cin, N,
          _stream_of(x))
    return out


GROUPNORM_SPLIT = True  # large groups: statistics + apply as two fully parallel launches (A/B switch)


def groupnorm(x, gamma, beta, groups, eps, silu=False, tokens=False, chan_bias=None, conv_bias=None):
    """x [N,C,H,W] NCHW 16-bit -> GroupNorm(+SiLU) as [N,C,H,W], or [N,H*W,C] when ``tokens``.
    ``conv_bias`` [C] / ``chan_bias`` [N,C] (optional): normalise round16(round16(x + conv_bias) + chan_bias) instead of
    x (each add only when given) -- see ed_groupnorm."""
    N, C, H, W = x.shape
    if chan_bias is not None:
        assert tuple(chan_bias.shape) == (N, C)
    if conv_bias is not None:
        assert tuple(conv_bias.shape) == (C,)
    out = torch.empty((N, H * W, C) if tokens else (N, C, H, W), dtype=x.dtype, device=x.device)
    TIMER.note_work("ed_groupnorm", nbytes=3.0 * x.numel() * x.element_size())  # statistics pass + apply pass + write
    ws = None
    if GROUPNORM_SPLIT:
        nbytes = _hip.lib().ed_groupnorm_workspace(N, C, H * W, groups)
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device) if nbytes else None
    _call("ed_groupnorm", _dev(x, None, "x"), _dev(gamma, x.dtype, "gamma"), _dev(beta, x.dtype, "beta"),
          _opt(conv_bias, x.dtype, "conv_bias"), _opt(chan_bias, x.dtype, "chan_bias"), _dev(out, None, "out"),
          _opt(ws, torch.float32, "workspace"), _code(x, "x"), N, C, H * W, groups, float(eps),
          int(silu), int(tokens), _stream())
    return out


def groupnorm_nhwc(x, gamma, beta, groups, eps, silu=False, chan_bias=None, conv_bias=None):
    """x [N,C,H,W] in torch.channels_last memory format (16-bit) -> GroupNorm(+SiLU), same format; optional
    ``conv_bias`` [C] / ``chan_bias`` [N,C] as in ``groupnorm``."""
    N, C, H, W = x.shape
    if not x.is_cuda:
        _reject("x must be a tensor on the MI355X; no CPU fallback")
    if not x.is_contiguous(memory_format=torch.channels_last):
        _reject("x must be channels_last")
    out = torch.empty_like(x, memory_format=torch.channels_last)
    nbytes = _hip.lib().ed_groupnorm_nhwc_workspace(N, C, H * W, groups)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    TIMER.note_work("ed_groupnorm_nhwc", nbytes=3.0 * x.numel() * x.element_size())
    _call("ed_groupnorm_nhwc", x.data_ptr(), _dev(gamma, x.dtype, "gamma"), _dev(beta, x.dtype, "beta"),
          _opt(conv_bias, x.dtype, "conv_bias"), _opt(chan_bias, x.dtype, "chan_bias"), out.data_ptr(),
          _dev(ws, torch.float32, "workspace"), _code(x, "x"), N, C, H * W, groups, float(eps), int(silu), _stream())
    return out


def groupnorm_nhwc_cat_ok(x1, x2, groups):
    """can ed_groupnorm_nhwc_cat normalise cat([x1, x2], 1) of these two tensors in place?"""
    def cl16(x):
        return (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.bfloat16)
                and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last))
    if not (cl16(x1) and cl16(x2) and x1.dtype == x2.dtype and x1.device == x2.device and x1.shape[0] == x2.shape[0]
            and x1.shape[2:] == x2.shape[2:]):
        return False
    C1, C = x1.shape[1], x1.shape[1] + x2.shape[1]
    return C1 % 8 == 0 and C % 8 == 0 and C % groups == 0 and C // groups >= 8 and groups <= 256 and C <= 4096


def groupnorm_nhwc_cat(x1, x2, gamma, beta, groups, eps, silu=False):
    """GroupNorm(+SiLU) of torch.cat([x1, x2], dim=1) for two channels_last 16-bit tensors [N,C1,H,W], [N,C2,H,W], read in place
    (the concatenation is never written) -> channels_last [N, C1 + C2, H, W].  See ed_groupnorm_nhwc_cat."""
    if not groupnorm_nhwc_cat_ok(x1, x2, groups):
        _reject("groupnorm_nhwc_cat: two channels_last 16-bit tensors of the same dtype / batch / size on the MI355X, C1 % 8 == 0, "
                "(C1 + C2) / groups >= 8; no CPU fallback")
    N, C1, H, W = x1.shape
    C2 = x2.shape[1]
    out = torch.empty((N, C1 + C2, H, W), dtype=x1.dtype, device=x1.device, memory_format=torch.channels_last)
    nbytes = _hip.lib().ed_groupnorm_nhwc_workspace(N, C1 + C2, H * W, groups)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x1.device)
    TIMER.note_work("ed_groupnorm_nhwc_cat", nbytes=3.0 * out.numel() * out.element_size())
    _LAUNCH["device"] = x1.device
    _call("ed_groupnorm_nhwc_cat", x1.data_ptr(), x2.data_ptr(), _dev(gamma, x1.dtype, "gamma"), _dev(beta, x1.dtype, "beta"),
          out.data_ptr(), ws.data_ptr(), _DTYPE[x1.dtype], N, C1, C2, H * W, groups, float(eps), int(silu), _stream())
    return out


def layernorm(x, gamma, beta, eps):
    """x [..., D] contiguous 16-bit -> LayerNorm over D."""
    D = x.shape[-1]
    out = torch.empty_like(x)
    TIMER.note_work("ed_layernorm", nbytes=2.0 * x.numel() * x.element_size())
    _call("ed_layernorm", _dev(x, None, "x"), _dev(gamma, x.dtype, "gamma"), _dev(beta, x.dtype, "beta"),
          _dev(out, None, "out"), _code(x, "x"), x.numel() // D, D, float(eps), _stream())
    return out


def add_layernorm(a, b, gamma, beta, eps):
    """a, b [..., D] contiguous 16-bit -> (a + b, LayerNorm(a + b)); the sum is rounded to the 16-bit type first."""
    D = a.shape[-1]
    assert a.shape == b.shape and a.dtype == b.dtype
    s, out = torch.empty_like(a), torch.empty_like(a)
    TIMER.note_work("ed_add_layernorm", nbytes=4.0 * a.numel() * a.element_size())
    _call("ed_add_layernorm", _dev(a, None, "a"), _dev(b, a.dtype, "b"), _dev(gamma, a.dtype, "gamma"),
          _dev(beta, a.dtype, "beta"), _dev(s, None, "sum"), _dev(out, None, "out"), _code(a, "a"), a.numel() // D, D,
          float(eps), _stream())
    return s, out


# ---- the fp32-residual-stream mode (models.UNet2DConditionModel.residual_fp32): the layers that read / write the stream ----------------
def _stream_ok(x, w):
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and w.dtype in (torch.float16, torch.bfloat16)
            and w.device == x.device)


def layernorm_s32(x, gamma, beta, eps):
    """x [..., D] contiguous fp32 (the residual stream) -> LayerNorm over D in gamma's 16-bit dtype (ed_layernorm_s32)."""
    if not _stream_ok(x, gamma) or not x.is_contiguous():
        _reject("layernorm_s32: x must be a contiguous fp32 tensor on the MI355X and gamma 16-bit; no CPU fallback")
    D = x.shape[-1]
    out = torch.empty(x.shape, dtype=gamma.dtype, device=x.device)
    TIMER.note_work("ed_layernorm_s32", nbytes=x.numel() * 6.0)
    _LAUNCH["device"] = x.device
    _call("ed_layernorm_s32", x.data_ptr(), _dev(gamma, None, "gamma"), _dev(beta, gamma.dtype, "beta"), out.data_ptr(),
          _DTYPE[gamma.dtype], x.numel() // D, D, float(eps), _stream())
    return out


def add_layernorm_s32(a, b, gamma, beta, eps):
    """a [..., D] 16-bit (a branch result), b [..., D] fp32 (the residual stream) -> (a + b in fp32, LayerNorm(a + b) in a's dtype)."""
    if not (_stream_ok(b, a) and a.shape == b.shape and a.is_contiguous() and b.is_contiguous() and gamma.dtype == a.dtype):
        _reject("add_layernorm_s32: a must be 16-bit and b fp32, same shape, contiguous, on the MI355X; no CPU fallback")
    D = a.shape[-1]
    s, out = torch.empty_like(b), torch.empty_like(a)
    TIMER.note_work("ed_add_layernorm_s32", nbytes=a.numel() * 12.0)
    _LAUNCH["device"] = a.device
    _call("ed_add_layernorm_s32", a.data_ptr(), b.data_ptr(), _dev(gamma, None, "gamma"), _dev(beta, a.dtype, "beta"), s.data_ptr(),
          out.data_ptr(), _DTYPE[a.dtype], a.numel() // D, D, float(eps), _stream())
    return s, out


def groupnorm_nhwc_s32(x, gamma, beta, groups, eps, silu=False):
    """x [N,C,H,W] fp32 in channels_last memory (the residual stream) -> GroupNorm(+SiLU) in gamma's 16-bit dtype, channels_last."""
    if not _stream_ok(x, gamma) or x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last):
        _reject("groupnorm_nhwc_s32: x must be an fp32 channels_last [N,C,H,W] tensor on the MI355X and gamma 16-bit; no CPU fallback")
    N, C, H, W = x.shape
    out = torch.empty((N, C, H, W), dtype=gamma.dtype, device=x.device, memory_format=torch.channels_last)
    nbytes = _hip.lib().ed_groupnorm_nhwc_workspace(N, C, H * W, groups)
    ws = torch.empty(max(4, nbytes // 4), dtype=torch.float32, device=x.device)
    TIMER.note_work("ed_groupnorm_nhwc_s32", nbytes=x.numel() * 10.0)
    _LAUNCH["device"] = x.device
    _call("ed_groupnorm_nhwc_s32", x.data_ptr(), _dev(gamma, None, "gamma"), _dev(beta, gamma.dtype, "beta"), out.data_ptr(),
          ws.data_ptr(), _DTYPE[gamma.dtype], N, C, H * W, groups, float(eps), int(silu), _stream())
    return out


def bias_residual_add(h, h_bias, res, res_bias=None):
    """round16(res (+ res_bias[c])) + round16(h + h_bias[c]) for 16-bit [N,C,H,W] tensors that are both NCHW-contiguous
    or both channels_last (ResnetBlock2D's closing add); the result has the same memory format."""
    N, C, H, W = h.shape
    assert res.shape == h.shape and res.dtype == h.dtype
    if not (h.is_cuda and res.is_cuda):
        _reject("bias_residual_add: h and res must be tensors on the MI355X; no CPU fallback")
    cl = (not h.is_contiguous()) and h.is_contiguous(memory_format=torch.channels_last) and C % 8 == 0
    fmt = torch.channels_last if cl else torch.contiguous_format
    # mixed layouts only arise on fallback paths (a torch GroupNorm returning NCHW inside a channels-last model): follow
    # the convolution output's layout, re-laying-out the other operand
    h = h.contiguous(memory_format=fmt)
    res = res.contiguous(memory_format=fmt)
    out = torch.empty_like(h, memory_format=fmt)
    TIMER.note_work("ed_bias_residual_add", nbytes=3.0 * h.numel() * h.element_size())
    _call("ed_bias_residual_add", h.data_ptr(), _opt(h_bias, h.dtype, "h_bias"), res.data_ptr(),
          _opt(res_bias, h.dtype, "res_bias"), out.data_ptr(), _code(h, "h"), N, C, H * W, int(cl), _stream_of(h))
    return out


def tokens_add_nchw(x, tokens):
    """x [N,C,H,W] + tokens [N,H*W,C] (16-bit, contiguous) -> [N,C,H,W]: the residual add closing a transformer."""
    N, C, H, W = x.shape
    assert tuple(tokens.shape) == (N, H * W, C) and tokens.dtype == x.dtype
    out = torch.empty_like(x)
    TIMER.note_work("ed_tokens_add_nchw", nbytes=3.0 * x.numel() * x.element_size())
    _call("ed_tokens_add_nchw", _dev(x, None, "x"), _dev(tokens, None, "tokens"), _dev(out, None, "out"), _code(x, "x"),
          N, C, H * W, _stream())
    return out


def groupnorm_f32(x, gamma, beta, groups, eps, silu=False):
    """x fp32 [N,C,H,W] NCHW-contiguous -> GroupNorm(+SiLU), fp32 (the VAE's normalisation layers; ed_groupnorm_f32)."""
    N, C, H, W = x.shape
    if (H * W) % 4:
        _reject("groupnorm_f32: H*W must be a multiple of 4")
    out = torch.empty_like(x)
    nbytes = _hip.lib().ed_groupnorm_f32_workspace(N, C, H * W, groups)
    ws = torch.empty(max(1, nbytes // 4), dtype=torch.float32, device=x.device)
    TIMER.note_work("ed_groupnorm_f32", nbytes=3.0 * x.numel() * 4)
    _call("ed_groupnorm_f32", _dev(x, torch.float32, "x"), _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta"),
          _dev(out, torch.float32, "out"), _dev(ws, torch.float32, "workspace"), N, C, H * W, groups, float(eps), int(silu),
          _stream())
    return out


# ---- the fp32 VAE's ResnetBlock convolutions on split fp16 operands (csrc/vae_kernels.hip, gemm_kernels.hip OUT32) ------------------
def groupnorm_nhwc_f32_ok(C, groups):
    return C % groups == 0 and (C // groups) % 4 == 0 and groups <= 256 and C <= 4096


def groupnorm_nhwc_f32(x, gamma, beta, groups, eps, silu=False, split=False):
    """x [N,C,H,W] fp32 in channels_last memory -> GroupNorm(+SiLU) in fp32, returned as fp32 channels_last [N,C,H,W] or, with
    ``split``, as the fp16 channels_last [N,3C,H,W] = [hi | lo | hi] operand of ``conv3x3_f32out`` (hi = fp16(y), lo = fp16(y - hi)).
    See ed_groupnorm_nhwc_f32."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        _reject("groupnorm_nhwc_f32: x must be an fp32 [N,C,H,W] tensor on the MI355X; no CPU fallback")
    N, C, H, W = x.shape
    cl = torch.channels_last
    if not x.is_contiguous(memory_format=cl):
        _reject("groupnorm_nhwc_f32: x must be channels_last")
    if not groupnorm_nhwc_f32_ok(C, groups):
        _reject(f"groupnorm_nhwc_f32: unsupported C = {C}, groups = {groups}")
    out = (torch.empty((N, 3 * C, H, W), dtype=torch.float16, device=x.device, memory_format=cl) if split
           else torch.empty_like(x, memory_format=cl))
    nbytes = _hip.lib().ed_groupnorm_nhwc_f32_workspace(N, C, H * W, groups)
    ws = torch.empty(max(4, nbytes // 4), dtype=torch.float32, device=x.device)
    TIMER.note_work("ed_groupnorm_nhwc_f32", nbytes=x.numel() * (4.0 + 4.0 + (6.0 if split else 4.0)))
    _call("ed_groupnorm_nhwc_f32", x.data_ptr(), _dev(gamma, torch.float32, "gamma"), _dev(beta, torch.float32, "beta"), out.data_ptr(),
          _dev(ws, torch.float32, "workspace"), N, C, H * W, groups, float(eps), int(silu), int(split), _stream_of(x))
    return out


def absmax_f32(x):
    """max |x| of a contiguous-in-memory fp32 tensor as a 1-element device tensor (ed_absmax_f32): the per-tensor figure the raw-stream
    split scales by.  No host synchronisation."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32):
        _reject("absmax_f32: x must be an fp32 tensor on the MI355X; no CPU fallback")
    if not (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))):
        _reject("absmax_f32: x must be dense (contiguous or channels_last)")
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    TIMER.note_work("ed_absmax_f32", nbytes=x.numel() * 4.0)
    _LAUNCH["device"] = x.device
    _call("ed_absmax_f32", x.data_ptr(), x.numel(), out.data_ptr(), _stream())
    return out


def split_f32(x, upsample2x=False, absmax=None):
    """x [N,C,H,W] fp32 in channels_last memory -> the fp16 channels_last [N,3C,UH,UW] = [hi | lo | hi] operand of ``conv3x3_f32out`` for a
    raw (un-normalised) activation, U = 2 with ``upsample2x`` (nearest-neighbour upsampling folded in).  ``absmax`` (``absmax_f32(x)``, a
    1-element device tensor): the activation is scaled by 2^-e, e = max(0, exponent(absmax) - 14), before the split -- exact over the
    whole fp32 range; hand the SAME tensor to ``conv3x3_f32out(act_absmax=)``, which multiplies by 2^e.  Without it hi and lo saturate at
    fp16's largest finite value (exact up to 65504).  See ed_split_f32_nhwc."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32):
        _reject("split_f32: x must be an fp32 [N,C,H,W] tensor on the MI355X; no CPU fallback")
    if absmax is not None and not (isinstance(absmax, torch.Tensor) and absmax.device == x.device and absmax.dtype == torch.float32
                                   and absmax.numel() == 1):
        _reject("split_f32: absmax must be a 1-element fp32 tensor on x's device")
    N, C, H, W = x.shape
    cl = torch.channels_last
    if not x.is_contiguous(memory_format=cl) or C % 4:
        _reject("split_f32: x must be channels_last with C % 4 == 0")
    u = 2 if upsample2x else 1
    out = torch.empty((N, 3 * C, u * H, u * W), dtype=torch.float16, device=x.device, memory_format=cl)
    TIMER.note_work("ed_split_f32_nhwc", nbytes=x.numel() * (4.0 + 6.0 * u * u))
    _LAUNCH["device"] = x.device
    _call("ed_split_f32_nhwc", x.data_ptr(), out.data_ptr(), N, C, H, W, int(bool(upsample2x)),
          None if absmax is None else absmax.data_ptr(), _stream())
    return out


def split_conv_weight(w):
    """fp32 Conv2d weight [N,Cin,3,3] -> (fp16 channels_last [N,3Cin,3,3] = [wh | wh | wl] of 2^k w, 2^-k): the B operand of
    ``conv3x3_f32out`` for an A operand [xh | xl | xh].  2^k puts max |w| in [2^13, 2^14), so wl = fp16(2^k w - wh) stays out of
    fp16's subnormal range for every weight above 2^-24 of the largest (exact: a power of two)."""
    import math
    wmax = float(w.detach().abs().max())
    k = 0 if not (wmax > 0.0 and math.isfinite(wmax)) else max(-24, min(24, 13 - math.frexp(wmax)[1] + 1))
    ws = w.detach().float() * (2.0 ** k)
    hi = ws.half()
    lo = (ws - hi.float()).half()
    return torch.cat([hi, hi, lo], dim=1).contiguous(memory_format=torch.channels_last), 2.0 ** -k


def conv3x3_f32out_ok(B, H, W, Cin3, N):
    M = B * H * W
    return Cin3 % 64 == 0 and N % 8 == 0 and M * Cin3 * 2 < 2 ** 31 - 16 and N * 9 * Cin3 * 2 < 2 ** 31 - 16 and M < 2 ** 31


def conv3x3_f32out(a, w, bias=None, residual=None, out_scale=1.0, act_absmax=None):
    """a [B,Cin',H,W] fp16 and w [N,Cin',3,3] fp16 in channels_last memory (split operands: Cin' = 3 Cin, ``groupnorm_nhwc_f32(split=True)``
    / ``split_conv_weight``) -> out_scale * conv2d(a, w, stride 1, padding 1) + bias + residual as fp32 channels_last [B,N,H,W]; bias fp32
    [N], residual fp32 channels_last.  ``act_absmax``: the tensor ``split_f32`` scaled ``a`` by (the result is multiplied by 2^e).
    See ed_conv3x3_nhwc_f32out."""
    if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dim() == 4 and a.dtype == torch.float16):
        _reject("conv3x3_f32out: a must be an fp16 [B,C,H,W] tensor on the MI355X; no CPU fallback")
    B, C3, H, W = a.shape
    N = w.shape[0]
    cl = torch.channels_last
    if tuple(w.shape) != (N, C3, 3, 3) or w.dtype != torch.float16 or not conv3x3_f32out_ok(B, H, W, C3, N):
        _reject(f"conv3x3_f32out: unsupported shape a {tuple(a.shape)} w {tuple(w.shape)}")
    if not a.is_contiguous(memory_format=cl) or not w.is_contiguous(memory_format=cl):
        _reject("conv3x3_f32out: a and w must be channels_last")
    out = torch.empty((B, N, H, W), dtype=torch.float32, device=a.device, memory_format=cl)
    if residual is not None and (residual.shape != out.shape or residual.dtype != torch.float32 or not residual.is_contiguous(memory_format=cl)):
        _reject("conv3x3_f32out: residual must be an fp32 channels_last tensor of the output's shape")
    TIMER.note_work("ed_conv3x3_nhwc_f32out", flops=2.0 * B * H * W * 9 * C3 * N,
                    nbytes=2.0 * (B * H * W * C3 + 9 * C3 * N) + 4.0 * B * H * W * N * (2 if residual is not None else 1))
    _call("ed_conv3x3_nhwc_f32out", a.data_ptr(), w.data_ptr(), _opt(bias, torch.float32, "bias"),
          None if residual is None else residual.data_ptr(), out.data_ptr(), _DTYPE[torch.float16], B, H, W, C3, N, float(out_scale),
          None if act_absmax is None else act_absmax.data_ptr(), _stream_of(a))
    return out


def conv3x3_f32out_s2_ok(B, Hi, Wi, Cin3, N):
    """can ed_conv3x3_nhwc_f32out_s2 take a split operand [B, Cin3, Hi, Wi] (the INPUT size)?"""
    return (Hi % 2 == 0 and Wi % 2 == 0 and Cin3 % 64 == 0 and N % 8 == 0 and B * Hi * Wi * Cin3 * 2 < 2 ** 31 - 16
            and N * 9 * Cin3 * 2 < 2 ** 31 - 16)


def conv3x3_f32out_s2(a, w, bias=None, out_scale=1.0, act_absmax=None):
    """a [B,Cin',2H,2W] fp16 split operand (``split_f32`` of the raw fp32 stream) and w [N,Cin',3,3] fp16 (``split_conv_weight``), channels_last ->
    out_scale * conv2d(F.pad(a, (0, 1, 0, 1)), w, stride 2) + bias as fp32 channels_last [B,N,H,W]: the VAE encoder's Downsample2D at fp32
    accuracy on the MFMA pipe.  See ed_conv3x3_nhwc_f32out_s2."""
    if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dim() == 4 and a.dtype == torch.float16):
        _reject("conv3x3_f32out_s2: a must be an fp16 [B,C,H,W] tensor on the MI355X; no CPU fallback")
    B, C3, Hi, Wi = a.shape
    N = w.shape[0]
    cl = torch.channels_last
    if tuple(w.shape) != (N, C3, 3, 3) or w.dtype != torch.float16 or not conv3x3_f32out_s2_ok(B, Hi, Wi, C3, N):
        _reject(f"conv3x3_f32out_s2: unsupported shape a {tuple(a.shape)} w {tuple(w.shape)}")
    if not a.is_contiguous(memory_format=cl) or not w.is_contiguous(memory_format=cl):
        _reject("conv3x3_f32out_s2: a and w must be channels_last")
    H, W = Hi // 2, Wi // 2
    out = torch.empty((B, N, H, W), dtype=torch.float32, device=a.device, memory_format=cl)
    TIMER.note_work("ed_conv3x3_nhwc_f32out_s2", flops=2.0 * B * H * W * 9 * C3 * N, nbytes=2.0 * (B * Hi * Wi * C3 + 9 * C3 * N) + 4.0 * B * H * W * N)
    _call("ed_conv3x3_nhwc_f32out_s2", a.data_ptr(), w.data_ptr(), _opt(bias, torch.float32, "bias"), out.data_ptr(), _DTYPE[torch.float16],
          B, H, W, C3, N, float(out_scale), None if act_absmax is None else act_absmax.data_ptr(), _stream_of(a))
    return out


def softmax_rows_(x, scale=1.0):
    """x fp32 [..., cols] contiguous -> softmax(scale * x) over the last dim, IN PLACE (ed_softmax_rows)."""
    cols = x.shape[-1]
    if cols % 4 and x.numel() != cols:
        _reject("softmax_rows_: cols must be a multiple of 4")
    TIMER.note_work("ed_softmax_rows", nbytes=3.0 * x.numel() * 4)
    _call("ed_softmax_rows", _dev(x, torch.float32, "x"), x.numel() // cols, cols, float(scale), _stream())
    return x


VAE_ATTENTION_CHUNK_BYTES = 1 << 30  # score-matrix bytes materialised at a time by vae_attention (a few thousand rows keep the GEMMs efficient)


def vae_attention(q, k, v):
    """softmax(q k^T / sqrt(C)) v for the VAE's single-head attention, fp32 [B, N, C]: two library fp32 GEMMs around
    ed_softmax_rows, over chunks of query rows so that the [rows, N] score block stays under VAE_ATTENTION_CHUNK_BYTES
    (N = 32768 tokens for the 1024x2048 decode: 4.3 GB of scores per sample).  Replaces SDPA / AOTriton."""
    B, N, C = q.shape
    scale = C ** -0.5
    # rows per chunk: the [B, rows, N] fp32 score block stays under the cap for any B (at least 64 rows)
    rows = max(64, min(N, VAE_ATTENTION_CHUNK_BYTES // (4 * N * B) // 64 * 64)) if N > 256 else N
    kt = k.transpose(1, 2)
    if rows >= N:
        return torch.bmm(softmax_rows_(torch.bmm(q, kt), scale), v)
    out = torch.empty_like(q)
    for a in range(0, N, rows):
        s = torch.bmm(q[:, a:a + rows], kt)                      # [B, rows, N]
        out[:, a:a + rows] = torch.bmm(softmax_rows_(s, scale), v)
    return out


# kernel variant (identical results): bit 0 = V staging (0: ds_read_b64_tr_b16 from a row-major V tile, 1: V^T tile in
# LDS); bit 1 = 64 query rows per wave (256 per workgroup) instead of 32.  None = choose per shape: the 64-row variant is
# faster only when there is enough work to fill the chip with half as many workgroups (measured, batch 20: N=4096
# 848 vs 776 TFLOP/s; N=1024 and the 77-key cross attention: no gain or slower; profiles/r2_s7_probe_attn.jsonl)
# Round 3: 4 = software-pipelined self-attention kernel, 5 = the same with the lazy row maximum (no per-tile max after the
# first tile; exact redo when a lane's sum of numerators exceeds 2^6), 8 = small-KV kernel (Nk <= 96: the 77-token cross
# attention).  Round 4: 6 = 5 on exponent-domain queries (the caller folds scale * log2 e into q -- models.Attention folds it
# into the query projection weights -- and the reference maximum rides in the MFMA accumulator's initial value, so a
# numerator is one v_exp_f32: 157 instead of 189 (lazy) / 214 (exact) instructions per 64-key tile and wave).
FLASH_V_PATH = None
FLASH_HEAD_DIMS = (40, 64, 80, 160)   # 64: SDXL / SD 2.x (all the variants above); 40 / 80 / 160: SD 1.x's 8 heads (k_flash_attn_gen)
# Measured inside the SDXL UNet forward on the MI355X (profiles/r4_s2_unet_forward_ab.txt, batch 20 / 6, all other kernels
# equal): v_path 4 155.2 / 56.5 ms, v_path 5 154.0 / 56.0 ms, v_path 6 155.3 / 56.4 ms.  Removing 12 % (5) and a further 17 % (6)
# of the loop's instructions moves the forward by < 1 %: the kernel is NOT instruction-issue-bound as round 3 concluded from
# SQ_ACTIVE_INST_ANY -- every wave re-reads the whole K and V tile from LDS for its 32 query rows (16 KiB per 16 MFMAs), which
# no variant changes.  5 is the default; 6 stays selectable (FLASH_EXP2 / ED_FLASH_VARIANT=6) and tested.
FLASH_EXP2 = False   # self-attention through v_path 6 where the pipelined kernel applies (see flash_prescale)
_ENV_VARIANT = __import__("os").environ.get("ED_FLASH_VARIANT")  # A/B: "legacy" = the round-2 choice, or a variant number
LOG2E = 1.4426950408889634


def _pipe_fits(Nk, row_stride):
    """The pipelined kernels address K / V with 32-bit byte offsets and want at least two full tiles."""
    return Nk >= 128 and (Nk + 128) * row_stride * 2 < 2 ** 31


def flash_prescale(Nq, Nk, row_stride, scale=0.125):
    """-> the factor a caller must fold into q (softmax scale * log2 e) to run the exponent-domain kernel (v_path 6) for
    a self-attention of this shape, or None when that kernel does not apply (then q stays as it is).  ``row_stride``:
    elements between consecutive tokens of k / v (3 * inner for slices of a fused QKV projection)."""
    if FLASH_V_PATH is not None:
        want = int(FLASH_V_PATH)
    elif _ENV_VARIANT is not None:
        want = -1 if _ENV_VARIANT == "legacy" else int(_ENV_VARIANT)
    else:
        want = 6 if FLASH_EXP2 else -1
    return scale * LOG2E if (want == 6 and _pipe_fits(Nk, row_stride)) else None


def _flash_variant(B, heads, Nq, Nk, k=None, v=None, prescaled=False):
    fits = k is None or _pipe_fits(Nk, max(k.stride(1), v.stride(1)))
    if prescaled:
        if not (Nk >= 128 and fits):
            _reject("flash_attention: exponent-domain q needs the pipelined kernel (Nk >= 128, 32-bit K / V offsets)")
        return 6
    if FLASH_V_PATH is not None and int(FLASH_V_PATH) != 6:
        return int(FLASH_V_PATH)
    legacy = 2 if (Nq >= 2048 and Nk >= 1024 and B * heads * (Nq // 256) >= 1024) else 0
    if _ENV_VARIANT == "legacy":
        return legacy
    if _ENV_VARIANT is not None and int(_ENV_VARIANT) != 6:
        want = int(_ENV_VARIANT)
        if want in (4, 5, 7, 9, 10):
            return want if (Nk >= 128 and fits) else (8 if Nk <= 96 else legacy)
        return want if (want != 8 or Nk <= 96) else legacy
    if Nk <= 96:
        return 8
    if Nk >= 128 and fits:
        return FLASH_DEFAULT_PIPE
    return legacy


FLASH_DEFAULT_PIPE = 5   # pipelined variant for natural-domain q (4 exact / 5 lazy maximum: +0.4-0.8 % on the forward, profiles/r4_s1_attention_variant_in_unet.txt)


def flash_attention(q, k, v, heads, v_path=None, prescaled=False):
    """q [B,Nq,H*d], k / v [B,Nk,H*d] 16-bit, d = head dim in FLASH_HEAD_DIMS (last dim contiguous; batch / token strides free,
    so column slices of a fused QKV projection are fine) -> softmax(q k^T / sqrt(d)) v as a contiguous [B,Nq,H*d] tensor.  ``prescaled``: q already
    carries the factor ``flash_prescale`` returned (exponent-domain kernel, v_path 6).  See ed_flash_attention."""
    B, Nq, HD = q.shape
    Nk = k.shape[1]
    hd = HD // heads
    if HD != heads * hd or hd not in FLASH_HEAD_DIMS or k.shape != (B, Nk, HD) or v.shape != (B, Nk, HD):
        _reject(f"flash_attention: head_dim must be one of {FLASH_HEAD_DIMS} and shapes consistent (q {tuple(q.shape)}, "
                f"k {tuple(k.shape)}, v {tuple(v.shape)}, heads {heads})")
    if hd != 64 and (prescaled or v_path is not None):
        _reject("flash_attention: kernel variants exist for head_dim 64 only")
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        if not t.is_cuda or t.dtype != q.dtype or t.stride(2) != 1:
            _reject(f"flash_attention: {name} must be a 16-bit tensor on the MI355X with unit last stride; no CPU fallback")
        if _LAUNCH["device"] is None:
            _LAUNCH["device"] = t.device
        elif _LAUNCH["device"] != t.device:
            _reject(f"flash_attention: {name} lives on {t.device}, q on {_LAUNCH['device']}")
    if v_path is not None and (int(v_path) == 6) != bool(prescaled):
        _reject("flash_attention: v_path 6 takes exponent-domain q (prescaled=True) and nothing else does")
    out = torch.empty(B, Nq, HD, dtype=q.dtype, device=q.device)
    # algorithmic work: QK^T and PV contractions (4 B H Nq Nk 64 flop); q, k, v read once, out written once
    TIMER.note_work("ed_flash_attention", flops=4.0 * B * heads * Nq * Nk * hd,
                    nbytes=2.0 * q.element_size() * HD * B * (Nq + Nk))
    variant = 0 if hd != 64 else (_flash_variant(B, heads, Nq, Nk, k, v, prescaled) if v_path is None else int(v_path))
    _call("ed_flash_attention", q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _code(q, "q"), B, heads, Nq, Nk,
          hd, q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), out.stride(0), out.stride(1),
          hd ** -0.5, variant, _stream())
    return out
