"""Lane-level replay of elasticdiffusion_official_amd/csrc/gemm_kernels.hip on the CPU (numpy): index algebra AND schedule hazards.

The kernel was written with no GPU at hand, so everything that can be checked without one is checked here:

* every expression that turns (workgroup, wave, lane, tile, phase) into a global byte offset, an LDS byte address, an MFMA
  operand slot or an output address is restated below, one to one with the .hip file (same names), and the whole kernel is
  replayed lane by lane: LDS-DMA (64 lanes x 16 B to wave-uniform base + 16 lane), swizzled fragment reads,
  v_mfma_f32_16x16x32 (A[i][k]: lane i + 16 (k >> 3), element k & 7; B[k][j]: lane j + 16 (k >> 3); D[i][j]: lane
  j + 16 (i >> 2), register i & 3), the epilogue's packed stores.  Inputs are small integers, so x @ W^T is exact in
  fp32 and the GEMM part is compared bit for bit; the GELU polynomial is compared with erf in float64;
* the schedule is replayed with the two wave rows half a phase apart (the second one passes one extra barrier first) under
  the two adversarial timings the hardware allows:
    "dma_early_read_late": an LDS-DMA lands the moment it is issued, a fragment read samples LDS only at the s_waitcnt
                           that retires it, and after every DMA issued in that barrier interval (write-after-read races);
    "dma_late_read_early": an LDS-DMA lands only at the issuing wave's counted vmcnt that retires it (end of that barrier
                           interval), a fragment read samples LDS the moment it is issued and before anything lands in
                           its interval (read-after-write races).
  A schedule that gives the exact result under both is free of LDS races between barrier intervals, whatever the DMA
  latency.  (`--break war|raw|lgkm` moves one staging step / weakens one wait / drops the early lgkmcnt and shows the replay catching it.)

Run:  python tools/emulate_gemm_kernel.py            (about a minute)
"""
import argparse
import math
import sys

import numpy as np

BM, BN, BK, SUB, W_REGION, BUF = 256, 128, 64, 1024, 32768, 65536


def swz(p):
    return p ^ (((p >> 9) & 1) << 5)


def x_sub(rg, kh):
    return (rg * 2 + kh) * SUB


def w_sub(rg, kh):
    return W_REGION + (rg * 2 + kh) * SUB


LANES = np.arange(64)


# tools/gemm_sched/gemm_sched.hip's descriptors: (slot of the 8 pieces, pos, two-phase loop)
SCHEDS = {
    "product": ((0, 0, 2, 2, 4, 4, 6, 6), None, False),
    "late": ((1, 1, 3, 3, 5, 5, 6, 6), None, False),
    "mid": ((1, 1, 3, 3, 5, 5, 6, 6), None, False),
    "spread": ((0, 1, 2, 3, 4, 5, 6, 6), None, False),
    "mfma_all": ((1, 1, 3, 3, 5, 5, 7, 7), None, False),
    "r4": ((0, 0, 2, 4, 6, 6, 6, 6), None, False),
    "two_read": ((0, 0, 2, 2, 2, 2, 2, 2), None, True),
    "two_mfma": ((0, 0, 3, 3, 3, 3, 3, 3), None, True),
    "bad_early_b": ((0, 0, 0, 0, 4, 4, 6, 6), None, False),     # BROKEN ON PURPOSE: value rows re-staged in the phase that reads them
}


class Wave:
    """One wave's registers, DMA queue and pending reads."""

    def __init__(self, blk, wave):
        self.blk, self.wave = blk, wave
        self.wrow, self.wcol = wave >> 2, wave & 3
        lane = LANES
        ps = swz(16 * lane)
        srow, skb = ps >> 6, ps & 63
        rb = blk.K * 2
        xrb = rb // 9 if blk.conv else rb                   # CONV: one pixel's Cin values
        xrow0 = blk.m0 + ((wave & 3) + (4 if getattr(blk, "rows_mode", False) else 8) * (wave >> 2)) * 16 + srow
        x0 = xrow0 * xrb + skb
        self.x_voff = [x0, x0 + 64 * xrb]
        self.px_mask = [np.zeros(64, np.int64), np.zeros(64, np.int64)]
        if blk.conv:
            img_h, img_w = blk.conv
            for h in range(2):
                m = xrow0 + 64 * h
                rem = m % (img_h * img_w)
                py, px = rem // img_w, rem % img_w
                for t in range(9):
                    yy, xx = py + t // 3 - 1, px + t % 3 - 1
                    ok = (m < blk.M) & (yy >= 0) & (yy < img_h) & (xx >= 0) & (xx < img_w)
                    self.px_mask[h] |= ok.astype(np.int64) << t
        w0 = (blk.n0 + 32 * (wave >> 1) + 8 * (srow >> 2) + (srow & 3) + 4 * (wave & 1)) * rb + skb
        self.w_voff = [w0, w0 + blk.gap * rb]
        rd = swz((lane & 15) * 64 + (lane >> 4) * 16)
        self.xrd = rd + self.wrow * 8 * (2 * SUB)
        self.wrd = rd + W_REGION + self.wcol * 2 * (2 * SUB)
        self.acc = np.zeros((8, 4, 64, 4), np.float64)
        self.done = []        # persistent variant: (m0, n0, accumulators) of the tiles this wave has finished
        self.frag = {}        # name -> [64, 8] values
        self.vmq = []         # outstanding DMAs, oldest first: (lds_byte_base, values[64, 8])
        self.pending = []     # outstanding fragment reads, oldest first: (name, lds byte addresses[64])


class Block:
    def __init__(self, x, w, bias, M, K, I, m0, n0, mode, breakage=None, epi=0, conv=None, row_bias=None, residual=None,
                 rows_per_sample=1, rows_mode=False):
        self.rows_mode = rows_mode                      # round 6: 128-row tile, each wave row runs its m-half 0 only (tile_phases_rows)
        self.x, self.w, self.bias, self.M, self.K, self.I, self.m0, self.n0 = x, w, bias, M, K, I, m0, n0
        # round 4 epilogue addends of the plain projection / convolution: row_bias [M / rows_per_sample, I] (the time embedding),
        # residual [M, I]; the kernel reads 8 consecutive columns (16 bytes) per (lane, 16-row block, half) of each
        self.row_bias, self.residual, self.rows_per_sample = row_bias, residual, rows_per_sample
        self.epi, self.gap = epi, (I if epi == 0 else BN)
        self.conv = conv                                # None, or (img_h, img_w): x is NHWC [B, img_h, img_w, K / 9]
        self.cpt = K // (9 * BK) if conv else 1
        self.mode, self.breakage = mode, breakage
        self.lds = np.full(2 * BUF // 2, np.nan)      # 16-bit elements; NaN = never written
        self.waves = [Wave(self, i) for i in range(8)]
        self.landing = []                               # DMAs retired in this interval (late mode): applied at its end

    # ---- memory side ----------------------------------------------------------------------------------------------
    def gload(self, which, voff, soff):
        """16 bytes per lane from x (which = 0) or W (1) through a raw buffer: out-of-range offsets read as zeros.  The kernel keeps the
        scalar and immediate offsets at 0 (soff is always 0 here); the model still range-checks the VGPR offset alone and asserts
        that an in-range one gives an in-range address."""
        src = self.x if which == 0 else self.w
        nbytes = src.size * 2
        out = np.zeros((64, 8))
        for l in range(64):
            if voff[l] + 16 <= nbytes:
                a = voff[l] + soff
                assert a + 16 <= nbytes, "in-range VGPR offset + scalar offset leaves the buffer"
                out[l] = src.reshape(-1)[a // 2:a // 2 + 8]
            else:
                assert voff[l] >= nbytes
        return out

    def k_pos(self, tile):
        """KPos of the .hip file, derived from the position in the walk (the kernel advances it incrementally: k_next).  Round 6: the
        convolution walks channel-block-major -- the 9 taps of channel block 0, then of block 1, ... (L2 reuse across the taps)."""
        pos = (0, 0, 0)
        for _ in range(tile):                       # k_next of the .hip file
            t, tap, ct = pos
            t, tap = t + 1, tap + 1 if self.conv else tap
            if self.conv and tap == 9:
                tap, ct = 0, ct + 1
            pos = (t, tap, ct)
        assert pos == ((tile, tile % 9, tile // 9) if self.conv else (tile, 0, 0))
        return pos

    def w_index(self, tile):
        """KPos::w -- the K-tile index into W's rows ([N, 3, 3, Cin]: k = tap Cin + channel) of walk position `tile`"""
        _, tap, ct = self.k_pos(tile)
        return tap * self.cpt + ct if self.conv else tile

    def land(self, base, vals):
        e0 = base // 2
        self.lds[e0:e0 + 512] = vals.reshape(-1)        # wave-uniform base + 16 bytes x lane

    def dma(self, wv, which, lds_base, voff, soff):
        if self.pass_ == "rest":
            return
        vals = self.gload(which, voff, soff)
        if self.mode == "dma_early_read_late":
            self.land(lds_base, vals)
        wv.vmq.append((lds_base, vals))

    def wait_vm(self, wv, n):
        if self.pass_ == "issue_dma":
            return
        while len(wv.vmq) > n:
            base, vals = wv.vmq.pop(0)
            if self.mode == "dma_late_read_early":
                self.landing.append((base, vals))      # lands no later than this wait; visible to others after the barrier

    def sample(self, addr):
        a = addr // 2
        return np.stack([self.lds[a[l]:a[l] + 8] for l in range(64)])

    def read(self, wv, name, addr):
        if self.pass_ == "issue_dma":
            return
        if self.mode == "dma_late_read_early":
            wv.frag[name] = self.sample(addr)
        else:
            wv.pending.append((name, addr))

    def wait_lgkm(self, wv, n):
        if self.pass_ == "issue_dma":
            return
        while len(wv.pending) > n:
            name, addr = wv.pending.pop(0)
            wv.frag[name] = self.sample(addr)

    # ---- the kernel's helpers, same names ---------------------------------------------------------------------------
    def stage_x(self, wv, bufi, tile, h):
        rg = (wv.wave & 3) + 8 * (wv.wave >> 2) + 4 * h
        dst = bufi * BUF + x_sub(0, 0) + rg * (2 * SUB)
        if self.conv:
            _, tap, ct = self.k_pos(tile)
            dy, dx = tap // 3 - 1, tap - 3 * (tap // 3) - 1
            delta = (dy * self.conv[1] + dx) * (self.K // 9 * 2) + ct * (BK * 2)
            vo = np.where((wv.px_mask[h] >> tap) & 1, wv.x_voff[h] + delta, 0x80000000)
            self.dma(wv, 0, dst, vo, 0)
            self.dma(wv, 0, dst + SUB, vo + 64, 0)
            return
        vo = wv.x_voff[h] + tile * (BK * 2)
        self.dma(wv, 0, dst, vo, 0)
        self.dma(wv, 0, dst + SUB, vo + 64, 0)

    def stage_w(self, wv, bufi, tile, g):
        rg = 8 * g + wv.wave
        vo = wv.w_voff[g] + self.w_index(tile) * (BK * 2)
        dst = bufi * BUF + w_sub(0, 0) + rg * (2 * SUB)
        self.dma(wv, 1, dst, vo, 0)
        self.dma(wv, 1, dst + SUB, vo + 64, 0)

    def read_x(self, wv, bufi, mh):
        for mf in range(4):
            for kh in range(2):
                self.read(wv, ("x", mf, kh), wv.xrd + bufi * BUF + x_sub(mh * 4 + mf, kh))

    def read_w(self, wv, bufi, g):
        for nf in range(2):
            for kh in range(2):
                self.read(wv, ("wg" if g else "wv", nf, kh), wv.wrd + bufi * BUF + w_sub(8 * g + nf, kh) - W_REGION)

    def mma16(self, wv, mh, g):
        if self.pass_ == "issue_dma":
            return
        assert not wv.pending, "MFMA issued with fragment reads outstanding"
        for kh in range(2):
            for mf in range(4):
                for nf in range(2):
                    a = wv.frag[("wg" if g else "wv", nf, kh)]      # A operand: [lane, 8]
                    b = wv.frag[("x", mf, kh)]                      # B operand
                    A = np.zeros((16, 32))
                    B = np.zeros((32, 16))
                    for kg in range(4):
                        A[:, 8 * kg:8 * kg + 8] = a[16 * kg:16 * kg + 16]
                        B[8 * kg:8 * kg + 8, :] = b[16 * kg:16 * kg + 16].T
                    D = A @ B
                    accv = wv.acc[mh * 4 + mf][g * 2 + nf]
                    for i in range(16):
                        accv[16 * (i >> 2):16 * (i >> 2) + 16, i & 3] += D[i, :]

    # ---- program: a list of barrier-separated segments per wave ------------------------------------------------------
    def tile_segments(self, wv, bufi, tile, s1, s2, first=False):
        """``first`` (round 4, "early start"): tile 0 of a workgroup when the prologue waited only for the operands of phase 1
        (W value rows + x m-half 0: 4 of its 14 DMAs).  The gate rows (phase 2) and x m-half 1 (phase 3) are then retired by a
        ``vmcnt(10)`` in front of BOTH barriers of phases 1 and 2: with the two wave rows half a phase apart, the barrier that
        publishes a wave's DMAs to the row that reads first is that row's second barrier and the other row's first one."""
        brk = self.breakage
        segs = []
        early = (lambda: self.wait_vm(wv, 12 if brk == "early" else 10)) if first else (lambda: None)

        def ph1a():
            self.read_w(wv, bufi, 0)
            self.read_x(wv, bufi, 0)
            if s1:
                self.stage_x(wv, bufi ^ 1, tile + 1, 1)
            if brk == "war" and s2:
                self.stage_x(wv, bufi, tile + 2, 0)    # BROKEN ON PURPOSE: x m-half 0 re-staged in the phase that reads it
            if brk != "lgkm":                           # BROKEN ON PURPOSE without it: value rows re-staged one phase later
                self.wait_lgkm(wv, 8)
            early()
        segs.append(ph1a)
        segs.append(lambda: (self.wait_lgkm(wv, 0), self.mma16(wv, 0, 0), early()))

        def ph2a():
            self.read_w(wv, bufi, 1)
            if s2:
                self.stage_w(wv, bufi, tile + 2, 0)
            early()
        segs.append(ph2a)
        segs.append(lambda: (self.wait_lgkm(wv, 0), self.mma16(wv, 0, 1), early()))

        def ph3a():
            self.read_x(wv, bufi, 1)
            if s2 and brk != "war":
                self.stage_x(wv, bufi, tile + 2, 0)
        segs.append(ph3a)
        segs.append(lambda: (self.wait_lgkm(wv, 0), self.mma16(wv, 1, 1)))

        def ph4a():
            if s2:
                self.stage_w(wv, bufi, tile + 2, 1)
                self.wait_vm(wv, 8 if brk == "raw" else 6)   # BROKEN ON PURPOSE with 8: x m-half 1 of tile + 1 may not have landed
            else:
                self.wait_vm(wv, 0)
        segs.append(ph4a)
        segs.append(lambda: self.mma16(wv, 1, 0))
        return segs

    def tile_segments_b1(self, wv, tile, s1, s2, nxt):
        """tools/gemm_persist v2 (`tile_phases_b1`): a K tile in buffer 1; ``nxt`` = (m0, n0) of the workgroup's next output tile when
        this is the LAST K tile and it prefetches -- each phase then stages one half of the next tile's K tile 0 into buffer 0 (W value
        rows, x m-half 0, W gate rows, x m-half 1) and phase 4 waits for nothing."""
        if nxt is None:
            return self.tile_segments(wv, 1, tile, s1, s2)
        assert not s1 and not s2
        brk = self.breakage

        def nxt_wave():
            keep = self.m0, self.n0
            self.m0, self.n0 = nxt
            fresh = Wave(self, wv.wave)
            self.m0, self.n0 = keep
            return fresh

        def stage(kind, half, bufi=0):
            fresh, mine = nxt_wave(), (wv.x_voff, wv.w_voff)
            wv.x_voff, wv.w_voff = fresh.x_voff, fresh.w_voff
            (self.stage_x if kind == "x" else self.stage_w)(wv, bufi, 0, half)
            wv.x_voff, wv.w_voff = mine

        segs = []
        # BROKEN ON PURPOSE ("pf"): the first prefetch goes to buffer 1, whose value rows this very phase reads
        segs.append(lambda: (self.read_w(wv, 1, 0), self.read_x(wv, 1, 0), stage("w", 0, 1 if brk == "pf" else 0), self.wait_lgkm(wv, 8)))
        segs.append(lambda: (self.wait_lgkm(wv, 0), self.mma16(wv, 0, 0)))
        segs.append(lambda: (self.read_w(wv, 1, 1), stage("x", 0)))
        segs.append(lambda: (self.wait_lgkm(wv, 0), self.mma16(wv, 0, 1)))
        segs.append(lambda: (self.read_x(wv, 1, 1), stage("w", 1)))
        segs.append(lambda: (self.wait_lgkm(wv, 0), self.mma16(wv, 1, 1)))
        segs.append(lambda: stage("x", 1))
        segs.append(lambda: self.mma16(wv, 1, 0))
        return segs

    # ---- tools/gemm_sched: the same loop from a schedule descriptor (slot / pos of the 8 pieces, see gemm_sched.hip) -------------
    def x_vo(self, wv, tile, h):
        """gemm_sched.hip's x_vo: the lane addresses of m half h at K tile `tile` (convolution: the tap's pixel or the out-of-range sentinel)"""
        if self.conv:
            _, tap, ct = self.k_pos(tile)
            dy, dx = tap // 3 - 1, tap - 3 * (tap // 3) - 1
            delta = (dy * self.conv[1] + dx) * (self.K // 9 * 2) + ct * (BK * 2)
            return np.where((wv.px_mask[h] >> tap) & 1, wv.x_voff[h] + delta, 0x80000000)
        return wv.x_voff[h] + tile * (BK * 2)

    def piece(self, wv, bufi, tile, i, s1, s2):
        op, k = i >> 1, i & 1
        if op == 0:
            if s1:
                rg = (wv.wave & 3) + 8 * (wv.wave >> 2) + 4
                self.dma(wv, 0, (bufi ^ 1) * BUF + x_sub(0, 0) + rg * (2 * SUB) + k * SUB, self.x_vo(wv, tile + 1, 1) + 64 * k, 0)
        elif op == 2:
            if s2:
                rg = (wv.wave & 3) + 8 * (wv.wave >> 2)
                self.dma(wv, 0, bufi * BUF + x_sub(0, 0) + rg * (2 * SUB) + k * SUB, self.x_vo(wv, tile + 2, 0) + 64 * k, 0)
        elif s2:
            g = 0 if op == 1 else 1
            rg = 8 * g + wv.wave
            self.dma(wv, 1, bufi * BUF + w_sub(0, 0) + rg * (2 * SUB) + k * SUB, wv.w_voff[g] + self.w_index(tile + 2) * (BK * 2) + 64 * k, 0)

    def tile_segments_sched(self, wv, bufi, tile, s1, s2, sched, first=False):
        slot, two = SCHEDS[sched][0], SCHEDS[sched][2]
        upto = lambda last, first_id=0: sum(slot[i] <= last for i in range(first_id, 8))    # noqa: E731
        issue = lambda sl: [self.piece(wv, bufi, tile, i, s1, s2) for i in range(8) if slot[i] == sl]   # noqa: E731
        early = (lambda n: self.wait_vm(wv, n)) if first else (lambda n: None)
        tile_wait = lambda last: self.wait_vm(wv, upto(last, 2) if s2 else 0)   # noqa: E731
        if two:
            assert slot[0] <= 2 and slot[1] <= 2 and not first
            return [
                lambda: (self.read_w(wv, bufi, 0), self.read_x(wv, bufi, 0), self.read_w(wv, bufi, 1), issue(0), self.wait_lgkm(wv, 0)),
                lambda: (issue(1), self.mma16(wv, 0, 0), self.mma16(wv, 0, 1)),
                lambda: (self.read_x(wv, bufi, 1), issue(2), tile_wait(2), self.wait_lgkm(wv, 0)),
                lambda: (issue(3), self.mma16(wv, 1, 1), self.mma16(wv, 1, 0)),
            ]
        assert slot[0] <= 6 and slot[1] <= 6
        return [
            lambda: (self.read_w(wv, bufi, 0), self.read_x(wv, bufi, 0), issue(0), self.wait_lgkm(wv, 8), early(14 + upto(0) - 6)),
            lambda: (self.wait_lgkm(wv, 0), issue(1), self.mma16(wv, 0, 0), early(14 + upto(1) - 6)),
            lambda: (self.read_w(wv, bufi, 1), issue(2), early(14 + upto(2) - 8)),
            lambda: (self.wait_lgkm(wv, 0), issue(3), self.mma16(wv, 0, 1), early(14 + upto(3) - 8)),
            lambda: (self.read_x(wv, bufi, 1), issue(4)),
            lambda: (self.wait_lgkm(wv, 0), issue(5), self.mma16(wv, 1, 1)),
            lambda: (issue(6), tile_wait(6)),
            lambda: (issue(7), self.mma16(wv, 1, 0)),
        ]

    def tile_segments_half(self, wv, bufi, tile, s1, s2):
        """gemm_sched.hip's tile_phases_half: the value half only (a column tile whose gate half lies beyond N), 4 intervals per K tile"""
        P = lambda i: self.piece(wv, bufi, tile, i, s1, s2)    # noqa: E731
        return [
            lambda: (self.read_w(wv, bufi, 0), self.read_x(wv, bufi, 0), P(0), P(1), self.wait_lgkm(wv, 0)),
            lambda: self.mma16(wv, 0, 0),
            lambda: (self.read_x(wv, bufi, 1), P(2), P(3), P(4), P(5),
                     # BROKEN ON PURPOSE with 6 ("half_raw"): x m-half 1 of tile + 1 may not have landed when the tile ends
                     self.wait_vm(wv, (6 if self.breakage == "half_raw" else 4) if s2 else 0), self.wait_lgkm(wv, 0)),
            lambda: self.mma16(wv, 1, 0),
        ]

    def tile_segments_rows(self, wv, bufi, tile, s1, s2):
        """gemm_kernels.hip's tile_phases_rows (round 6): a 128-row tile, m-half 0 only, both column halves -- tile_phases_half with the roles
        of "x m-half 1" and "W gate rows" exchanged"""
        return [
            lambda: (self.read_w(wv, bufi, 0), self.read_x(wv, bufi, 0), self.stage_w(wv, bufi ^ 1, tile + 1, 1) if s1 else None,
                     self.wait_lgkm(wv, 0)),
            lambda: self.mma16(wv, 0, 0),
            lambda: (self.read_w(wv, bufi, 1),
                     (self.stage_w(wv, bufi, tile + 2, 0), self.stage_x(wv, bufi, tile + 2, 0)) if s2 else None,
                     # BROKEN ON PURPOSE with 6 ("rows_raw"): the W gate rows of tile + 1 may not have landed when the tile ends
                     self.wait_vm(wv, (6 if self.breakage == "rows_raw" else 4) if s2 else 0), self.wait_lgkm(wv, 0)),
            lambda: self.mma16(wv, 0, 1),
        ]

    def retarget(self, wv, m0, n0):
        """tools/gemm_persist: the workgroup moves on to its next tile -- the addresses `setup()` recomputes, a fresh accumulator."""
        self.m0, self.n0 = m0, n0
        fresh = Wave(self, wv.wave)
        wv.x_voff, wv.px_mask, wv.w_voff = fresh.x_voff, fresh.px_mask, fresh.w_voff
        wv.acc = np.zeros((8, 4, 64, 4), np.float64)

    def program_persist(self, wv, tiles, v2=False):
        """The persistent experiment (tools/gemm_persist/gemm_persist.hip): this workgroup's tiles back to back; the NEXT tile's
        prologue DMAs are issued in the interval that follows the barrier pair ending the current tile (before its epilogue, which
        touches no LDS), and waited for at the top of the next tile exactly like a fresh workgroup's."""
        nt = self.K // BK
        two = bool(self.sched) and SCHEDS[self.sched][2]     # the 4-interval loop inside the persistent one (r5_patches/0006: no early start)
        early_start = nt >= 3 and not two
        prefetch = v2 and nt >= 6 and nt % 2 == 0      # v2: the last K tile (buffer 1) stages the next tile's K tile 0 into buffer 0
        assert not (two and v2)
        segs = []

        def issue(j):
            m0, n0 = tiles[j]
            if self.pass_ != "rest":       # (the early-DMA mode runs every interval twice: bookkeeping in its first pass only)
                if j > 0:
                    wv.done.append((self.m0_of[wv.wave], self.n0_of[wv.wave], wv.acc.copy()))
                self.m0_of[wv.wave], self.n0_of[wv.wave] = m0, n0
                self.retarget(wv, m0, n0)
            if not (prefetch and j > 0):
                self.stage_w(wv, 0, 0, 0)
                self.stage_x(wv, 0, 0, 0)
                self.stage_w(wv, 0, 0, 1)
                self.stage_x(wv, 0, 0, 1)
            if nt > 1:
                self.stage_w(wv, 1, 1, 0)
                self.stage_x(wv, 1, 1, 0)
                self.stage_w(wv, 1, 1, 1)
                self.wait_vm(wv, 10 if early_start else 6)
            else:
                self.wait_vm(wv, 0)

        for j in range(len(tiles)):
            segs.append(lambda j=j: issue(j))
            if wv.wrow == 1:
                segs.append(lambda: None)
            t = 0
            while t + 1 < nt:
                if two:
                    segs += self.tile_segments_sched(wv, 0, t, True, t + 2 < nt, sched=self.sched)
                    segs += self.tile_segments_sched(wv, 1, t + 1, t + 2 < nt, t + 3 < nt, sched=self.sched)
                    t += 2
                    continue
                segs += self.tile_segments(wv, 0, t, True, t + 2 < nt, first=(t == 0 and early_start))
                pfn = prefetch and t + 2 == nt and j + 1 < len(tiles)
                segs += self.tile_segments_b1(wv, t + 1, t + 2 < nt, t + 3 < nt, tiles[j + 1] if pfn else None)
                t += 2
            if t < nt:
                segs += self.tile_segments_sched(wv, 0, t, False, False, sched=self.sched) if two else self.tile_segments(wv, 0, t, False, False)
            if wv.wrow == 0:
                segs.append(lambda: None)
        segs.append(lambda: wv.done.append((self.m0_of[wv.wave], self.n0_of[wv.wave], wv.acc.copy())) if self.pass_ != "rest" else None)
        return segs

    def run_persist(self, tiles, v2=False):
        self.m0_of, self.n0_of = {}, {}
        progs = [self.program_persist(wv, tiles, v2) for wv in self.waves]
        self._run_programs(progs)

    def program(self, wv):
        nt = self.K // BK
        segs = []
        if self.rows_mode:
            def prologue_rows():
                self.stage_w(wv, 0, 0, 0)
                self.stage_x(wv, 0, 0, 0)
                self.stage_w(wv, 0, 0, 1)
                if nt > 1:
                    self.stage_w(wv, 1, 1, 0)
                    self.stage_x(wv, 1, 1, 0)
                    self.wait_vm(wv, 4)
                else:
                    self.wait_vm(wv, 0)
            segs.append(prologue_rows)
            if wv.wrow == 1:
                segs.append(lambda: None)
            t = 0
            while t + 1 < nt:
                segs += self.tile_segments_rows(wv, 0, t, True, t + 2 < nt)
                segs += self.tile_segments_rows(wv, 1, t + 1, t + 2 < nt, t + 3 < nt)
                t += 2
            if t < nt:
                segs += self.tile_segments_rows(wv, 0, t, False, False)
            if wv.wrow == 0:
                segs.append(lambda: None)
            return segs
        if self.half_mode and self.epi == 1 and self.n0 + BN >= self.I:      # half tile: the gate half of this tile is beyond N
            def prologue_half():
                self.stage_w(wv, 0, 0, 0)
                self.stage_x(wv, 0, 0, 0)
                self.stage_x(wv, 0, 0, 1)
                if nt > 1:
                    self.stage_w(wv, 1, 1, 0)
                    self.stage_x(wv, 1, 1, 0)
                    self.wait_vm(wv, 4)
                else:
                    self.wait_vm(wv, 0)
            segs.append(prologue_half)
            if wv.wrow == 1:
                segs.append(lambda: None)
            t = 0
            while t + 1 < nt:
                segs += self.tile_segments_half(wv, 0, t, True, t + 2 < nt)
                segs += self.tile_segments_half(wv, 1, t + 1, t + 2 < nt, t + 3 < nt)
                t += 2
            if t < nt:
                segs += self.tile_segments_half(wv, 0, t, False, False)
            if wv.wrow == 0:
                segs.append(lambda: None)
            return segs

        def prologue():
            self.stage_w(wv, 0, 0, 0)
            self.stage_x(wv, 0, 0, 0)
            self.stage_w(wv, 0, 0, 1)
            self.stage_x(wv, 0, 0, 1)
            if nt > 1:
                self.stage_w(wv, 1, 1, 0)
                self.stage_x(wv, 1, 1, 0)
                self.stage_w(wv, 1, 1, 1)
                self.wait_vm(wv, 10 if early_start else 6)   # early start: only the 4 DMAs phase 1 reads
            else:
                self.wait_vm(wv, 0)
        early_start = nt >= 3 and self.breakage != "noearly" and not (self.sched and SCHEDS[self.sched][2])
        segs.append(prologue)
        if wv.wrow == 1:
            segs.append(lambda: None)          # the extra barrier of the second wave row
        ts = (lambda *a, **k: self.tile_segments_sched(*a, sched=self.sched, **k)) if self.sched else self.tile_segments
        t = 0
        while t + 1 < nt:
            segs += ts(wv, 0, t, True, t + 2 < nt, first=(t == 0 and early_start))
            segs += ts(wv, 1, t + 1, t + 2 < nt, t + 3 < nt)
            t += 2
        if t < nt:
            segs += ts(wv, 0, t, False, False)
        if wv.wrow == 0:
            segs.append(lambda: None)          # pairs the extra barrier
        return segs

    def run(self):
        self._run_programs([self.program(wv) for wv in self.waves])

    def _run_programs(self, progs):
        n = len(progs[0])
        assert all(len(p) == n for p in progs), "wave rows execute different numbers of barriers"
        for seg in range(n):
            # interval between barrier seg-1 and barrier seg: every wave runs its segment `seg`; order inside an interval is
            # free on the hardware, the two modes make the adversarial choice (module docstring)
            if self.mode == "dma_early_read_late":
                for self.pass_ in ("issue_dma", "rest"):      # every DMA of the interval lands before any read is sampled
                    for w in range(8):
                        progs[w][seg]()
            else:
                self.pass_ = "all"
                for w in (range(7, -1, -1) if self.flip else range(8)):
                    progs[w][seg]()
                for base, vals in self.landing:   # everything retired in this interval is visible after the barrier
                    self.land(base, vals)
                self.landing = []
        for wv in self.waves:
            assert not wv.vmq and not wv.pending, "wave ended with loads outstanding"

    pass_ = "all"
    flip = False
    sched = None
    half_mode = False

    # ---- epilogue ---------------------------------------------------------------------------------------------------
    def epilogue(self, out_lin, out):
        for wv in self.waves:
            lane = LANES
            ncol = self.n0 + 32 * wv.wcol + 8 * (lane >> 4)
            for mb in range(4 if self.rows_mode else 8):
                m = self.m0 + (64 if self.rows_mode else 128) * wv.wrow + 16 * mb + (lane & 15)
                for l in range(64):
                    if m[l] >= self.M:
                        continue
                    for nf in range(2):
                        for j in range(4):
                            n = ncol[l] + 4 * nf + j
                            if self.epi == 0:
                                v = wv.acc[mb][nf][l, j] + self.bias[n]
                                g = wv.acc[mb][2 + nf][l, j] + self.bias[self.I + n]
                                out_lin[m[l], n] = v
                                out_lin[m[l], self.I + n] = g
                                out[m[l], n] = np.float32(v) * gelu_as(np.float32(g))
                            else:
                                for half, ok in ((0, ncol[l] < self.I), (1, ncol[l] + self.gap < self.I)):
                                    if not ok:       # ok_v / ok_g of the kernel: a whole 8-column group is in or out
                                        continue
                                    col = n + half * self.gap
                                    v = wv.acc[mb][2 * half + nf][l, j] + self.bias[col]
                                    if self.row_bias is not None:
                                        v += self.row_bias[m[l] // self.rows_per_sample, col]
                                    if self.residual is not None:
                                        v += self.residual[m[l], col]
                                    out_lin[m[l], col] = v


def run_persist_case(M, K, N, mode, G, flip=False, seed=0, v2=False, breakage=None, sched=None):
    """Plain projection through the persistent variant with a grid of G workgroups (tile ids b, b + G, ...)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(-4, 5, size=(M, K)).astype(np.float64)
    w = rng.integers(-4, 5, size=(N, K)).astype(np.float64)
    bias = rng.integers(-8, 9, size=N).astype(np.float64) / 4
    lin = np.full((M, N), np.nan)
    nbn = -(-N // (2 * BN))
    nb = -(-M // BM) * nbn

    def tile_of(bid):
        q, r, xcd = nb >> 3, nb & 7, bid & 7
        tid = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (bid >> 3)
        nbm = nb // nbn
        per_group = 8 * nbn
        first = (tid // per_group) * 8
        rows_here = min(nbm - first, 8)
        return (first + (tid % per_group) % rows_here) * BM, ((tid % per_group) // rows_here) * 2 * BN

    seen = set()
    for b in range(min(G, nb)):
        tiles = [tile_of(bid) for bid in range(b, nb, G)]
        seen.update(tiles)
        blk = Block(x, w, bias, M, K, N, tiles[0][0], tiles[0][1], mode, breakage, 1, None)
        blk.flip = flip
        blk.sched = sched
        blk.run_persist(tiles, v2)
        for wv in blk.waves:
            assert len(wv.done) == len(tiles)
        for j, (m0, n0) in enumerate(tiles):
            blk.m0, blk.n0 = m0, n0
            saved = [wv.acc for wv in blk.waves]
            for wv in blk.waves:
                assert wv.done[j][:2] == (m0, n0)
                wv.acc = wv.done[j][2]
            blk.epilogue(lin, None)
            for wv, a in zip(blk.waves, saved):
                wv.acc = a
    assert len(seen) == nb
    return np.array_equal(lin, x @ w.T + bias)


def gelu_as(x):
    x = np.float32(x)
    z = np.float32(abs(x)) * np.float32(0.70710678118654752)
    t = np.float32(1.0) / (np.float32(0.3275911) * z + np.float32(1.0))
    p = np.float32(1.061405429) * t + np.float32(-1.453152027)
    p = p * t + np.float32(1.421413741)
    p = p * t + np.float32(-0.284496736)
    p = p * t + np.float32(0.254829592)
    q = p * t * np.float32(2.0) ** (np.float32(-1.4426950408889634) * z * z)
    h = np.float32(0.5) * x * q
    return x - h if x > 0 else h


def run_case(M, K, I, mode, flip=False, breakage=None, seed=0, epi=0, conv=None, addends=False, sched=None, half=False, rows=False):
    """epi 0: GEGLU (W [2 I, K]); epi 1: plain projection, I = output columns (W [I, K]); conv = (B, H, W): 3x3 convolution of an
    NHWC image with Cin = K / 9 as an implicit GEMM (M = B H W)."""
    rng = np.random.default_rng(seed)
    wrows = 2 * I if epi == 0 else I
    x = rng.integers(-4, 5, size=(M, K // 9 if conv else K)).astype(np.float64)
    w = rng.integers(-4, 5, size=(wrows, K)).astype(np.float64)
    bias = rng.integers(-8, 9, size=wrows).astype(np.float64) / 4
    lin = np.full((M, wrows), np.nan)
    out = np.full((M, I), np.nan)
    rps = conv[1] * conv[2] if conv else M          # rows per sample: H W for a convolution, one sample otherwise
    row_bias = rng.integers(-8, 9, size=(M // rps, wrows)).astype(np.float64) / 4 if addends else None
    residual = rng.integers(-8, 9, size=(M, wrows)).astype(np.float64) / 4 if addends else None
    nbn = I // BN if epi == 0 else -(-I // (2 * BN))
    TBM = BM // 2 if rows else BM
    nb = -(-M // TBM) * nbn
    seen = set()
    for bid in range(nb):
        q, r, xcd = nb >> 3, nb & 7, bid & 7
        tid = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (bid >> 3)
        nbm = nb // nbn
        per_group, = (8 * nbn,)
        first = (tid // per_group) * 8
        rows_here = min(nbm - first, 8)
        rb, cb = first + (tid % per_group) % rows_here, (tid % per_group) // rows_here
        assert 0 <= rb < nbm and 0 <= cb < nbn
        seen.add((rb, cb))
        blk = Block(x, w, bias, M, K, I, rb * TBM, cb * (BN if epi == 0 else 2 * BN), mode, breakage, epi, conv[1:] if conv else None,
                    row_bias, residual, rps, rows_mode=rows)
        blk.flip = flip
        blk.sched = sched
        blk.half_mode = half
        blk.run()
        blk.epilogue(lin, out)
    assert len(seen) == nb, "workgroup remap is not a bijection"
    if conv:
        Bn, H, Wd = conv
        Cin = K // 9
        img = np.zeros((Bn, H + 2, Wd + 2, Cin))
        img[:, 1:-1, 1:-1] = x.reshape(Bn, H, Wd, Cin)
        cols = np.concatenate([img[:, dy:dy + H, dx:dx + Wd] for dy in range(3) for dx in range(3)], axis=-1)   # [B,H,W,9 Cin], tap-major
        ref = cols.reshape(M, K) @ w.T + bias
    else:
        ref = x @ w.T + bias
    if addends:
        ref = ref + np.repeat(row_bias, rps, axis=0) + residual
    ok_lin = np.array_equal(lin, ref)
    if epi == 1:
        return ok_lin, 0.0
    gel = 0.5 * ref[:, I:] * (1 + np.vectorize(math.erf)(ref[:, I:] / math.sqrt(2)))
    want = ref[:, :I] * gel
    # the polynomial's absolute error (5e-7) is multiplied by the value branch: normalise by it
    err = np.nanmax(np.abs(out - want) / (1 + np.abs(ref[:, :I]) * (1 + np.abs(ref[:, I:])))) if not np.isnan(out).any() else float("nan")
    return ok_lin, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--break", dest="breakage", choices=["war", "raw", "lgkm", "early", "pf", "half_raw", "rows_raw"], default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--persist", action="store_true", help="replay the persistent experiment (tools/gemm_persist) instead")
    ap.add_argument("--sched-in-persist", action="store_true", help="with --persist: the 4-interval K loop inside the persistent loop")
    ap.add_argument("--persist2", action="store_true", help="... its v2: the next tile's K tile 0 staged during the last K tile")
    ap.add_argument("--sched", default=None, help="replay a schedule descriptor of tools/gemm_sched (name, or 'all')")
    ap.add_argument("--half", action="store_true", help="replay the half-tile mode of tools/gemm_sched (value half only where the gate half is beyond N)")
    ap.add_argument("--rows", action="store_true", help="replay the 128-row tile mode (round 6: tile_phases_rows)")
    a = ap.parse_args()
    if a.breakage == "rows_raw":
        caught = 0
        for mode in ("dma_early_read_late", "dma_late_read_early"):
            try:
                ok, _ = run_case(256, 448, 256, mode, breakage="rows_raw", epi=1, rows=True)
            except AssertionError:
                ok = False
            caught += not ok
        print("replay", "caught the deliberately broken schedule" if caught else "DID NOT catch the broken schedule")
        sys.exit(0 if caught else 1)
    if a.rows:
        bad = 0
        for (M, K, N) in [(300, 256, 104), (128, 192, 320), (200, 448, 384), (128, 64, 640), (260, 128, 200), (128, 512, 256)][:None if not a.quick else 3]:
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                for flip in ((False, True) if mode == "dma_late_read_early" else (False,)):
                    ok, _ = run_case(M, K, N, mode, flip, epi=1, rows=True)
                    print(f"128-row tiles M={M} K={K} ({K // BK} K tiles) N={N} {mode:>20s}{' flipped' if flip else ''}: {'exact' if ok else 'WRONG'}")
                    bad += not ok
        for (Bn, H, Wd, Cin, N) in [(2, 12, 12, 64, 320), (1, 9, 20, 128, 104)]:
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                ok, _ = run_case(Bn * H * Wd, 9 * Cin, N, mode, epi=1, conv=(Bn, H, Wd), rows=True)
                ok2, _ = run_case(Bn * H * Wd, 9 * Cin, N, mode, epi=1, conv=(Bn, H, Wd), addends=True, seed=1, rows=True)
                print(f"128-row tiles conv3x3 B={Bn} {H}x{Wd} Cin={Cin} N={N} {mode:>20s}: {'exact' if ok else 'WRONG'}; with addends: {'exact' if ok2 else 'WRONG'}")
                bad += (not ok) + (not ok2)
        sys.exit(1 if bad else 0)
    if a.breakage == "half_raw":
        caught = 0
        for mode in ("dma_early_read_late", "dma_late_read_early"):
            try:
                ok, _ = run_case(256, 448, 320, mode, breakage="half_raw", epi=1, half=True)
            except AssertionError:
                ok = False
            caught += not ok
        print("replay", "caught the deliberately broken schedule" if caught else "DID NOT catch the broken schedule")
        sys.exit(0 if caught else 1)
    if a.half:
        bad = 0
        for (M, K, N) in [(300, 256, 104), (256, 192, 320), (300, 448, 384), (256, 64, 640), (256, 128, 200),      # 200: no half tile (control)
                           (256, 64, 320), (256, 128, 320), (300, 320, 104), (256, 512, 320)][:None if not a.quick else 3]:    # 1, 2, 5, 8 K tiles
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                for flip in ((False, True) if mode == "dma_late_read_early" else (False,)):
                    ok, _ = run_case(M, K, N, mode, flip, epi=1, half=True)
                    print(f"half tiles M={M} K={K} ({K // BK} K tiles) N={N} {mode:>20s}{' flipped' if flip else ''}: {'exact' if ok else 'WRONG'}")
                    bad += not ok
        for (Bn, H, Wd, Cin, N) in [(2, 12, 12, 64, 320), (1, 9, 20, 128, 104)]:
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                ok, _ = run_case(Bn * H * Wd, 9 * Cin, N, mode, epi=1, conv=(Bn, H, Wd), half=True)
                ok2, _ = run_case(Bn * H * Wd, 9 * Cin, N, mode, epi=1, conv=(Bn, H, Wd), addends=True, seed=1, half=True)
                print(f"half tiles conv3x3 B={Bn} {H}x{Wd} Cin={Cin} N={N} {mode:>20s}: {'exact' if ok else 'WRONG'}; with addends: {'exact' if ok2 else 'WRONG'}")
                bad += (not ok) + (not ok2)
        sys.exit(1 if bad else 0)
    if a.sched:
        bad = 0
        for name in (list(SCHEDS) if a.sched == "all" else [a.sched]):
            want_bad = name.startswith("bad_")
            caught = 0
            for (M, K, N) in [(300, 448, 200), (256, 384, 256), (256, 128, 256), (256, 64, 128)]:
                for mode in ("dma_early_read_late", "dma_late_read_early"):
                    for flip in ((False, True) if mode == "dma_late_read_early" else (False,)):
                        try:
                            ok, _ = run_case(M, K, N, mode, flip, epi=1, sched=name)
                        except AssertionError as e:
                            ok = False
                        caught += not ok
                        if not want_bad:
                            print(f"sched {name:>10s} M={M} K={K} ({K // BK} K tiles) N={N} {mode:>20s}{' flipped' if flip else ''}: {'exact' if ok else 'WRONG'}")
                            bad += not ok
            if name in ("product", "two_read"):      # the convolution runs these two (gemm_sched.hip: Conv<S_product>, Conv<S2_read>)
                for (Bn, H, Wd, Cin, N) in [(2, 12, 12, 64, 128), (1, 9, 20, 128, 200)]:
                    for mode in ("dma_early_read_late", "dma_late_read_early"):
                        ok, _ = run_case(Bn * H * Wd, 9 * Cin, N, mode, epi=1, conv=(Bn, H, Wd), sched=name)
                        print(f"sched {name:>10s} conv3x3 B={Bn} {H}x{Wd} Cin={Cin} N={N} {mode:>20s}: {'exact' if ok else 'WRONG'}")
                        bad += not ok
            if want_bad:
                print(f"sched {name}: replay", "caught the deliberately broken schedule" if caught else "DID NOT catch the broken schedule")
                bad += not caught
        sys.exit(1 if bad else 0)
    if a.breakage == "pf":
        a.persist2 = True
    if a.persist and a.sched_in_persist:       # the persistent loop around the 4-interval K loop (r5_patches/0006)
        bad = 0
        for (M, K, N, G) in [(768, 192, 768, 4), (512, 256, 512, 3), (300, 64, 640, 2), (512, 448, 512, 3)]:
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                for flip in ((False, True) if mode == "dma_late_read_early" else (False,)):
                    ok = run_persist_case(M, K, N, mode, G, flip, sched="two_read")
                    print(f"persistent + 4-interval loop M={M} K={K} ({K // BK} K tiles) N={N} grid {G} {mode:>20s}{' flipped' if flip else ''}: {'exact' if ok else 'WRONG'}")
                    bad += not ok
        sys.exit(1 if bad else 0)
    if a.persist or a.persist2:
        bad = 0
        shapes = [(768, 192, 768, 4), (512, 256, 512, 3), (300, 64, 640, 2)]
        if a.persist2:      # 6 and 8 K tiles prefetch; 7 (odd) and 4 (short) keep v1's order
            shapes = [(768, 384, 768, 4), (512, 512, 512, 3), (512, 448, 512, 3), (300, 256, 640, 2)] if not a.breakage else [(512, 384, 512, 3)]
        for (M, K, N, G) in shapes:
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                for flip in ((False, True) if mode == "dma_late_read_early" else (False,)):
                    try:
                        ok = run_persist_case(M, K, N, mode, G, flip, v2=a.persist2, breakage=a.breakage)
                    except AssertionError as e:
                        if not a.breakage:
                            raise
                        ok = False
                    print(f"persistent{' v2' if a.persist2 else ''} M={M} K={K} ({K // BK} K tiles) N={N} grid {G} {mode:>20s}"
                          f"{' flipped' if flip else ''}: {'exact' if ok else 'WRONG'}")
                    bad += not ok
        if a.breakage:
            print("replay", "caught the deliberately broken schedule" if bad else "DID NOT catch the broken schedule")
            sys.exit(0 if bad else 1)
        sys.exit(1 if bad else 0)
    # gelu polynomial against erf, float64
    xs = np.linspace(-12, 12, 48001)
    ge = np.array([gelu_as(v) for v in xs], np.float64)
    gr = 0.5 * xs * (1 + np.vectorize(math.erf)(xs / math.sqrt(2)))
    print(f"gelu_as vs erf: max abs err {np.abs(ge - gr).max():.3e} (fp16 output ulp at 1.0 is 9.8e-4)")
    assert np.abs(ge - gr).max() < 2e-6
    cases = [(300, 192, 256), (256, 64, 128), (512, 256, 128), (256, 640, 128)] if not a.quick else [(300, 256, 128)]
    bad = 0
    for (M, K, I) in cases:
        for mode in ("dma_early_read_late", "dma_late_read_early"):
            for flip in ((False, True) if mode == "dma_late_read_early" else (False,)):
                ok, err = run_case(M, K, I, mode, flip, a.breakage)
                verdict = "exact" if ok else "WRONG"
                print(f"M={M} K={K} ({K // BK} tiles) I={I} {mode:>20s}{' flipped' if flip else ''}: projection {verdict}, "
                      f"geglu rel err {err:.2e}")
                bad += (not ok) or not (err < 2e-6)
    if not a.breakage and not a.quick:
        for (M, K, N) in [(300, 128, 640), (256, 192, 200), (256, 64, 256)]:     # plain projection: ragged column blocks
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                ok, _ = run_case(M, K, N, mode, epi=1)
                print(f"linear M={M} K={K} N={N} {mode:>20s}: projection {'exact' if ok else 'WRONG'}")
                bad += not ok
    if not a.breakage:
        for (Bn, H, Wd, Cin, N) in ([(2, 12, 12, 64, 128)] if a.quick else [(2, 12, 12, 64, 128), (1, 9, 20, 128, 200), (3, 8, 8, 64, 256)]):
            for mode in ("dma_early_read_late", "dma_late_read_early"):
                ok, _ = run_case(Bn * H * Wd, 9 * Cin, N, mode, epi=1, conv=(Bn, H, Wd))
                ok2, _ = run_case(Bn * H * Wd, 9 * Cin, N, mode, epi=1, conv=(Bn, H, Wd), addends=True, seed=1)
                print(f"conv3x3 B={Bn} {H}x{Wd} Cin={Cin} N={N} {mode:>20s}: {'exact' if ok else 'WRONG'}; with per-sample bias + "
                      f"residual (batch seams inside a tile): {'exact' if ok2 else 'WRONG'}")
                bad += (not ok) + (not ok2)
    if a.breakage:
        print("replay", "caught the deliberately broken schedule" if bad else "DID NOT catch the broken schedule")
        sys.exit(0 if bad else 1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
