"""Schedule-level emulation of the software-pipelined attention kernels (csrc/attention_kernels.hip, k_flash_attn_pipe):
what a container without a GPU can check about them.

1. ``lds_hazards``: the kernel's LDS traffic as a list of inter-barrier intervals.  All waves of a workgroup are always
   inside the SAME interval (nobody passes barrier b+1 before everybody reached it), in any interleaving, so the rule is:
   within one interval no buffer may be both read and written (the write of one wave may land before or after the read
   of another), and every read must find the tile it expects.  The exact kernel is hazard-free with 2 K buffers; the
   lazy-maximum variant re-reads K(t) in its slow path while other waves already stage K(t+2) -- with 2 K buffers that
   is the race the round-3 accuracy tests caught on the GPU (buffer of K(t) == buffer K(t+2) is written into); with 3 it
   is clean.  The buffer rotation below is the kernel's (kb_cur / kb_next / kb_write, vb_prev / vb_cur / vb_next).

2. ``softmax_schedule``: the softmax of one 64-key tile as the kernel slices it (16 slices behind the 16 MFMAs; flat value
   index f = 16 kb + r), the deferred-rescale algebra (reference maximum in the exp2 domain, alpha, l) and the lazy variant
   (no maximum after the first tile, redo when a lane's numerator sum exceeds 2^6), in fp64 against a plain softmax(S) V --
   every value must be exponentiated exactly once, packed in the group the second MFMA expects, and every rescale applied
   exactly once to everything accumulated before it.

    python tools/emulate_attention_pipeline.py
"""
import numpy as np

RESCALE_LOG2 = 6.0


# ---------------------------------------------------------------------------------------------------------------------
# 1. LDS hazards
# ---------------------------------------------------------------------------------------------------------------------
def lds_intervals(n_tiles, n_full, NK, lazy, slow_path_tiles=()):
    """-> list of intervals; an interval = list of (op, buffer, tile) with op in {"r", "w"}.  Mirrors the kernel's order:
    prologue | barrier | qk(tile 0) | barrier | iteration 0 .. n_full-1 (each closed by a barrier) | drain + ragged tail."""
    iv = []
    cur = [("w", ("k", 0), 0), ("w", ("v", 0), 0)]
    if n_tiles > 1:
        cur.append(("w", ("k", 1), 1))
    iv.append(cur)                                            # __syncthreads()
    if n_full == 0:
        iv.append([("r", ("k", 0), 0), ("r", ("v", 0), 0)])   # ragged tail only
        return iv
    iv.append([("r", ("k", 0), 0)])                            # prologue qk_tile(tile 0); __syncthreads()
    kb_cur, kb_next, kb_write = 0, 1, (2 if NK == 3 else 0)
    vb_prev, vb_cur, vb_next = 2, 0, 1
    for t in range(n_full):
        cur = []
        if t + 1 < n_full:
            cur.append(("r", ("k", kb_next), t + 1))           # S(t+1) = K(t+1) Q^T
        if t > 0:
            cur.append(("r", ("v", vb_prev), t - 1))           # O += V(t-1)^T P(t-1)^T
        if lazy and t > 0 and t in slow_path_tiles:
            cur.append(("r", ("k", kb_cur), t))                # slow path: S(t) recomputed from K(t)
        cur.append(("w", ("k", kb_write), t + 2))              # staging (unconditional; past-the-end tiles are zeros)
        cur.append(("w", ("v", vb_next), t + 1))
        iv.append(cur)                                         # __syncthreads()
        vb_prev, vb_cur, vb_next = vb_cur, vb_next, vb_prev
        kb_cur, kb_next, kb_write = kb_next, kb_write, (kb_cur if NK == 3 else kb_next)
    tail = [("r", ("v", vb_prev), n_full - 1)]                 # drain
    if n_tiles > n_full:
        tail += [("r", ("k", kb_cur), n_full), ("r", ("v", vb_cur), n_full)]
    iv.append(tail)
    return iv


def lds_hazards(n_tiles, n_full, NK, lazy, slow_path_tiles=()):
    """-> list of human-readable hazards (empty = clean)."""
    content, problems = {}, []
    for i, ops in enumerate(lds_intervals(n_tiles, n_full, NK, lazy, slow_path_tiles)):
        reads = [(b, t) for op, b, t in ops if op == "r"]
        writes = [(b, t) for op, b, t in ops if op == "w"]
        for b, t in reads:
            if any(wb == b for wb, _ in writes):
                problems.append(f"interval {i}: {b} is read (tile {t}) and written in the same interval")
            if content.get(b) != t:
                problems.append(f"interval {i}: read of {b} expects tile {t}, buffer holds {content.get(b)}")
        for b, t in writes:
            content[b] = t
    return problems


# ---------------------------------------------------------------------------------------------------------------------
# 2. the sliced softmax + deferred rescale, one query row, fp64
# ---------------------------------------------------------------------------------------------------------------------
def exact_slices(s, other_half_max, sl, st):
    """softmax_slice for i = 0..15 on the 32 scores ``s`` of one lane (two 16-register blocks); ``other_half_max`` is what
    slice 2's ``__shfl_xor(mx, 32)`` brings in from the lane holding the row's other 32 keys.  -> (p[32], alpha)."""
    s = s.copy()
    done = np.zeros(32, dtype=int)
    packed, mx, alpha, psum = {}, -np.inf, 1.0, 0.0
    for i in range(16):
        if i < 2:
            mx = max(mx, s[16 * i:16 * i + 16].max())
        elif i == 2:
            mbn = max(mx, other_half_max) * sl
            st["use"] = mbn if (mbn - st["mb"] > RESCALE_LOG2) else st["mb"]   # first tile: mb = -inf -> mbn
            alpha = 2.0 ** (st["mb"] - st["use"])                               # 2^-inf = 0 on the first tile
            st["mb"], psum = st["use"], 0.0
        elif i < 15:
            g, w = (i - 3) // 3, (i - 3) % 3
            f0, n = 8 * g + 3 * w, (2 if w == 2 else 3)
            for f in range(f0, f0 + n):
                s[f] = 2.0 ** (s[f] * sl - st["use"])
                done[f] += 1
                psum += s[f]
            if w == 2:
                assert done[8 * g:8 * g + 8].tolist() == [1] * 8, "group packed before all its values were exponentiated"
                packed[g] = s[8 * g:8 * g + 8].copy()
        else:
            st["l"] = st["l"] * alpha + psum
    assert done.tolist() == [1] * 32 and sorted(packed) == [0, 1, 2, 3]
    return np.concatenate([packed[g] for g in range(4)]), alpha


def lazy_slices(s, sl, st):
    """softmax_slice_lazy for i = 0..15: numerators against the standing reference, S consumed in place.
    -> (p[32], psum); the iteration's check / slow path / l update follow in ``softmax_schedule``."""
    e = s.copy()
    done = np.zeros(32, dtype=int)
    packed, psum = {}, 0.0
    for i in range(16):
        for f in (2 * i, 2 * i + 1):
            e[f] = 2.0 ** (e[f] * sl - st["mb"])
            done[f] += 1
            psum += e[f]
        if i & 3 == 3:
            g = i >> 2
            assert done[8 * g:8 * g + 8].tolist() == [1] * 8
            packed[g] = e[8 * g:8 * g + 8].copy()
    assert done.tolist() == [1] * 32
    return np.concatenate([packed[g] for g in range(4)]), psum


def softmax_schedule(scores, v, sl, lazy):
    """One query row over n tiles of 64 keys as the pipelined kernel orders it: P(t-1) V(t-1) is accumulated BEFORE tile t's
    rescale is applied.  ``scores`` [n, 64]: the row's two lanes (hi = 0 / 1) hold 32 keys each, their own running sum l and
    the SAME reference mb; ``v`` [n, 64, D].  -> (out[D], number of slow-path tiles)."""
    n, D = scores.shape[0], v.shape[2]
    o = np.zeros(D)
    lanes = [dict(mb=-np.inf, l=0.0, use=0.0), dict(mb=-np.inf, l=0.0, use=0.0)]
    p_prev, slow_tiles = None, 0
    for t in range(n):
        halves = [scores[t, :32], scores[t, 32:]]
        if lazy and t > 0:
            fast = [lazy_slices(halves[h], sl, lanes[h]) for h in (0, 1)]
            if all(psum <= 2.0 ** RESCALE_LOG2 for _, psum in fast):   # __any(!(psum <= 2^6)) is false for the wave
                ps, alphas = [p for p, _ in fast], [1.0, 1.0]
                for h in (0, 1):
                    lanes[h]["l"] += fast[h][1]
            else:                                                        # resoftmax_tile: exact, the reference only rises
                slow_tiles += 1
                ps, alphas = [], []
                for h in (0, 1):
                    use = max(lanes[h]["mb"], scores[t].max() * sl)
                    alphas.append(2.0 ** (lanes[h]["mb"] - use))
                    lanes[h]["mb"] = use
                    e = 2.0 ** (halves[h] * sl - use)
                    lanes[h]["l"] = lanes[h]["l"] * alphas[-1] + e.sum()
                    ps.append(e)
        else:
            res = [exact_slices(halves[h], halves[1 - h].max(), sl, lanes[h]) for h in (0, 1)]
            ps, alphas = [r[0] for r in res], [r[1] for r in res]
        assert alphas[0] == alphas[1] and lanes[0]["mb"] == lanes[1]["mb"], "the two lanes of a row must agree"
        if p_prev is not None:
            o += p_prev @ v[t - 1]            # MFMAs of the region: O += V(t-1)^T P(t-1)^T (at the old reference)
        o *= alphas[0]                        # then the (rare) rescale by this tile's alpha
        p_prev = np.concatenate(ps)
    o += p_prev @ v[n - 1]                    # drain
    return o / (lanes[0]["l"] + lanes[1]["l"]), slow_tiles


def exp2_schedule(scores, v):
    """v_path 6 (exponent-domain q: ``scores`` are exponents already, sl = 1): the S chains of tile t+1 start from -mb AS OF
    ITERATION t (the MFMA C operand ``negm``), a numerator is 2^S with no subtraction, and a slow-path tile (lazy check failed)
    moves the reference, the already started S(t+1) and ``negm`` by the same amount.  -> (out[D], slow-path tiles)."""
    n, D = scores.shape[0], v.shape[2]
    mb = scores[0].max()                       # prologue: exact maximum of tile 0 (both lanes, after the shuffle)
    negm = -mb
    s_cur = scores[0] - mb                     # sA -= mx
    l, o, p_prev, slow_tiles = [0.0, 0.0], np.zeros(D), None, 0
    for t in range(n):
        s_next = scores[t + 1] + negm if t + 1 < n else None      # region: S(t+1) = K(t+1) Q'^T + negm
        if p_prev is not None:
            o += p_prev @ v[t - 1]                                 # region: O += V(t-1)^T P(t-1)^T (old reference)
        p = 2.0 ** s_cur                                           # lazy slices: one exp2 per value
        psum, alpha = [p[:32].sum(), p[32:].sum()], 1.0
        if not all(x <= 2.0 ** RESCALE_LOG2 for x in psum):        # __any(!(psum <= 2^6)): resoftmax_tile from the raw scores
            slow_tiles += 1
            use = max(mb, scores[t].max())
            alpha, d, mb = 2.0 ** (mb - use), use - mb, use
            p = 2.0 ** (scores[t] - use)
            psum = [p[:32].sum(), p[32:].sum()]
            if s_next is not None:
                s_next = s_next - d
            negm = -mb
        for h in (0, 1):
            l[h] = l[h] * alpha + psum[h]
        o *= alpha
        p_prev, s_cur = p, s_next
    o += p_prev @ v[n - 1]
    return o / (l[0] + l[1]), slow_tiles


def reference(scores, v, sl):
    x = scores.reshape(-1) * sl
    p = 2.0 ** (x - x.max())
    return (p / p.sum()) @ v.reshape(-1, v.shape[2])


if __name__ == "__main__":
    for NK, lazy in ((2, False), (3, True), (2, True)):
        bad = []
        for n_tiles in range(1, 9):
            for n_full in (n_tiles, n_tiles - 1):
                bad += lds_hazards(n_tiles, n_full, NK, lazy, slow_path_tiles=range(1, n_full))
        print(f"NK={NK} lazy={lazy}: {len(bad)} hazards" + (f" e.g. {bad[0]}" if bad else ""))
    rng = np.random.default_rng(0)
    for lazy in (False, True):
        worst, slow = 0.0, 0
        for trial in range(20):
            n = int(rng.integers(1, 12))
            sc = rng.normal(size=(n, 64)) * 3.0
            if trial % 3 == 0 and n > 2:
                sc[n - 2, 7] += 60.0       # a late tile far above the reference
            if trial % 3 == 1:
                sc += np.arange(n)[:, None] * 2.5   # creeping maximum
            v = rng.normal(size=(n, 64, 8))
            got, s_ = softmax_schedule(sc, v, 0.18, lazy)
            worst = max(worst, np.abs(got - reference(sc, v, 0.18)).max())
            slow += s_
        print(f"lazy={lazy}: max |err| vs plain softmax {worst:.2e}, slow-path tiles {slow}")
    worst, slow = 0.0, 0
    for trial in range(30):
        n = int(rng.integers(1, 12))
        sc = rng.normal(size=(n, 64)) * 3.0
        if trial % 3 == 0 and n > 2:
            sc[n - 2, 7] += 60.0
        if trial % 3 == 1:
            sc += np.arange(n)[:, None] * 2.5
        v = rng.normal(size=(n, 64, 8))
        got, s_ = exp2_schedule(sc, v)
        worst = max(worst, np.abs(got - reference(sc, v, 1.0)).max())
        slow += s_
    print(f"exp2 (v_path 6): max |err| vs plain softmax {worst:.2e}, slow-path tiles {slow}")
