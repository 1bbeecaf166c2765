"""Host-side random draws of the hot path.

Parity target = the reference constructed with ``device=cpu``: every draw comes from the torch CPU
generator and numpy's global MT19937, in the reference's order.  None of the draws depends on a UNet output -- only on
earlier draws -- so the host can run arbitrarily far ahead of the GPU; the results (8 KiB of pick indices per
resampling step, the initial latent, the RePaint noise) are staged in pinned memory and uploaded asynchronously.
Nothing here synchronises with the device.

Reference lines ("ED:n" = /root/reference/elastic_diffusion.py):
  pick indices     random_sample_exclude_mask ED:501-520, random_downsample ED:534-544, exclude mask ED:673-675
  pad-strip reseed make_denoised_background ED:331-335, 359 (md5 seed -> draws -> reseed from numpy)
  RePaint noise    undo_step ED:692-704
"""
import hashlib

import numpy as np
import torch


MAX_RESAMPLING_STEPS = 126  # K = R+1 steps are stamped into an int8 table (values -1 .. K-1 <= 126)


def seed_everything(seed, seed_np=True):
    """ED:165-171 (the reference also seeds the CUDA generator; device-side generators are never used here)."""
    torch.manual_seed(seed)
    if seed_np:
        np.random.seed(seed)


def strip_seed(dim, side, Hs, Ws, t):
    """md5 -> first 4 bytes (ED:321-324) of the id string built at ED:331, 381, 386; ``t`` must print like the
    reference's 0-d int64 CPU tensor ("tensor(981)")."""
    ident = f"{dim}_{side}_{Hs}_{Ws}_{t}"
    return int(hashlib.md5(ident.encode()).hexdigest()[:8], 16)


def strip_draws(dim, side, Hs, Ws, t, latent_channels=4):
    """The three draws make_denoised_background takes from the md5-seeded generator (ED:335-356): the background
    colour, the VAE posterior noise and the forward-diffusion noise.  A private generator seeded identically
    yields the same numbers as the reference's re-seeded global generator."""
    g = torch.Generator().manual_seed(strip_seed(dim, side, Hs, Ws, t))
    colour = torch.rand(1, 3, generator=g)
    post = torch.randn(1, latent_channels, Hs, Ws, generator=g)
    fwd = torch.randn(1, latent_channels, Hs, Ws, generator=g)
    return colour, post, fwd


def replay_strip_reseeds(n_nonempty_strips):
    """Side effect every non-empty strip leaves on the GLOBAL generators: one numpy draw, and the torch generator
    re-seeded from it (ED:359).  The md5 seeding and the draws in between are overwritten by this re-seed, so they
    need not be replayed."""
    for _ in range(n_nonempty_strips):
        # == torch.manual_seed for the CPU generator (the only one the path draws from); torch.manual_seed would
        # also re-seed every device generator, ~100x the cost, 100+ times per image
        torch.default_generator.manual_seed(int(np.random.randint(100000)))


class PickSampler:
    """Per call of approximate_latent_direction_w_resampling (ED:649-690): the K = R+1 pick-index vectors."""

    def __init__(self, N):
        self.N = N
        self.rows = torch.arange(N)
        self.base = self.rows * 4

    def _sample_excluding(self, exclude, hi=4, max_iteration=50):
        """ED:501-520.  Every draw is a ``torch.randint`` on the global CPU generator with the reference's sizes, in
        the reference's order.  The mask bookkeeping uses flat 1-D gathers / masked_scatter_: torch's 2-D advanced
        indexing ``mask[arange(N), idx]`` costs milliseconds per call once torch has many intra-op threads (it did on
        the 256-core GPU box), and this loop runs ~40 rounds per resampling step."""
        flat = exclude.view(-1)
        idx = torch.randint(0, hi, (self.N,))
        bad = flat.index_select(0, self.base + idx)
        m = int(torch.count_nonzero(bad))
        while m > 0 and max_iteration > 0:
            idx.masked_scatter_(bad, torch.randint(0, hi, (m,)))  # == idx[bad] = randint(...): filled in index order
            bad = flat.index_select(0, self.base + idx)
            m = int(torch.count_nonzero(bad))
            max_iteration -= 1
        if m > 0:  # every choice excluded for some pixels: unconstrained redraw (ED:514-518)
            idx.masked_scatter_(bad, torch.randint(0, hi, (m,)))
        return idx

    def draw(self, K, drop_p, after_step, out=None, stamp=None):
        """-> uint8 [K, N].  ``after_step()`` is invoked after each step's draws, where the reference runs the
        global UNet call whose pad strips re-seed the generators.  ``stamp`` (int8 [N,4], optional) receives, per
        reduced pixel and choice q, the last step that picked q (-1 = never): the K pick masks folded into one table
        for ed_fill_directions."""
        if out is None:
            out = torch.empty(K, self.N, dtype=torch.uint8)
        if stamp is not None:
            stamp.fill_(-1)
        exclude = torch.zeros(self.N, 4, dtype=torch.bool)
        prev = None
        for k in range(K):
            if k == 0:
                idx = torch.zeros(self.N, dtype=torch.long)  # nearest=True on the first step (ED:669, 535-536)
            else:
                idx = self._sample_excluding(exclude)
            if prev is not None:
                drop = torch.randint(0, 101, (self.N,))
                drop[drop <= 100 * drop_p] = 0
                drop[drop >= 100 * drop_p] = 1
                idx = idx * drop + prev * (1 - drop)
            flat_pos = self.base + idx
            exclude.view(-1).index_fill_(0, flat_pos, True)
            prev = idx
            out[k].copy_(idx)
            if stamp is not None:
                stamp.view(-1).index_fill_(0, flat_pos, k)
            after_step()
        return out


def draw_noise_into(buf):
    """One ``torch.randn(sample.shape)`` per leading index of ``buf`` (ED:701), written in place into (pinned) memory.
    ``normal_()`` on a slice runs the same CPU kernel as ``torch.randn`` of that shape."""
    for k in range(buf.shape[0]):
        buf[k].normal_()
    return buf
