"""Deterministic tiny stand-ins for the UNet / VAE the hot path drives (test infrastructure).

They are injected (a) into the real reference code when tests/golden/make_golden.py generates fixtures,
(b) into the oracle, and (c) into the HIP product path in the ``-m gpu`` parity tests, so that all three run
the *same* model arithmetic and every difference is a glue difference.  The modules expose exactly the attributes
the reference touches (elastic_diffusion.py:156, 161-163, 236-238, 268-270, 338, 350, 422-426).

Design constraints:
  * position sensitive (3x3 conv + fixed positional pattern): an off-by-one in any crop / pad / scatter shows up;
  * depends on the text embedding and the SDXL added conditions: cond - uncond is non-zero, batch-row mix-ups show;
  * depends on t;
  * never touches the global torch / numpy RNG (weights come from a private generator);
  * fp32, bounded output, so a 50-step DDIM loop stays finite.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Out(dict):
    __getattr__ = dict.__getitem__


class FakeUNet(nn.Module):
    def __init__(self, sample_size=64, cross_dim=32, xl=False, pooled_dim=16, seed=7):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.xl = xl
        self.config = SimpleNamespace(sample_size=sample_size, in_channels=4, addition_time_embed_dim=8)
        self.conv = nn.Conv2d(4, 4, 3, padding=1)
        self.emb = nn.Linear(cross_dim, 4)
        with torch.no_grad():
            self.conv.weight.copy_(torch.randn(4, 4, 3, 3, generator=g) * 0.15)
            self.conv.bias.copy_(torch.randn(4, generator=g) * 0.05)
            self.emb.weight.copy_(torch.randn(4, cross_dim, generator=g) * 0.3)
            self.emb.bias.zero_()
        pos = torch.randn(1, 4, sample_size, sample_size, generator=g) * 0.1
        self.register_buffer("pos", pos)
        if xl:
            self.add_embedding = SimpleNamespace(linear_1=SimpleNamespace(in_features=8 * 6 + pooled_dim))
            self.add_w = nn.Parameter(torch.randn(pooled_dim, 4, generator=g) * 0.2)
        for p in self.parameters():
            p.requires_grad_(False)

    @property
    def dtype(self):
        return self.conv.weight.dtype

    def forward(self, x, t, encoder_hidden_states=None, added_cond_kwargs=None,
                down_block_additional_residuals=None, mid_block_additional_residual=None, **_):
        tt = torch.as_tensor(t, dtype=torch.float32, device=x.device) / 1000.0
        if tt.dim() > 0:  # one timestep per row (fused batches of several images in flight)
            tt = tt.view(-1, 1, 1, 1)
        e = self.emb(encoder_hidden_states.to(x.dtype).mean(dim=1))  # (B,4)
        pos = self.pos
        if pos.shape[-2:] != x.shape[-2:]:
            pos = pos[..., : x.shape[-2], : x.shape[-1]]
            pos = F.pad(pos, (0, x.shape[-1] - pos.shape[-1], 0, x.shape[-2] - pos.shape[-2]))
        y = 0.55 * x + 0.3 * self.conv(x) + 0.1 * torch.sin(2.0 * x + tt) + 0.2 * e[:, :, None, None] + pos * (0.5 + tt)
        if self.xl and added_cond_kwargs is not None:
            a = added_cond_kwargs["text_embeds"].to(x.dtype) @ self.add_w  # (B,4)
            ids = added_cond_kwargs["time_ids"].to(x.dtype).sum(dim=1, keepdim=True) * 1e-5
            y = y + 0.1 * (a + ids)[:, :, None, None]
        if down_block_additional_residuals is not None:
            for r in down_block_additional_residuals:
                y = y + 0.05 * r
        if mid_block_additional_residual is not None:
            y = y + 0.05 * mid_block_additional_residual
        return _Out(sample=y)


class FakeControlNet(nn.Module):
    """Returns residuals shaped like the latent so FakeUNet can add them (the real ones are feature maps)."""

    def __init__(self, seed=11):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.mix = nn.Conv2d(3, 4, 3, padding=1)
        with torch.no_grad():
            self.mix.weight.copy_(torch.randn(4, 3, 3, 3, generator=g) * 0.2)
            self.mix.bias.zero_()
        for p in self.parameters():
            p.requires_grad_(False)

    @property
    def dtype(self):
        return self.mix.weight.dtype

    def forward(self, x, t, encoder_hidden_states=None, controlnet_cond=None, conditioning_scale=1.0,
                guess_mode=False, return_dict=False, added_cond_kwargs=None, **_):
        c = F.avg_pool2d(self.mix(controlnet_cond.to(x.dtype)), 8)
        r = conditioning_scale * (c + 0.1 * x)
        return [r, 0.5 * r], 0.25 * r


class _Gaussian:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        # diffusers DiagonalGaussianDistribution.sample: randn_tensor(mean.shape) from the global generator
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise


class FakeVAE(nn.Module):
    def __init__(self, seed=13):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.config = SimpleNamespace(block_out_channels=(1, 1, 1, 1), scaling_factor=0.18215, force_upcast=False)
        self.enc = nn.Conv2d(3, 8, 3, padding=1)
        self.dec = nn.Conv2d(4, 3, 3, padding=1)
        self.post_quant_conv = nn.Conv2d(4, 4, 1)
        with torch.no_grad():
            self.enc.weight.copy_(torch.randn(8, 3, 3, 3, generator=g) * 0.2)
            self.enc.bias.copy_(torch.tensor([0.1, -0.2, 0.3, 0.0, -3.0, -3.5, -2.5, -3.0]))
            self.dec.weight.copy_(torch.randn(3, 4, 3, 3, generator=g) * 0.2)
            self.dec.bias.zero_()
            self.post_quant_conv.weight.copy_(torch.eye(4).view(4, 4, 1, 1))
            self.post_quant_conv.bias.zero_()
        for p in self.parameters():
            p.requires_grad_(False)

    @property
    def dtype(self):
        return self.enc.weight.dtype

    @property
    def device(self):
        return self.enc.weight.device

    def encode(self, x):
        h = F.avg_pool2d(x, 8)
        return SimpleNamespace(latent_dist=_Gaussian(self.enc(h)))

    def decode(self, z):
        z = self.post_quant_conv(z)
        up = F.interpolate(z, scale_factor=8, mode="nearest")
        return SimpleNamespace(sample=torch.tanh(self.dec(up)))


def synthetic_text_embeds(batch, cross_dim=32, pooled_dim=16, seed=1234, xl=False):
    """(uncond, pooled_uncond), (cond, pooled_cond) -- stands in for CLIP (elastic_diffusion.py:255-265)."""
    g = torch.Generator().manual_seed(seed)
    un = torch.randn(1, 77, cross_dim, generator=g).repeat(batch, 1, 1)
    co = torch.randn(batch, 77, cross_dim, generator=g)
    if xl:
        pun = torch.randn(1, pooled_dim, generator=g).repeat(batch, 1)
        pco = torch.randn(batch, pooled_dim, generator=g)
    else:
        pun, pco = un, co
    return (un, pun), (co, pco)
