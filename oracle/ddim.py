"""ORACLE (test infrastructure, never shipped in the product path).

CPU restatement of the third-party scheduler the reference drives:
``diffusers==0.21.4`` ``DDIMScheduler`` (pinned by /root/reference/environment.yaml:21).
``diffusers`` is NOT vendored under /root/reference and is not installable in this image, so this file
restates its published eta=0 DDIM algorithm from the call sites the reference uses:

  * ``DDIMScheduler.from_pretrained(model_key, subfolder="scheduler")``   elastic_diffusion.py:153
  * ``scheduler.set_timesteps(num_inference_steps)``                        elastic_diffusion.py:1001
  * ``scheduler.scale_model_input(latent, t)`` (identity for DDIM)          elastic_diffusion.py:402
  * ``scheduler.step(noise_pred, t, latent)`` -> prev_sample / pred_original_sample
                                                                            elastic_diffusion.py:776, 920, 1033, 1054
  * ``scheduler.add_noise(bg, noise, t)``                                   elastic_diffusion.py:358
  * ``scheduler.betas[t]``, ``scheduler.config.num_train_timesteps``,
    ``scheduler.num_inference_steps``                                       elastic_diffusion.py:693-699

PARITY NOTE: "parity unpinned" for this file -- the reference holds no tests / golden vectors for the
scheduler and the dependency is absent, so the arithmetic below is pinned only to diffusers' published
formulas (DDIM paper eq. 12, eta = 0) and to the SD / SDXL scheduler_config.json values.  The *glue* parity
(everything under /root/reference) is pinned with golden vectors produced by driving the real reference
code with THIS scheduler injected (tests/golden/make_golden.py).

All arithmetic is fp32 torch-CPU, in the operation order diffusers uses (0-d fp32 coefficient tensors
combined with the sample by separate mul / sub / div / add), because the HIP kernels are tested
bit-for-bit against it.
"""
from types import SimpleNamespace

import numpy as np
import torch

# scheduler_config.json values shared by runwayml/stable-diffusion-v1-5, stabilityai/stable-diffusion-2(-1)-base
# and stabilityai/stable-diffusion-xl-base-1.0 (the keys DDIMScheduler consumes).
SD_SCHEDULER_CONFIG = dict(
    num_train_timesteps=1000,
    beta_start=0.00085,
    beta_end=0.012,
    beta_schedule="scaled_linear",
    clip_sample=False,
    set_alpha_to_one=False,
    steps_offset=1,
    prediction_type="epsilon",
    timestep_spacing="leading",
)


class StepOutput(dict):
    """dict with attribute access; the reference indexes ``ddim_out['prev_sample']``."""

    __getattr__ = dict.__getitem__


class DDIMOracle:
    def __init__(self, **overrides):
        cfg = dict(SD_SCHEDULER_CONFIG)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        n = cfg["num_train_timesteps"]
        if cfg["beta_schedule"] == "scaled_linear":
            self.betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
        elif cfg["beta_schedule"] == "linear":
            self.betas = torch.linspace(cfg["beta_start"], cfg["beta_end"], n, dtype=torch.float32)
        else:
            raise NotImplementedError(cfg["beta_schedule"])
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, n)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing != "leading":
            raise NotImplementedError(self.config.timestep_spacing)
        ratio = n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, **_):
        if self.num_inference_steps is None:
            raise ValueError("set_timesteps first")
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        if self.config.prediction_type == "epsilon":
            pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
            pred_epsilon = model_output
        elif self.config.prediction_type == "v_prediction":
            pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
            pred_epsilon = (alpha_prod_t ** 0.5) * model_output + (beta_prod_t ** 0.5) * sample
        else:
            raise NotImplementedError(self.config.prediction_type)
        if self.config.clip_sample:
            pred_original_sample = pred_original_sample.clamp(-1.0, 1.0)
        assert eta == 0.0, "the reference never passes eta"
        std_dev_t = 0.0
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        return StepOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(dtype=original_samples.dtype)
        timesteps = timesteps.to(torch.long)
        sa = (ac[timesteps] ** 0.5).flatten()
        so = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.dim() < original_samples.dim():
            sa = sa.unsqueeze(-1)
            so = so.unsqueeze(-1)
        return sa * original_samples + so * noise
