"""Host-side schedules: the DDIM coefficient tables the HIP kernels consume and the RRG weight schedules.

``DDIMSchedule`` plays the role of the ``diffusers`` ``DDIMScheduler`` the reference loads at
/root/reference/elastic_diffusion.py:153 (diffusers==0.21.4 is not vendored; its eta=0 algorithm is restated).
Only *scalars* are produced here -- every per-element operation of ``scheduler.step`` / ``add_noise`` /
``undo_step`` runs in libelastic_hip.so.  Scalars are computed with fp32 torch ops in the same order diffusers uses
(``alphas_cumprod[t] ** 0.5`` etc.) so that the kernels reproduce the reference's torch-CPU results bit for bit.
"""
from types import SimpleNamespace

import numpy as np
import torch


class DDIMSchedule:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 set_alpha_to_one=False, steps_offset=1, prediction_type="epsilon", timestep_spacing="leading",
                 clip_sample=False):
        if prediction_type != "epsilon":
            raise NotImplementedError("the HIP DDIM kernel implements epsilon prediction (SD1.x/2.x-base/SDXL)")
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not used by any SD / SDXL scheduler config")
        if timestep_spacing != "leading":
            raise NotImplementedError(timestep_spacing)
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, set_alpha_to_one=set_alpha_to_one,
                                      steps_offset=steps_offset, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing, clip_sample=clip_sample)
        if beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    @classmethod
    def from_config_dir(cls, model_dir):
        """``DDIMScheduler.from_pretrained(model_key, subfolder="scheduler")`` (ED:153) for a local HF snapshot: reads
        ``<model_dir>/scheduler/scheduler_config.json`` when present (keys this class does not know are ignored, values
        it cannot honour -- v-prediction, trailing spacing -- raise NotImplementedError), defaults otherwise."""
        import json
        import os
        path = os.path.join(model_dir, "scheduler", "scheduler_config.json")
        if not os.path.isfile(path):
            return cls()
        cfg = json.load(open(path))
        known = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "set_alpha_to_one", "steps_offset",
                 "prediction_type", "timestep_spacing", "clip_sample")
        return cls(**{k: cfg[k] for k in known if k in cfg})

    def set_timesteps(self, num_inference_steps):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError(f"num_inference_steps {num_inference_steps} > num_train_timesteps {n}")
        self.num_inference_steps = num_inference_steps
        ratio = n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts + self.config.steps_offset)  # int64: str(t) == "tensor(981)"
        return self.timesteps

    # ---- scalar tables ---------------------------------------------------------------------------
    def step_coefficients(self, t):
        """(sqrt(1-abar_t), sqrt(abar_t), sqrt(abar_prev), sqrt(1-abar_prev)) as python floats holding fp32 values."""
        t = int(t)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        sqrt_beta_t = (1 - a_t) ** 0.5
        sqrt_alpha_t = a_t ** 0.5
        sqrt_alpha_prev = a_prev ** 0.5
        sqrt_1m_alpha_prev = (1 - a_prev - 0.0) ** 0.5
        return tuple(float(v) for v in (sqrt_beta_t, sqrt_alpha_t, sqrt_alpha_prev, sqrt_1m_alpha_prev))

    def add_noise_coefficients(self, t):
        a = self.alphas_cumprod[int(t)]
        return float(a ** 0.5), float((1 - a) ** 0.5)

    def undo_coefficients(self, t_next):
        """[(sqrt(1-beta_{t'+k}), sqrt(beta_{t'+k}))] for the n_train // n_steps forward sub-steps of undo_step
        (elastic_diffusion.py:692-704)."""
        n_sub = self.config.num_train_timesteps // self.num_inference_steps
        b = self.betas[int(t_next): int(t_next) + n_sub]
        if len(b) != n_sub:
            raise IndexError("undo_step would index betas past num_train_timesteps (the reference raises too)")
        return torch.stack([(1 - b) ** 0.5, b ** 0.5], dim=1).contiguous()  # fp32 [n_sub, 2]


class CosineScheduler:
    """RRG weight: factor * (0.5 (1 + cos(pi i / steps))) ** cosine_scale, zero from ``steps`` on
    (elastic_diffusion.py:96-107)."""

    def __init__(self, steps, cosine_scale, factor=0.01):
        self.steps = steps
        self.cosine_scale = cosine_scale
        self.factor = factor

    def __call__(self, t, *args, **kwargs):
        if t >= self.steps:
            return 0
        return self.factor * ((0.5 * (1 + np.cos(np.pi * t / self.steps))) ** self.cosine_scale)


class LinearScheduler:
    """elastic_diffusion.py:73-82"""

    def __init__(self, steps, start_val, stop_val):
        self.steps = steps
        self.start_val = start_val
        self.stop_val = stop_val

    def __call__(self, t, *args, **kwargs):
        if t >= self.steps:
            return self.stop_val
        return self.start_val + (self.stop_val - self.start_val) / self.steps * t


class ConstScheduler:
    """elastic_diffusion.py:85-94"""

    def __init__(self, steps, start_val, stop_val):
        self.steps = steps
        self.start_val = start_val
        self.stop_val = stop_val

    def __call__(self, t, *args, **kwargs):
        return self.stop_val if t >= self.steps else self.start_val
