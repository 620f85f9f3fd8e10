"""DDIM (+ DDPM.add_noise) restated from the published DDIM update (Song et al. 2020) with the diffusers 0.13.1
conventions the reference relies on: scaled-linear betas in fp32, `steps_offset`, `set_alpha_to_one`, leading
timestep spacing, `scale_model_input` = identity, `init_noise_sigma` = 1.
Reference call sites: /root/reference/model/pipeline.py:366-367,420-424,451,461; inference.py:48."""
import json
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from ..utils import BaseOutput


@dataclass
class DDIMSchedulerOutput(BaseOutput):
    prev_sample: torch.FloatTensor
    pred_original_sample: Optional[torch.FloatTensor] = None


def _betas(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas):
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise NotImplementedError(beta_schedule)


class _SchedulerBase:
    config_name = "scheduler_config.json"
    order = 1

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        with open(os.path.join(path, subfolder or "", cls.config_name)) as f:
            cfg = json.load(f)
        return cls.from_config(cfg, **kw)

    @classmethod
    def from_config(cls, cfg, **kw):
        import inspect
        params = inspect.signature(cls.__init__).parameters
        init = {k: v for k, v in dict(cfg).items() if k in params}
        init.update(kw)
        return cls(**init)

    def add_noise(self, original_samples, noise, timesteps):
        acp = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sqrt_alpha_prod = (acp[timesteps] ** 0.5).flatten()
        while len(sqrt_alpha_prod.shape) < len(original_samples.shape):
            sqrt_alpha_prod = sqrt_alpha_prod.unsqueeze(-1)
        sqrt_one_minus = ((1 - acp[timesteps]) ** 0.5).flatten()
        while len(sqrt_one_minus.shape) < len(original_samples.shape):
            sqrt_one_minus = sqrt_one_minus.unsqueeze(-1)
        return sqrt_alpha_prod * original_samples + sqrt_one_minus * noise

    def scale_model_input(self, sample, timestep=None):
        return sample


class DDIMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon"):
        self.config = type("Cfg", (), dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                           beta_end=beta_end, beta_schedule=beta_schedule, clip_sample=clip_sample,
                                           set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                           prediction_type=prediction_type))()
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device)
        self.timesteps += self.config.steps_offset

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        assert self.config.prediction_type == "epsilon"
        pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
        if self.config.clip_sample:
            pred_original_sample = torch.clamp(pred_original_sample, -1, 1)
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** 0.5
        if use_clipped_model_output:
            model_output = (sample - alpha_prod_t ** 0.5 * pred_original_sample) / beta_prod_t ** 0.5
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5 * model_output
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype,
                                             device=model_output.device)
            prev_sample = prev_sample + std_dev_t * variance_noise
        if not return_dict:
            return (prev_sample,)
        return DDIMSchedulerOutput(prev_sample=prev_sample, pred_original_sample=pred_original_sample)


class DDPMScheduler(_SchedulerBase):
    """Only `add_noise` is used by the reference training step (train_StorySalon_stage2.py:291-300)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, **_):
        self.betas = _betas(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.init_noise_sigma = 1.0


class _Unsupported(_SchedulerBase):
    def __init__(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("only DDIM/DDPM are on the oracle path")


PNDMScheduler = DPMSolverMultistepScheduler = EulerAncestralDiscreteScheduler = _Unsupported
EulerDiscreteScheduler = LMSDiscreteScheduler = _Unsupported
