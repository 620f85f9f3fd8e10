"""DDIM schedule tables for the sampling loop (host side, fp32, computed exactly as diffusers' DDIMScheduler does
for the reference's scheduler_config.json: scaled-linear betas 0.00085..0.012, 1000 train steps, steps_offset=1,
set_alpha_to_one=False, clip_sample=False, eta=0).  Reference call sites: model/pipeline.py:366-367 (set_timesteps),
:420-424 (add_noise), :461 (step); config: ckpt/stable-diffusion-v1-5/scheduler/scheduler_config.json:1-13."""
from __future__ import annotations

import json
import os
from typing import List, Optional

import torch


class DDIMSchedule:
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", steps_offset: int = 1, set_alpha_to_one: bool = False,
                 clip_sample: bool = False, trained_betas=None, **_ignored):
        if beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for DDIMSchedule")
        if clip_sample:
            raise NotImplementedError("clip_sample=True is not on the StoryGen path")
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0

    @classmethod
    def from_pretrained(cls, path: str, subfolder: Optional[str] = "scheduler") -> "DDIMSchedule":
        with open(os.path.join(path, subfolder or "", "scheduler_config.json")) as f:
            return cls(**{k: v for k, v in json.load(f).items() if not k.startswith("_")})

    def timesteps(self, n: int) -> List[int]:
        ratio = self.num_train_timesteps // n
        return [int(round(i * ratio)) + self.steps_offset for i in reversed(range(n))]

    def add_noise_coef(self, t: int):
        """(sqrt(abar_t), sqrt(1 - abar_t)) as fp32 python floats."""
        a = self.alphas_cumprod[t]
        return float(a ** 0.5), float((1 - a) ** 0.5)

    def step_coef(self, t: int, n: int):
        """(sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev)) for x_t -> x_{t - T/n}, eta = 0."""
        prev = t - self.num_train_timesteps // n
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5)
