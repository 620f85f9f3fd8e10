"""Schedule tables for the sampling loop (host side, fp32).

DDIMSchedule: computed exactly as diffusers' DDIMScheduler does for the scheduler the reference's inference.py builds
(inference.py:48: scaled-linear betas 0.00085..0.012, 1000 train steps, steps_offset=1, set_alpha_to_one=False,
clip_sample=False, eta=0).  PNDMSchedule: the class the shipped ckpt/stable-diffusion-v1-5/scheduler/scheduler_config.json
names (`_class_name: PNDMScheduler`, skip_prk_steps=true) — the PLMS linear-multistep rule of diffusers 0.13.1
(`PNDMScheduler.step_plms` / `_get_prev_sample`), restated from the published algorithm (diffusers is not installed here).
Reference call sites: model/pipeline.py:7-16,47-75 (accepted scheduler classes), :366-367 (set_timesteps), :420-424
(add_noise), :461 (step)."""
from __future__ import annotations

import json
import os
from typing import List, Optional

import torch


# keys of a compatible scheduler's config that do not change the DDIM / PLMS arithmetic (diffusers ignores them the same way
# when inference.py:48 loads the shipped PNDM scheduler_config.json into a DDIMScheduler)
_IGNORED_KEYS = ("_class_name", "_diffusers_version", "_name_or_path", "_use_default_values", "skip_prk_steps")


class DDIMSchedule:
    kind = "ddim"          # which update kernel the sampler launches (ddim: sg_cfg_ddim_step_f32, plms: sg_cfg_plms_step_f32)
    row_len = 4            # floats step_row() contributes to a row of the sampler's per-step table

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", steps_offset: int = 1, set_alpha_to_one: bool = False,
                 clip_sample: bool = False, trained_betas=None, prediction_type: str = "epsilon", **unknown):
        unknown = {k: v for k, v in unknown.items() if k not in _IGNORED_KEYS}
        if unknown:
            raise NotImplementedError(f"{type(self).__name__}: unsupported scheduler config keys {sorted(unknown)}")
        if prediction_type != "epsilon":
            raise NotImplementedError(f"prediction_type={prediction_type!r}: only epsilon prediction is on the StoryGen path")
        self._config = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                            beta_schedule=beta_schedule, steps_offset=steps_offset, set_alpha_to_one=set_alpha_to_one,
                            trained_betas=None if trained_betas is None else [float(b) for b in trained_betas],
                            prediction_type=prediction_type,
                            # always False here (True raises below) and written out explicitly: diffusers' DDIMScheduler DEFAULTS
                            # clip_sample to True, so a saved config without the key would make the reference's inference.py:48
                            # (`DDIMScheduler.from_pretrained(ckpt, subfolder="scheduler")`) clip x0 on a checkpoint written here
                            clip_sample=False)
        if trained_betas is not None:
            betas = torch.tensor([float(b) for b in trained_betas], dtype=torch.float32)
            if betas.numel() != num_train_timesteps:
                raise ValueError(f"trained_betas has {betas.numel()} entries for {num_train_timesteps} train timesteps")
        elif beta_schedule == "scaled_linear":
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

    @property
    def config(self) -> dict:
        return dict(self._config, _class_name=self.diffusers_name)

    diffusers_name = "DDIMScheduler"

    def save_pretrained(self, save_directory: str):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "scheduler_config.json"), "w") as f:
            json.dump(dict(self.config, _diffusers_version="0.13.1"), f, indent=2)

    save_config = save_pretrained

    def key(self) -> tuple:
        """Hashable identity of the schedule (cache key of the pipeline's sampler)."""
        c = self._config
        return (type(self).__name__,) + tuple((k, tuple(v) if isinstance(v, list) else v) for k, v in sorted(c.items()))

    def step_row(self, k: int, ts: List[int], n: int) -> List[float]:
        """The scalars the update kernel needs at loop index k (timesteps `ts` = self.timesteps(n))."""
        return list(self.step_coef(int(ts[k]), n))

    def timesteps(self, n: int) -> List[int]:
        ratio = self.num_train_timesteps // n
        return [int(round(i * ratio)) + self.steps_offset for i in reversed(range(n))]

    def add_noise_coef(self, t: int):
        """(sqrt(abar_t), sqrt(1 - abar_t)) as fp32 python floats (memoised per timestep: a sampler's step table asks for the same few
        hundred values a thousand times per prepare(), ~10 us of tensor indexing each)."""
        memo = self.__dict__.setdefault("_add_noise_memo", {})
        t = int(t)
        v = memo.get(t)
        if v is None:
            a = self.alphas_cumprod[t]
            v = memo[t] = (float(a ** 0.5), float((1 - a) ** 0.5))
        return v

    def step_coef(self, t: int, n: int):
        """(sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev)) for x_t -> x_{t - T/n}, eta = 0."""
        memo = self.__dict__.setdefault("_step_coef_memo", {})
        v = memo.get((int(t), int(n)))
        if v is None:
            prev = t - self.num_train_timesteps // n
            a_t = self.alphas_cumprod[t]
            a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
            v = memo[(int(t), int(n))] = (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5))
        return v


class PNDMSchedule(DDIMSchedule):
    """diffusers 0.13.1 PNDMScheduler with skip_prk_steps=True (the SD-1.5 default): pseudo linear multistep (PLMS).

    n inference steps make n + 1 UNet evaluations: timesteps = [s_{n-1}, s_{n-2}, s_{n-2}, s_{n-3}, ..., s_0]
    (`set_timesteps`).  With e_c the (guided) epsilon of call c and `ets` the history of calls c != 1:
        c = 0 : e' = e_0,                               x <- prev(x, t, t - r, e'),  the incoming sample is kept
        c = 1 : e' = (e_1 + e_0) / 2,                   x <- prev(kept sample, t + r, t, e')      (e_1 is not stored)
        c = 2 : e' = (3 e_2 - e_0) / 2
        c = 3 : e' = (23 e_3 - 16 e_2 + 5 e_0) / 12
        c >= 4: e' = (55 e_c - 59 ets[-2] + 37 ets[-3] - 9 ets[-4]) / 24
    prev(x, t, p, e) = sqrt(a_p / a_t) x - (a_p - a_t) e / (a_t sqrt(1 - a_p) + sqrt(a_t (1 - a_t) a_p))   (`_get_prev_sample`).
    step_row() encodes one call as [A, Bc, w0..w3, slot_cur, slot1..slot3, push, use_kept, keep]: the update kernel computes
    e' = w0 e + sum_i w_i hist[slot_i], optionally stores e in hist[slot_cur] first (a 4-deep ring), and applies
    x <- A x_src - Bc e' with x_src = the kept sample (use_kept) or the current latents."""
    kind = "plms"
    row_len = 13
    diffusers_name = "PNDMScheduler"

    def __init__(self, skip_prk_steps: bool = False, **kw):
        if not skip_prk_steps:
            raise NotImplementedError("PNDM with Runge-Kutta warm-up steps (skip_prk_steps=false) is not on the StoryGen path; the "
                                      "shipped scheduler_config.json sets skip_prk_steps=true")
        kw.pop("clip_sample", None)
        super().__init__(**kw)
        self._config["skip_prk_steps"] = True
        self._config.pop("set_alpha_to_one", None)
        self._config.pop("clip_sample", None)            # not a PNDMScheduler key
        self._config["set_alpha_to_one"] = kw.get("set_alpha_to_one", False)

    def timesteps(self, n: int) -> List[int]:
        ratio = self.num_train_timesteps // n
        base = [int(round(i * ratio)) + self.steps_offset for i in range(n)]
        return (base[:-1] + base[-2:-1] + base[-1:])[::-1]

    def _prev_coef(self, t: int, prev: int):
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        A = (a_p / a_t) ** 0.5
        denom = a_t * (1.0 - a_p) ** 0.5 + (a_t * (1.0 - a_t) * a_p) ** 0.5
        return A, (a_p - a_t) / denom

    def step_row(self, k: int, ts: List[int], n: int) -> List[float]:
        ratio = self.num_train_timesteps // n
        t = int(ts[k])
        if k == 1:
            A, Bc = self._prev_coef(t + ratio, t)
        else:
            A, Bc = self._prev_coef(t, t - ratio)
        pushes = k if k < 2 else k - 1             # history entries stored BEFORE this call (call 1 stores nothing)
        push = 0 if k == 1 else 1
        cur = pushes % 4                           # ring slot this call's epsilon goes to (if pushed)
        n_hist = pushes + push                     # len(ets) after the append
        back = lambda j: (cur - j) % 4 if push else (pushes - j) % 4      # noqa: E731  slot of ets[-1-j] (push) / ets[-j] (no push)
        if k == 0:
            w, sl = [1.0, 0.0, 0.0, 0.0], [0, 0, 0]
        elif k == 1:
            w, sl = [0.5, 0.5, 0.0, 0.0], [back(1), 0, 0]
        elif n_hist == 2:
            w, sl = [1.5, -0.5, 0.0, 0.0], [back(1), 0, 0]
        elif n_hist == 3:
            w, sl = [23.0 / 12.0, -16.0 / 12.0, 5.0 / 12.0, 0.0], [back(1), back(2), 0]
        else:
            w, sl = [55.0 / 24.0, -59.0 / 24.0, 37.0 / 24.0, -9.0 / 24.0], [back(1), back(2), back(3)]
        return [A, Bc, *w, float(cur), *map(float, sl), float(push), float(k == 1), float(k == 0)]


def schedule_from_config(cfg, class_name: str = "") -> DDIMSchedule:
    """A diffusers scheduler `.config` (dict, FrozenDict or attribute object) -> DDIMSchedule / PNDMSchedule; raises
    NotImplementedError for every other scheduler class instead of silently running DDIM."""
    def get(k, default=None):
        if isinstance(cfg, dict):
            return cfg.get(k, default)
        return getattr(cfg, k, default)
    name = get("_class_name") or class_name
    keys = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "steps_offset", "set_alpha_to_one", "clip_sample",
            "trained_betas", "prediction_type", "skip_prk_steps")
    kw = {k: get(k) for k in keys if get(k) is not None}
    if "PNDM" in name:
        kw.pop("clip_sample", None)
        return PNDMSchedule(**kw)
    if "DDIM" in name or name in ("", "DDIMSchedule"):
        kw.pop("skip_prk_steps", None)
        return DDIMSchedule(**kw)
    raise NotImplementedError(f"scheduler {name!r}: the HIP loop implements DDIM (eta = 0) and PNDM/PLMS (skip_prk_steps)")
