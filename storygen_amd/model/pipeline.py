"""Drop-in for `model.pipeline.StableDiffusionPipeline` of the reference (/root/reference/model/pipeline.py:29-491).

Same constructor (vae, text_encoder, tokenizer, unet, scheduler) and `__call__` signature (:274-294).  The plumbing
on either side of the loop — prompt encoding (:87-196), VAE encoding of the prior frames (:387-404), latent
preparation (:235-271), VAE decoding (:198-205) — runs on whatever stock PyTorch modules the caller passes, exactly as
in the reference; the per-step loop body (:412-461) is the HIP sampler (storygen_amd/sampler.py): one hipGraph replay
per step.

Deliberate differences from the reference, none of which changes latents:
  * the `callback(i, t, latents)` index is the step index (the reference passes the last prior-frame index because its
    inner loop variable shadows `i`, pipeline.py:412,418 — SURVEY F6f);
  * `guidance_scale <= 1` raises (the reference's non-CFG branch passes a list where a tensor is required and cannot
    run, :430 — SURVEY F6g);
  * schedulers: DDIM (what the reference ships and uses, inference.py:48) and PNDM with `skip_prk_steps=true` (the class
    named by ckpt/stable-diffusion-v1-5/scheduler/scheduler_config.json); `scheduler` may be a
    storygen_amd.scheduler.DDIMSchedule / PNDMSchedule or any object whose `.config` carries the diffusers keys
    (`_class_name` or the object's class name selects the rule).  Anything else (Euler, LMS, DPM-solver, v-prediction,
    clip_sample) raises instead of silently running DDIM.
"""
from __future__ import annotations

from collections import namedtuple
from typing import Callable, List, Optional, Union

import torch

from ..sampler import STAGES, StoryGenSampler
from ..scheduler import DDIMSchedule, schedule_from_config
from .encoders import _HipModule

StableDiffusionPipelineOutput = namedtuple("StableDiffusionPipelineOutput", ["images", "nsfw_content_detected"])


def _as_schedule(scheduler) -> DDIMSchedule:
    if isinstance(scheduler, DDIMSchedule):
        return scheduler
    cfg = getattr(scheduler, "config", None)
    if cfg is None:
        raise TypeError("scheduler must be a DDIMSchedule / PNDMSchedule or expose a diffusers-style .config")
    return schedule_from_config(cfg, type(scheduler).__name__)


class StableDiffusionPipeline:
    """vae / text_encoder may be the HIP classes of storygen_amd.model (AutoencoderKL, CLIPTextModel) or any object with the same
    methods (the third-party torch modules the reference constructs); `hip_encoders()` swaps torch modules for the HIP classes."""

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler):
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        boc = getattr(getattr(vae, "config", None), "block_out_channels", (128, 256, 512, 512))
        self.vae_scale_factor = 2 ** (len(boc) - 1)                                       # :76
        self._sampler: Optional[StoryGenSampler] = None
        self._sampler_key = None
        self._progress_bar_config = {}

    def hip_encoders(self):
        """Replace a diffusers AutoencoderKL / transformers CLIPTextModel torch module by the HIP class holding the same weights."""
        from .encoders import AutoencoderKL, CLIPTextModel
        if isinstance(self.vae, torch.nn.Module):
            self.vae = AutoencoderKL.from_torch(self.vae).to(self._execution_device)
        if isinstance(self.text_encoder, torch.nn.Module):
            self.text_encoder = CLIPTextModel.from_torch(self.text_encoder).to(self._execution_device)
        return self

    # --------------------------------------------------------------------------------------------- plumbing
    @property
    def _execution_device(self) -> torch.device:
        return self.unet.device

    def enable_xformers_memory_efficient_attention(self, attention_op=None):   # inference.py:60
        pass

    def disable_xformers_memory_efficient_attention(self):
        pass

    def enable_vae_slicing(self):
        if hasattr(self.vae, "enable_slicing"):
            self.vae.enable_slicing()

    def disable_vae_slicing(self):
        if hasattr(self.vae, "disable_slicing"):
            self.vae.disable_slicing()

    def to(self, torch_device=None, torch_dtype=None):
        """DiffusionPipeline.to: moves every nn.Module component (inference.py:56 `pipeline.to(device)` usage pattern)."""
        for name in ("vae", "text_encoder", "unet"):
            m = getattr(self, name)
            if isinstance(m, (torch.nn.Module, _HipModule)):
                if torch_device is not None:
                    m.to(torch_device)
                if torch_dtype is not None and name != "unet":      # the HIP UNet keeps its parameters in their own dtype
                    m.to(torch_dtype)
        return self

    @property
    def device(self) -> torch.device:
        return self._execution_device

    def set_progress_bar_config(self, **kwargs):                                          # train_StorySalon_stage2.py:157
        self._progress_bar_config = dict(kwargs)

    def progress_bar(self, iterable=None, total=None):                                    # pipeline.py:411
        from tqdm.auto import tqdm
        if iterable is not None:
            return tqdm(iterable, **self._progress_bar_config)
        if total is not None:
            return tqdm(total=total, **self._progress_bar_config)
        raise ValueError("Either `total` or `iterable` has to be defined.")

    def save_pretrained(self, save_directory: str, **kwargs):
        """DiffusionPipeline.save_pretrained layout (train_StorySalon_stage2.py:348-357 saves the WHOLE pipeline): one
        sub-folder per component that can save itself + model_index.json naming each component's [library, class]."""
        import json
        import os
        os.makedirs(save_directory, exist_ok=True)
        index = {"_class_name": type(self).__name__, "_diffusers_version": "0.13.1"}
        for name in ("vae", "text_encoder", "tokenizer", "unet", "scheduler"):
            comp = getattr(self, name)
            if comp is None:
                index[name] = [None, None]
                continue
            lib = type(comp).__module__.split(".")[0]
            index[name] = [lib, type(comp).__name__]
            sub = os.path.join(save_directory, name)
            if name == "unet":
                comp.save_pretrained(sub, **kwargs)
            elif hasattr(comp, "save_pretrained"):
                comp.save_pretrained(sub)
            elif hasattr(comp, "save_config"):
                comp.save_config(sub)
        with open(os.path.join(save_directory, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)

    def check_inputs(self, prompt, height, width, callback_steps):                        # :223-233
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")

    def _embed(self, prompts: List[str], device, max_length=None) -> torch.Tensor:
        tok = self.tokenizer(prompts, padding="max_length", max_length=max_length or self.tokenizer.model_max_length,
                             truncation=True, return_tensors="pt")
        cfg = getattr(self.text_encoder, "config", None)
        mask = tok.attention_mask.to(device) if getattr(cfg, "use_attention_mask", False) else None
        return self.text_encoder(tok.input_ids.to(device), attention_mask=mask)[0]

    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt):
        """[uncond | text] embeddings, each repeated per image (:87-196)."""
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        text = self._embed(prompts, device).repeat_interleave(num_images_per_prompt, dim=0)
        if not do_classifier_free_guidance:
            return text
        if negative_prompt is None:
            neg = [""] * len(prompts)
        elif type(prompt) is not type(negative_prompt):
            raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {type(prompt)}.")
        elif isinstance(negative_prompt, str):
            neg = [negative_prompt]
        elif len(negative_prompt) != len(prompts):
            raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`: {prompt} "
                             f"has batch size {len(prompts)}. Please make sure that passed `negative_prompt` matches the batch size of `prompt`.")
        else:
            neg = list(negative_prompt)
        uncond = self._embed(neg, device, max_length=text.shape[1]).repeat_interleave(num_images_per_prompt, dim=0)
        return torch.cat([uncond, text])

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        shape = (batch_size, num_channels_latents, height // self.vae_scale_factor, width // self.vae_scale_factor)   # :235-271
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, device=g.device, dtype=dtype).to(device)
                                     for g in generator])
            else:
                gdev = generator.device if generator is not None else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents

    def decode_latents(self, latents):                                                   # :198-205
        image = self.vae.decode(latents / 0.18215).sample
        image = (image / 2 + 0.5).clamp(0, 1)
        return image.cpu().permute(0, 2, 3, 1).float().numpy()

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        if images.ndim == 3:
            images = images[None]
        return [Image.fromarray((im * 255).round().astype("uint8")) for im in images]

    # --------------------------------------------------------------------------------------------- the call
    @torch.no_grad()
    def __call__(self, stage: str, prompt: Union[str, List[str]], image_prompt: Optional[torch.Tensor] = None,
                 prev_prompt: Optional[List[Union[str, List[str]]]] = None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, guidance_scale: float = 7.5,
                 image_guidance_scale: float = 3.5, negative_prompt: Optional[Union[str, List[str]]] = None,
                 num_images_per_prompt: Optional[int] = 1, eta: float = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 latents: Optional[torch.Tensor] = None, output_type: Optional[str] = "pil", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: Optional[int] = 1):
        height = height or self.unet.config.sample_size * self.vae_scale_factor           # :347-348
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        if stage not in STAGES:
            stage = "no"                          # the reference treats every other string like 'no' (:425-427,436-438,444-445)
        if eta != 0.0:
            raise NotImplementedError("eta != 0 (stochastic DDIM) is not on the StoryGen path")
        batch_size = 1 if isinstance(prompt, str) else len(prompt)
        device = self._execution_device
        if not guidance_scale > 1.0:
            raise ValueError("guidance_scale must be > 1: the reference loop only works with classifier-free guidance")
        n = batch_size * num_images_per_prompt
        emb = self._encode_prompt(prompt, device, num_images_per_prompt, True, negative_prompt)       # :359
        uncond, text = emb[:n], emb[n:]
        prev = [self._encode_prompt(p, device, num_images_per_prompt, True, negative_prompt) for p in prev_prompt]   # :361-362
        dtype = text.dtype
        latents = self.prepare_latents(n, self.unet.in_channels, height, width, dtype, device, generator, latents)   # :372-381
        image_prompt = image_prompt.to(device=device, dtype=dtype)                        # [B, R, 3, H, W]  :387
        frames = image_prompt.transpose(0, 1)
        R = frames.shape[0]
        if len(prev) != R:
            raise ValueError(f"{R} prior frames but {len(prev)} previous prompts")
        zero = self.vae.encode(frames[0] * 0).latent_dist.sample() * 0.18215              # :390-393
        zero = zero.repeat(num_images_per_prompt, 1, 1, 1)
        imgs = torch.stack([(self.vae.encode(f).latent_dist.sample() * 0.18215).repeat(num_images_per_prompt, 1, 1, 1)
                            for f in frames])                                             # :397-404
        noise = torch.randn_like(imgs[0])                                                 # :409 (global generator)
        inputs = dict(latents=latents, image_prompts=imgs, zero_prompt=zero, noise=noise, text=text, uncond=uncond,
                      prev_text=torch.stack([p[n:] for p in prev]), prev_uncond=torch.stack([p[:n] for p in prev]))
        h, w = latents.shape[-2:]
        # the repacked weights are refreshed IN PLACE when the UNet's parameters change (optimizer step, load_state_dict),
        # so a cached sampler and its captured hipGraphs stay valid; only a new weights object (device change) rebuilds it
        wts = self.unet._engine_weights()
        schedule = _as_schedule(self.scheduler)
        # group schedule (sampler ref_ahead): the reference passes of 5 consecutive steps as one batched UNet call inside one hipGraph per
        # group — when nobody watches the intermediate latents (a callback sees every step: step-by-step graphs then) and the number of
        # UNet evaluations is a multiple of 5 (DDIM: 50 steps -> 50; PNDM: n + 1)
        evals = len(schedule.timesteps(num_inference_steps))
        G = 5 if (callback is None and stage in STAGES[:2] and evals % 5 == 0) else 1
        key = (n, h, w, R, text.shape[1], id(wts), schedule.key(), G)
        if self._sampler is None or self._sampler_key != key:
            self._sampler = StoryGenSampler(self.unet._arch, None, device, n, h, w, R, text.shape[1], schedule=schedule, weights=wts,
                                            ref_ahead=G)
            self._sampler_key = key
        smp = self._sampler
        smp.prepare(inputs, num_inference_steps, stage, guidance_scale, image_guidance_scale)
        with self.progress_bar(total=len(smp.timesteps)) as bar:
            for i, t in enumerate(smp.timesteps):                                         # :411-469
                smp.step(i)
                bar.update()
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, smp.latents.to(dtype, copy=True))                      # never a view of the sampler's buffer
        latents = smp.latents.to(dtype, copy=True)
        smp.check_guards()                         # a folded LayerNorm outside its range raises here instead of returning wrong latents
        if output_type == "latent":
            image = latents
        else:
            image = self.decode_latents(latents)                                          # :472
            if output_type == "pil":
                image = self.numpy_to_pil(image)
        if not return_dict:
            return (image, None)
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=None)
