"""Deterministic synthetic weights and inputs (there is no network for checkpoints or datasets).

Every tensor is drawn from its own CPU generator seeded by crc32(name) ^ seed, so the values do not depend on
module construction order, on which process asks, or on the torch device: the oracle run that produced
tests/golden/*.pt (oracle/make_golden.py, in the build container) and the HIP run on the GPU box see the
same numbers.  Scales follow PyTorch's default init (U(+-1/sqrt(fan_in))) with three deliberate changes so the
parity tests exercise what they should: norm affine parameters are perturbed away from (1, 0), and the
attention q/k projections get a gain so that softmax rows are peaked rather than uniform.
All values are rounded to fp16-representable numbers (and returned as fp32): the reference casts the UNet and its
inputs to fp16 (inference.py:73-75), so "the reference on identical weights/inputs" means these fp16 values; an fp32
evaluation of them is then the exact arithmetic of the fp16 model, and a parity error measures kernel arithmetic
only, not checkpoint quantisation.
Input recipe: SURVEY §8(d) "Synthetic inputs".
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict

import torch

from .arch import UNetArch, param_shapes

QK_GAIN = 3.0


def seed_int(name: str, seed: int) -> int:
    """The integer a tensor called `name` is seeded with (exposed so a caller that must go through the *global*
    generator — the reference pipeline draws its shared noise that way — can reproduce the same stream)."""
    return (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed_int(name, seed))
    return g


def fp16_exact(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float16).to(torch.float32)


def synthetic_tensor(name: str, shape, seed: int = 0, std: float = 1.0) -> torch.Tensor:
    return fp16_exact(torch.randn(tuple(shape), generator=_gen(name, seed), dtype=torch.float32) * std)


def synthetic_state_dict(arch: UNetArch, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """fp32 CPU state dict with the reference's keys and shapes."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes(arch).items():
        g = _gen(name, seed)
        is_norm = (".norm" in name or name.startswith("conv_norm_out")) and len(shape) == 1
        if is_norm:
            if name.endswith("weight"):
                t = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                t = 0.1 * torch.randn(shape, generator=g)
        else:
            if name.endswith("weight"):
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
            else:  # bias: fan_in of the matching weight
                wshape = param_shapes_cache(arch)[name[:-4] + "weight"]
                fan_in = 1
                for d in wshape[1:]:
                    fan_in *= d
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2.0 - 1.0) * bound
            if name.endswith("to_q.weight") or name.endswith("to_k.weight"):
                t = t * QK_GAIN
        sd[name] = fp16_exact(t)
    return sd


_shape_cache: Dict[int, "OrderedDict[str, tuple]"] = {}


def param_shapes_cache(arch: UNetArch):
    k = id(arch)
    if k not in _shape_cache:
        _shape_cache[k] = param_shapes(arch)
    return _shape_cache[k]


def synthetic_inputs(n_samples: int, n_ref: int, height: int, width: int, seed: int = 0,
                     cross_attention_dim: int = 768, seq_len: int = 77) -> Dict[str, torch.Tensor]:
    """Inputs of one `StableDiffusionPipeline.__call__` *after* its CLIP/VAE plumbing
    (/root/reference/model/pipeline.py:359-409): everything the per-step loop :412-461 consumes.

    latents        [N,4,h,w]    ~ N(0,1)         (prepare_latents, :372-381)
    image_prompts  [R,N,4,h,w]  ~ 0.8*N(0,1)     (0.18215*VAE(prior frame), :397-404)
    zero_prompt    [N,4,h,w]    ~ 0.1*N(0,1)     (0.18215*VAE(0), :390-395)
    noise          [N,4,h,w]    ~ N(0,1)         (:409, one tensor shared by every ref and step)
    text           [N,77,768], uncond [N,77,768] (_encode_prompt, :359)
    prev_text      [R,N,77,768], prev_uncond [R,N,77,768]   (:361-362)
    """
    s = seed
    f = synthetic_tensor
    return dict(
        latents=f("in.latents", (n_samples, 4, height, width), s),
        image_prompts=f("in.image_prompts", (n_ref, n_samples, 4, height, width), s, 0.8),
        zero_prompt=f("in.zero_prompt", (n_samples, 4, height, width), s, 0.1),
        # drawn un-rounded: the reference pipeline draws it from the global generator in the latents' dtype
        noise=torch.randn((n_samples, 4, height, width), generator=_gen("in.noise", s), dtype=torch.float32),
        text=f("in.text", (n_samples, seq_len, cross_attention_dim), s),
        uncond=f("in.uncond", (1, seq_len, cross_attention_dim), s).expand(n_samples, -1, -1).contiguous(),
        prev_text=f("in.prev_text", (n_ref, n_samples, seq_len, cross_attention_dim), s),
        prev_uncond=f("in.uncond", (1, seq_len, cross_attention_dim), s)
        .expand(n_ref, n_samples, -1, -1).contiguous(),
    )


def synthetic_train_batch(b: int, hw: int, cad: int, seed: int) -> Dict[str, torch.Tensor]:
    """Inputs of one stage-2 training step after its CLIP / VAE plumbing (/root/reference/train_StorySalon_stage2.py:265-302):
    latents, the 3 reference-frame latents, the two noise tensors, per-sample timesteps, text / previous-prompt embeddings and
    the 1/8-downsampled loss mask (BASELINE config 4)."""
    import torch.nn.functional as F
    f = synthetic_tensor
    g = torch.Generator().manual_seed(seed_int("train.aux", seed))
    return dict(latents=f("train.latents", (b, 4, hw, hw), seed, 0.8), ref_latents=f("train.ref_latents", (3, b, 4, hw, hw), seed, 0.8),
                noise=f("train.noise", (b, 4, hw, hw), seed), ref_noise=f("train.ref_noise", (b, 4, hw, hw), seed),
                timesteps=torch.randint(0, 1000, (b,), generator=g), text=f("train.text", (b, 77, cad), seed),
                prev_text=f("train.prev_text", (3, b, 77, cad), seed),
                # bilinear 1/8 downsample of a binary text/face mask gives values in [0, 1] (:268-270)
                mask=F.interpolate((torch.rand(b, 4, hw * 8, hw * 8, generator=g) > 0.7).float(), scale_factor=1 / 8.0, mode="bilinear",
                                   align_corners=False))
