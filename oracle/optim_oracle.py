"""TEST INFRASTRUCTURE — CPU restatement (torch fp32) of the optimizer side of the reference's training loop
(/root/reference/train_StorySalon_stage2.py:186-205 optimizer construction, :214-219 `get_scheduler`, :328-333 clip + step).
Only tests/ may import this module; the product (storygen_amd/optim.py) never does.

  * `adamw_step`       torch.optim.AdamW's single-tensor update — PINNED: tests/test_optim_host.py checks it against
                       torch.optim.AdamW itself (the class the reference instantiates when use_8bit_adam is false).
  * `clip_coef`        torch.nn.utils.clip_grad_norm_ (accelerate's clip_grad_norm_ after unscaling) — pinned the same way.
  * `lr_lambda`        diffusers 0.13.1 optimization.get_scheduler's multiplier for the schedule names it offers — diffusers is absent,
                       but its optimization.py restates transformers.optimization, which is installed: PINNED against transformers'
                       own schedule functions for all six names (tests/test_optim_host.py).  The reference's configs use "constant"
                       (config/*.yml `lr_scheduler: constant`).
  * `dynamic_map`, `adamw8bit_step`   block-wise 8-bit AdamW as published for bitsandbytes==0.35.4 (environment.yaml; CUDA-only, not
                       installable here): dynamic-tree code books (signed for the first moment, unsigned for the second), 2048-element
                       blocks with per-block absmax, states de-quantised, updated in fp32 and re-quantised to the nearest code —
                       PARITY UNPINNED against bitsandbytes; anchored by the property test that it tracks fp32 AdamW.
"""
import math
from typing import List, Optional, Tuple

import torch

BLOCK = 2048


def adamw_step(p, g, m, v, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
    """In place on p, m, v (torch/optim/adamw.py::_single_tensor_adamw, amsgrad=False, maximize=False)."""
    b1, b2 = betas
    p.mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def clip_coef(grads: List[torch.Tensor], max_norm: float) -> float:
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float(), 2) for g in grads]), 2)
    return float(torch.clamp(max_norm / (total + 1e-6), max=1.0))


def lr_lambda(name: str, warmup: int = 0, total: Optional[int] = None, cycles: float = 0.5, power: float = 1.0,
              lr_init: float = 1.0, lr_end: float = 1e-7):
    """step -> multiplier of the base learning rate (diffusers optimization.py, SchedulerType values)."""
    def warm(s):
        return float(s) / float(max(1, warmup))
    if name == "constant":
        return lambda s: 1.0
    if name == "constant_with_warmup":
        return lambda s: warm(s) if s < warmup else 1.0
    if name == "linear":
        return lambda s: warm(s) if s < warmup else max(0.0, float(total - s) / float(max(1, total - warmup)))
    if name == "cosine":
        return lambda s: warm(s) if s < warmup else max(
            0.0, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * float(s - warmup) / float(max(1, total - warmup)))))
    if name == "cosine_with_restarts":
        def f(s):
            if s < warmup:
                return warm(s)
            prog = float(s - warmup) / float(max(1, total - warmup))
            return 0.0 if prog >= 1.0 else max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(cycles) * prog) % 1.0))))
        return f
    if name == "polynomial":
        def g(s):
            if s < warmup:
                return warm(s)
            if s > total:
                return lr_end / lr_init
            return ((lr_init - lr_end) * (1.0 - (s - warmup) / (total - warmup)) ** power + lr_end) / lr_init
        return g
    raise ValueError(f"unknown lr scheduler {name!r}")


def dynamic_map(signed: bool = True, n: int = 7) -> torch.Tensor:
    """bitsandbytes functional.create_dynamic_map: 256 ascending values in [-1, 1] (signed) or [0, 1] (unsigned) — per decade
    10^(i-n+1) a linear grid of 2^i (signed; 2^(i+1) unsigned) bin centres between 0.1 and 1, plus 0 and 1."""
    data: List[float] = []
    for i in range(n):
        items = 2 ** i + 1 if signed else 2 ** (i + 1) + 1
        b = torch.linspace(0.1, 1, items)
        means = ((b[:-1] + b[1:]) / 2.0).tolist()
        data += [(10 ** (-(n - 1) + i)) * x for x in means]
        if signed:
            data += [-(10 ** (-(n - 1) + i)) * x for x in means]
    data += [0.0, 1.0]
    assert len(data) == 256
    return torch.tensor(sorted(data), dtype=torch.float32)


def nearest_code(code: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Index of the nearest entry of the ascending code book (ties to the lower index)."""
    hi = torch.bucketize(x, code, right=True).clamp(1, 255)           # code[hi-1] <= x < code[hi]
    lo = hi - 1
    pick_lo = (x - code[lo]) <= (code[hi] - x)
    idx = torch.where(pick_lo, lo, hi)
    idx = torch.where(x <= code[0], torch.zeros_like(idx), idx)
    return torch.where(x >= code[255], torch.full_like(idx, 255), idx).to(torch.uint8)


def adamw8bit_state(n: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Zero states: codes pointing at the 0.0 entry of each book, absmax 0."""
    nb = (n + BLOCK - 1) // BLOCK
    z1 = int((dynamic_map(True) == 0).nonzero()[0])
    z2 = int((dynamic_map(False) == 0).nonzero()[0])
    return (torch.full((n,), z1, dtype=torch.uint8), torch.full((n,), z2, dtype=torch.uint8), torch.zeros(nb), torch.zeros(nb))


def adamw8bit_step(p, g, c1, c2, a1, a2, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
    """In place on p (flat fp32), c1/c2 (uint8 codes), a1/a2 (per-block absmax)."""
    b1, b2 = betas
    code1, code2 = dynamic_map(True), dynamic_map(False)
    n = p.numel()
    nb = a1.numel()
    blk = torch.arange(n) // BLOCK
    m = code1[c1.long()] * a1[blk]
    v = code2[c2.long()] * a2[blk]
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, sbc2 = 1 - b1 ** step, math.sqrt(1 - b2 ** step)
    p.sub_(lr * weight_decay * p)
    p.sub_((lr * sbc2 / bc1) * (m / (v.sqrt() + eps * sbc2)))
    pad = nb * BLOCK - n
    mx1 = torch.cat([m.abs(), torch.zeros(pad)]).view(nb, BLOCK).amax(1)
    mx2 = torch.cat([v, torch.zeros(pad)]).view(nb, BLOCK).amax(1)
    a1.copy_(mx1), a2.copy_(mx2)
    r1 = torch.where(mx1 > 0, 1.0 / mx1, torch.zeros_like(mx1))[blk]
    r2 = torch.where(mx2 > 0, 1.0 / mx2, torch.zeros_like(mx2))[blk]
    c1.copy_(nearest_code(code1, m * r1)), c2.copy_(nearest_code(code2, v * r2))
