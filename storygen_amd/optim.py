"""Optimizer side of the reference's training loop on the HIP kernels of csrc/optim.hip:

  AdamW / AdamW8bit   torch.optim.AdamW / bitsandbytes.optim.AdamW8bit   /root/reference/train_StorySalon_stage2.py:186-205
  clip_grad_norm_     accelerator.clip_grad_norm_                        :329-330
  get_scheduler       diffusers.optimization.get_scheduler               :214-219 (config/*.yml: "constant")

Same constructor arguments and call order as the reference's loop (`clip_grad_norm_` -> `step()` -> `lr_scheduler.step()` ->
`zero_grad()`).  Parameters are fp32 CUDA tensors updated IN PLACE through their device pointers (their autograd version counter is
bumped, so the drop-in UNet's weight-staleness tag sees the change); gradients come from `p.grad` or from the {name: tensor} dict
`UNetTrainer.train_step*` returns (`set_grads`).  A whole step is a handful of launches per tensor and no host read: the global-norm
clipping coefficient is computed on the device.  bitsandbytes is CUDA-only, so AdamW8bit restates its published block-wise algorithm
(2048-element blocks, dynamic-tree code books, per-block absmax); tensors below `min_8bit_size` elements keep fp32 states, as there.
No CPU path: importing this module loads libstorygen_hip.so."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, List, Optional, Union

import torch

from . import _lib
from ._lib import AdamWDesc, check

lib = _lib.load()
BLOCK = 2048


def create_dynamic_map(signed: bool = True, n: int = 7) -> torch.Tensor:
    """The 256-entry dynamic-tree code book of 8-bit optimizers (Dettmers et al. 2021, as bitsandbytes builds it): for each of the n
    decades 10^(i-n+1) the centres of 2^i (signed) or 2^(i+1) (unsigned) equal bins between 0.1 and 1, mirrored when signed, plus 0, 1."""
    vals: List[float] = []
    for i in range(n):
        edges = torch.linspace(0.1, 1, (2 ** i if signed else 2 ** (i + 1)) + 1)
        centres = ((edges[:-1] + edges[1:]) / 2.0).tolist()
        scale = 10.0 ** (i - n + 1)
        vals += [scale * c for c in centres]
        if signed:
            vals += [-scale * c for c in centres]
    vals += [0.0, 1.0]
    if len(vals) != 256:
        raise AssertionError("dynamic map must have 256 entries")
    return torch.tensor(sorted(vals), dtype=torch.float32)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class AdamW:
    """torch.optim.AdamW(params, lr, betas, eps, weight_decay) on sg_adamw_f32.  `params`: an iterable of fp32 CUDA tensors /
    nn.Parameters, or a {name: tensor} dict (names are then the keys `set_grads` matches)."""
    eight_bit = False

    def __init__(self, params: Union[Iterable[torch.Tensor], Dict[str, torch.Tensor]], lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, min_8bit_size: int = 4096):
        if isinstance(params, dict):
            self.names, self.params = list(params.keys()), list(params.values())
        else:
            self.params = list(params)
            self.names = [str(i) for i in range(len(self.params))]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        for n, p in zip(self.names, self.params):
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise TypeError(f"{type(self).__name__}: parameter {n} must be a contiguous fp32 CUDA tensor (the reference keeps the "
                                f"UNet in fp32 while training, train_StorySalon_stage2.py:226-235), got {p.dtype} on {p.device}")
        self.param_groups = [dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay), params=self.params)]
        self.min_8bit_size = int(min_8bit_size)
        self.dev = self.params[0].device
        self.step_count = 0
        self.state: Dict[int, dict] = {}
        self._grads: Optional[Dict[str, torch.Tensor]] = None
        self._sumsq = torch.zeros(len(self.params), dtype=torch.float32, device=self.dev)    # per-tensor sums of squares ...
        self._total = torch.zeros(1, dtype=torch.float32, device=self.dev)                   # ... and their sum, read by the step kernels
        self._scratch = torch.empty(lib.sg_sumsq_scratch_floats(), dtype=torch.float32, device=self.dev)
        self._clip: Optional[float] = None
        if self.eight_bit:
            self._code1, self._code2 = create_dynamic_map(True).to(self.dev), create_dynamic_map(False).to(self.dev)
            self._zero1 = int((self._code1 == 0).nonzero()[0])
            self._zero2 = int((self._code2 == 0).nonzero()[0])

    # ------------------------------------------------------------------------------------------------ gradients
    def set_grads(self, grads: Dict[str, torch.Tensor]) -> None:
        """Use this {name: fp32 gradient} dict (UNetTrainer.train_step's second result) instead of `p.grad` for the next step."""
        missing = [n for n in self.names if n not in grads]
        if missing:
            raise KeyError(f"set_grads: no gradient for {missing[:3]} (+{max(0, len(missing) - 3)})")
        self._grads = grads

    def _grad(self, i: int) -> Optional[torch.Tensor]:
        g = self._grads[self.names[i]] if self._grads is not None else self.params[i].grad
        if g is None:
            return None
        if g.dtype != torch.float32 or not g.is_cuda or g.numel() != self.params[i].numel():
            raise TypeError(f"gradient of {self.names[i]} must be an fp32 CUDA tensor of {self.params[i].numel()} elements")
        return g if g.is_contiguous() else g.contiguous()

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ over every parameter that has a gradient: returns the total norm (a device scalar — reading
        it is the caller's choice) and arms the clipping of the next `step()`, which scales the gradients on the fly."""
        self._sumsq.zero_()
        for i in range(len(self.params)):
            g = self._grad(i)
            if g is not None:
                check(lib.sg_sumsq_f32(g.data_ptr(), g.numel(), self._sumsq[i:].data_ptr(), self._scratch.data_ptr(), _stream()), "sg_sumsq_f32")
        torch.sum(self._sumsq, dim=0, keepdim=True, out=self._total)
        self._clip = float(max_norm)
        return self._total.sqrt()[0]

    def zero_grad(self, set_to_none: bool = True) -> None:
        self._grads = None
        self._clip = None
        for p in self.params:
            if getattr(p, "grad", None) is not None:
                p.grad = None

    # ----------------------------------------------------------------------------------------------------- step
    def _state(self, i: int) -> dict:
        st = self.state.get(i)
        if st is None:
            p = self.params[i]
            n = p.numel()
            if self.eight_bit and n >= self.min_8bit_size:
                nb = lib.sg_adamw8bit_blocks(n)
                st = dict(bits=8, code1=torch.full((n,), self._zero1, dtype=torch.uint8, device=self.dev),
                          code2=torch.full((n,), self._zero2, dtype=torch.uint8, device=self.dev),
                          absmax1=torch.zeros(nb, dtype=torch.float32, device=self.dev),
                          absmax2=torch.zeros(nb, dtype=torch.float32, device=self.dev))
            else:
                st = dict(bits=32, exp_avg=torch.zeros(n, dtype=torch.float32, device=self.dev),
                          exp_avg_sq=torch.zeros(n, dtype=torch.float32, device=self.dev))
            self.state[i] = st
        return st

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0) -> None:
        """One update of every parameter that has a gradient.  grad_scale multiplies the gradients first (1 / loss scale when the
        caller scaled the loss itself; UNetTrainer's gradients are already unscaled)."""
        self.step_count += 1
        g0 = self.param_groups[0]
        for i, p in enumerate(self.params):
            g = self._grad(i)
            if g is None:
                continue
            st = self._state(i)
            d = AdamWDesc()
            d.param, d.grad, d.n = p.data_ptr(), g.data_ptr(), p.numel()
            d.lr, (d.beta1, d.beta2), d.eps, d.weight_decay = g0["lr"], g0["betas"], g0["eps"], g0["weight_decay"]
            d.step, d.grad_scale = self.step_count, float(grad_scale)
            if self._clip is not None:
                d.sumsq, d.n_sumsq, d.max_norm = self._total.data_ptr(), 1, self._clip
            if st["bits"] == 8:
                d.code1, d.code2 = st["code1"].data_ptr(), st["code2"].data_ptr()
                d.absmax1, d.absmax2 = st["absmax1"].data_ptr(), st["absmax2"].data_ptr()
                d.q_code1, d.q_code2 = self._code1.data_ptr(), self._code2.data_ptr()
                check(lib.sg_adamw8bit(C.byref(d), _stream()), "sg_adamw8bit")
            else:
                d.exp_avg, d.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                check(lib.sg_adamw_f32(C.byref(d), _stream()), "sg_adamw_f32")
            torch.autograd.graph.increment_version(p)          # the kernel wrote p behind autograd's back
        self._clip = None

    # ------------------------------------------------------------------------------------------------ (de)serialisation
    def state_dict(self) -> dict:
        return dict(step=self.step_count, param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}],
                    names=list(self.names), state={self.names[i]: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
                                                   for i, st in self.state.items()})

    def load_state_dict(self, sd: dict) -> None:
        self.step_count = int(sd["step"])
        self.param_groups[0].update(sd["param_groups"][0])
        index = {n: i for i, n in enumerate(self.names)}
        self.state = {}
        for name, st in sd["state"].items():
            i = index[name]
            want = self._state(i)
            if want["bits"] != st["bits"]:
                raise ValueError(f"optimizer state of {name} is {st['bits']}-bit, this optimizer keeps it in {want['bits']} bits")
            for k, v in st.items():
                if torch.is_tensor(v):
                    want[k].copy_(v)


class AdamW8bit(AdamW):
    """bitsandbytes.optim.AdamW8bit's algorithm on sg_adamw8bit (block-wise 8-bit moments); see the module docstring.
    One known deviation in operation order: the decoupled weight decay is applied to the parameter BEFORE the Adam update here (as
    torch.optim.AdamW does), where bitsandbytes' block-wise kernel applies p *= (1 - lr * wd) AFTER it — the results differ by
    lr^2 * wd * (update), i.e. 1e-12 relative per step at the reference's lr = 1e-5, wd = 1e-2."""
    eight_bit = True

    def state_bytes(self) -> int:
        return sum(v.numel() * v.element_size() for st in self.state.values() for v in st.values() if torch.is_tensor(v))


# ---------------------------------------------------------------------------------------------------- learning-rate schedules
SCHEDULES = ("constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial")


LR_END = 1e-7      # get_polynomial_decay_schedule_with_warmup's default lr_end (get_scheduler does not expose it)


def _multiplier(name: str, warmup: int, total: Optional[int], cycles: float, power: float, lr_init: float = 1.0):
    if name not in SCHEDULES:
        raise ValueError(f"{name!r} is not a valid lr scheduler, choose one of {SCHEDULES}")
    if name not in ("constant", "constant_with_warmup") and total is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")

    def f(s: int) -> float:
        if name == "constant":
            return 1.0
        if s < warmup:
            return float(s) / float(max(1, warmup))
        if name == "constant_with_warmup":
            return 1.0
        span = float(max(1, total - warmup))
        if name == "linear":
            return max(0.0, float(total - s) / span)
        prog = float(s - warmup) / span
        if name == "cosine":
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * prog)))
        if name == "cosine_with_restarts":
            return 0.0 if prog >= 1.0 else max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(cycles) * prog) % 1.0))))
        # polynomial: decays from the optimizer's initial lr to lr_end; LambdaLR multiplies by lr_init again
        if s > total:
            return LR_END / lr_init
        return ((lr_init - LR_END) * (1.0 - (s - warmup) / (total - warmup)) ** power + LR_END) / lr_init
    return f


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR as the loop uses it: `step()` after every optimizer step, `get_last_lr()` for the log."""

    def __init__(self, optimizer: AdamW, fn):
        self.optimizer, self.fn = optimizer, fn
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def _apply(self):
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = base * self.fn(self.last_epoch)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self) -> List[float]:
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self) -> dict:
        return dict(last_epoch=self.last_epoch, base_lrs=list(self.base_lrs))

    def load_state_dict(self, sd: dict) -> None:
        self.last_epoch, self.base_lrs = int(sd["last_epoch"]), list(sd["base_lrs"])
        self._apply()


def get_scheduler(name: str, optimizer: AdamW, num_warmup_steps: Optional[int] = None, num_training_steps: Optional[int] = None,
                  num_cycles: int = 1, power: float = 1.0) -> LambdaLR:
    """diffusers.optimization.get_scheduler (train_StorySalon_stage2.py:214-219).  As there, `num_cycles` reaches only the
    hard-restart schedule (the cosine schedule keeps its half cycle) and `power` only the polynomial one, whose end value is the
    library default lr_end = 1e-7 relative to the optimizer's initial learning rate.  The multipliers are pinned against
    transformers.optimization, whose functions diffusers' optimization.py restates (tests/test_optim_host.py)."""
    name = getattr(name, "value", name)
    if name != "constant" and num_warmup_steps is None:
        raise ValueError(f"{name} requires `num_warmup_steps`, please provide that argument.")
    cycles = float(num_cycles) if name == "cosine_with_restarts" else 0.5
    lr_init = float(getattr(optimizer, "defaults", {}).get("lr", optimizer.param_groups[0]["lr"]))
    return LambdaLR(optimizer, _multiplier(name, int(num_warmup_steps or 0), num_training_steps, cycles, power, lr_init))
