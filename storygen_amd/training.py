"""The body of the reference's stage-2 / COCO training loop on the HIP path:

  /root/reference/train_StorySalon_stage2.py:167-177   freeze everything, attn3 trainable
                                            :186-205   AdamW8bit / AdamW over the trainable parameters
                                            :214-219   get_scheduler
                                            :258-302   per-step plumbing: VAE-encode frames, CLIP-encode prompts, noise, timesteps
                                            :304-327   1-3 reference passes -> main pass -> masked MSE -> backward
                                            :328-333   clip_grad_norm_, optimizer.step(), lr_scheduler.step(), zero_grad()
                                            :348-357   checkpoint = the whole pipeline in diffusers folder layout
(train_COCO.py:286-330 is the same loop with all three frames at one noise level and an unmasked loss: `variant="coco"`;
train_StorySalon_stage1.py:258-300 is the loop without reference passes and with the `attn1` modules trainable:
`Stage2Trainer(..., trainable_modules=("attn1",))`).  `Stage2Trainer.step` takes either the raw batch of the
reference's dataset (images, prompts, masks — encoded here by the HIP AutoencoderKL / CLIPTextModel) or the already-encoded tensors
(`storygen_amd.synth.synthetic_train_batch`).  Data parallelism: one process per GPU, the attn3 gradients (49.6 M values) averaged
by ONE RCCL all-reduce per optimizer step (`train.allreduce_gradients`), exactly where accelerate's DDP wrapper would do it.
Out of scope (§8): the dataset classes, wandb / tensorboard trackers, the validation sample logger."""
from __future__ import annotations

import os
import random
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .optim import AdamW, AdamW8bit, get_scheduler
from .train import UNetTrainer, allreduce_gradients, any_rank


def accumulation_plan(mode: str, accum: int, world: int) -> dict:
    """What `gradient_accumulation_steps = k` means per call of Stage2Trainer.step (pure host logic; tests/test_optim_host.py pins it).

    mode "reference" = what train_StorySalon_stage2.py ACTUALLY does (:326-332).  The loop never enters `accelerator.accumulate`, so
    `accelerator.sync_gradients` is always True: every micro-batch runs backward on loss / k (accelerate scales the loss by
    1 / gradient_accumulation_steps), clips, steps the optimizer, steps the LR scheduler and zeroes the gradients, and `step` counts
    micro-batches.  The scheduler was built with warm-up and total steps multiplied by k (:215-220) and accelerate's wrapper advances
    it `num_processes` times per call (AcceleratedScheduler.step with split_batches = False).  So k only scales the gradients (and,
    with scale_lr, the learning rate) and stretches the schedule.
    mode "true" = real accumulation: k micro-batches are summed (each scaled 1 / k), then ONE clip + optimizer step + scheduler step;
    warm-up / total steps count optimizer steps (unscaled), the schedule does not depend on the number of processes."""
    if mode not in ("reference", "true"):
        raise ValueError(f"accumulation must be 'reference' or 'true', got {mode!r}")
    k = int(accum)
    if mode == "reference":
        return dict(grad_scale=1.0 / k, step_every=1, scheduler_steps_per_optimizer_step=int(world), schedule_multiplier=k)
    return dict(grad_scale=1.0 / k, step_every=k, scheduler_steps_per_optimizer_step=1, schedule_multiplier=1)


def use_refs_for(p: float) -> tuple:
    """The random number of reference frames of train_StorySalon_stage2.py:306-313: p < 0.3 -> frames 0,1,2; p < 0.6 -> 1,2; else 2."""
    return tuple(i for i in range(3) if (p < 0.3) or (0.3 <= p < 0.6 and i > 0) or (p >= 0.6 and i > 1))


class Stage2Trainer:
    def __init__(self, unet, batch_size: int, height: int = 64, width: int = 64, n_ref: int = 3, learning_rate: float = 1e-5,
                 adam_beta1: float = 0.9, adam_beta2: float = 0.999, adam_weight_decay: float = 1e-2, adam_epsilon: float = 1e-8,
                 use_8bit_adam: bool = True, max_grad_norm: float = 1.0, lr_scheduler: str = "constant", lr_warmup_steps: int = 0,
                 train_steps: Optional[int] = None, gradient_accumulation_steps: int = 1, scale_lr: bool = False,
                 trainable_modules: Sequence[str] = ("attn3",), use_graph: bool = True, vae=None, text_encoder=None, tokenizer=None,
                 seed: Optional[int] = None, variant: str = "storysalon", accumulation: str = "reference"):
        """unet: the drop-in UNet2DConditionModel in fp32 on the HIP device (the reference keeps it fp32, :226-235).  height / width are
        latent sizes.  vae / text_encoder / tokenizer are only needed for raw batches (`encode_batch`).  variant="coco" =
        train_COCO.py:286-316: always the three frames, each at noise level ref_t (no `* (3 - i)`), loss without the mask.
        accumulation: what gradient_accumulation_steps means — "reference" (default) = the reference's loop as written, "true" = real
        gradient accumulation (accumulation_plan)."""
        if variant not in ("storysalon", "coco"):
            raise ValueError(f"variant must be 'storysalon' or 'coco', got {variant!r}")
        self.variant = variant
        if tuple(trainable_modules) not in (("attn3",), ("attn1",)):
            raise NotImplementedError("the HIP backward produces weight gradients for the attn3 modules (stage 2 / COCO, "
                                      "train_StorySalon_stage2.py:170-177) or the attn1 modules (stage 1, train_StorySalon_stage1.py:175)")
        self.module = tuple(trainable_modules)[0]
        if self.module == "attn1":
            n_ref = 0                                                              # stage 1 has no prior frames (:288 image_hidden_states=None)
        if unet.dtype != torch.float32 or unet.device.type != "cuda":
            raise TypeError("Stage2Trainer: the UNet must be fp32 on the HIP device")
        self.unet, self.vae, self.text_encoder, self.tokenizer = unet, vae, text_encoder, tokenizer
        self.dev = unet.device
        unet.requires_grad_(False)                                                               # :167-177
        self.named: Dict[str, torch.Tensor] = {}
        for name, p in unet.named_parameters():
            if f".{self.module}." in name:
                p.requires_grad = True
                self.named[name] = p
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        if scale_lr:                                                                             # :181-184
            learning_rate = learning_rate * gradient_accumulation_steps * batch_size * world
        cls = AdamW8bit if use_8bit_adam else AdamW
        # .detach() shares storage AND the autograd version counter with the parameter, so the optimizer's in-place update shows up in
        # the drop-in UNet's weight-staleness tag (unet_2d_condition.py::_engine_weights)
        self.optimizer = cls({n: p.detach() for n, p in self.named.items()}, lr=learning_rate, betas=(adam_beta1, adam_beta2),
                             weight_decay=adam_weight_decay, eps=adam_epsilon)
        self.plan = accumulation_plan(accumulation, gradient_accumulation_steps, world)
        mult = self.plan["schedule_multiplier"]
        self.lr_scheduler = get_scheduler(lr_scheduler, self.optimizer, num_warmup_steps=lr_warmup_steps * mult,
                                          num_training_steps=None if train_steps is None else train_steps * mult)
        self.max_grad_norm, self.accum = max_grad_norm, int(gradient_accumulation_steps)
        self.trainer = UNetTrainer(unet._arch, unet.state_dict(), self.dev, batch_size, height, width, n_ref=n_ref,
                                   weights=unet._engine_weights(), trainable=self.module)
        if variant == "coco":
            self.trainer.ref_levels = "coco"
        self.use_graph = use_graph
        self.global_step, self._micro = 0, 0
        self._tainted = False                      # "true" accumulation: a micro-batch of the open window was skipped (non-finite gradients)
        self._acc: Optional[Dict[str, torch.Tensor]] = None
        self._rng = random.Random(seed)
        self.last_grad_norm: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------------------------------- per-step plumbing
    @torch.no_grad()
    def encode_batch(self, batch: dict, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        """train_StorySalon_stage2.py:265-302: raw dataset batch {image [b,3,H,W], mask [b,3,H,W], prompt [b strings], ref_image
        [b,3,3,H,W], ref_prompt [3 lists of b strings]} -> the tensors the device step consumes."""
        if self.vae is None or self.text_encoder is None or self.tokenizer is None:
            raise ValueError("encode_batch needs vae, text_encoder and tokenizer")
        dev = self.dev
        tok = lambda s: self.tokenizer(s, truncation=True, padding="max_length", max_length=self.tokenizer.model_max_length,   # noqa: E731
                                       return_tensors="pt").input_ids
        image = batch["image"].to(dev)
        mask = batch["mask"].to(dev, torch.float32)[:, [0]].repeat(1, 4, 1, 1)
        mask = F.interpolate(mask, scale_factor=1 / 8.0, mode="bilinear", align_corners=False)
        latents = self.vae.encode(image).latent_dist.sample(generator) * 0.18215
        b = latents.shape[0]
        gdev = generator.device if generator is not None else dev
        noise = torch.randn(latents.shape, generator=generator, device=gdev)
        ref_noise = torch.randn(latents.shape, generator=generator, device=gdev) if self.module == "attn3" else None   # stage 1 draws none (:277)
        timesteps = torch.randint(0, 1000, (b,), generator=generator, device=gdev)
        text = self.text_encoder(tok(batch["prompt"]).to(dev))[0]
        out = dict(latents=latents.float(), noise=noise, timesteps=timesteps, text=text, mask=mask)
        if self.module == "attn3":                                                  # prior frames and their prompts (:276-288)
            refs = torch.transpose(batch["ref_image"].to(dev), 0, 1)
            out["ref_latents"] = torch.stack([self.vae.encode(r).latent_dist.sample(generator) * 0.18215 for r in refs]).float()
            out["ref_noise"] = ref_noise
            out["prev_text"] = torch.stack([self.text_encoder(tok(p).to(dev))[0] for p in batch["ref_prompt"]])
        return out

    # ------------------------------------------------------------------------------------------------------- step
    def step(self, batch: dict, use_refs: Optional[Sequence[int]] = None) -> Dict[str, object]:
        """One micro-batch.  accumulation "reference" (default): clips, all-reduces and applies the optimizer on the gradient of
        loss / k every call, as the reference's loop does; "true": only every k-th call (accumulation_plan).
        Returns {"loss": device scalar tensor, "lr": float, "optimizer_step": bool}."""
        if "image" in batch:
            batch = self.encode_batch(batch)
        if self.module == "attn1":
            use_refs = ()
        elif self.variant == "coco":
            use_refs = (0, 1, 2)                                                       # train_COCO.py:301: every frame, every step
            batch = dict(batch, mask=torch.zeros_like(batch["mask"]))                  # :315: unmasked MSE
        elif use_refs is None:
            use_refs = use_refs_for(self._rng.uniform(0, 1))
        run = self.trainer.train_step_graph if self.use_graph else self.trainer.train_step
        loss, grads = run(batch, use_refs=tuple(use_refs))
        loss = loss.detach().clone()
        plan = self.plan
        # non-finite gradients at every loss scale: GradScaler skips the step (only the hipGraph step rescales and detects this; the
        # eager train_step runs at one fixed loss scale and does not check).  The micro-batch is not accumulated and taints its window.
        skipped = bool(self.use_graph and self.trainer.last_step_skipped)
        if skipped:
            self._tainted = True
        elif plan["step_every"] > 1:                                # real accumulation: sum k micro-batches, each scaled 1 / k
            if self._acc is None:
                self._acc = {n: torch.zeros_like(g) for n, g in grads.items()}
            for n, g in grads.items():
                self._acc[n].add_(g, alpha=plan["grad_scale"])
            grads = self._acc
        elif plan["grad_scale"] != 1.0:                             # reference: every micro-batch steps on the gradient of loss / k
            grads = {n: g * plan["grad_scale"] for n, g in grads.items()}
        self._micro += 1
        stepped = self._micro % plan["step_every"] == 0
        if not stepped:
            return dict(loss=loss, lr=self.lr_scheduler.get_last_lr()[0], optimizer_step=False, skipped=skipped)
        # The window closes here on EVERY rank (the micro-batch count is the same everywhere).  Whether it steps is decided collectively:
        # torch's GradScaler under DDP skips the optimizer step of the whole window on all ranks when any rank saw a non-finite gradient
        # (the inf travels through the gradient all-reduce); a rank-local decision would leave the other ranks alone in the all-reduce
        # below and desynchronise parameters and global_step.
        if any_rank(self._tainted, self.dev):
            self._drop_window()
            return dict(loss=loss, lr=self.lr_scheduler.get_last_lr()[0], optimizer_step=False, skipped=True)
        if stepped:
            allreduce_gradients(grads)                              # DDP's gradient average, :222
            self.optimizer.set_grads(grads)
            self.last_grad_norm = self.optimizer.clip_grad_norm_(self.max_grad_norm)                 # :329-330
            self.optimizer.step()
            self.optimizer.zero_grad()
            if self._acc is not None:
                for a in self._acc.values():
                    a.zero_()
            self.trainer.set_trainable_parameters(self.named)       # refresh the fp16 operand copies the kernels read
            self.global_step += 1
            for _ in range(plan["scheduler_steps_per_optimizer_step"]):     # :331 through accelerate's scheduler wrapper
                self.lr_scheduler.step()
        return dict(loss=loss, lr=self.lr_scheduler.get_last_lr()[0], optimizer_step=stepped)

    def _drop_window(self):
        """Close an accumulation window without an optimizer step (a micro-batch of it had non-finite gradients)."""
        self._tainted = False
        if self._acc is not None:
            for a in self._acc.values():
                a.zero_()

    # ------------------------------------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, logdir: str, scheduler=None) -> str:
        """:348-357: `StableDiffusionPipeline(vae, text_encoder, tokenizer, unet, scheduler).save_pretrained(logdir/checkpoint_<step>)`,
        plus the optimizer / lr-scheduler state next to it (the reference relies on accelerate's save_state for those)."""
        from .model import StableDiffusionPipeline
        path = os.path.join(logdir, f"checkpoint_{self.global_step}")
        StableDiffusionPipeline(vae=self.vae, text_encoder=self.text_encoder, tokenizer=self.tokenizer, unet=self.unet,
                                scheduler=scheduler).save_pretrained(path)
        torch.save(dict(optimizer=self.optimizer.state_dict(), lr_scheduler=self.lr_scheduler.state_dict(), global_step=self.global_step),
                   os.path.join(path, "training_state.pt"))
        return path

    def load_training_state(self, path: str) -> None:
        st = torch.load(os.path.join(path, "training_state.pt"), map_location="cpu", weights_only=False)
        self.optimizer.load_state_dict(st["optimizer"])
        self.lr_scheduler.load_state_dict(st["lr_scheduler"])
        self.global_step = int(st["global_step"])


def train(trainer: Stage2Trainer, batches, train_steps: int, checkpointing_steps: int = 0, logdir: Optional[str] = None, scheduler=None,
          log=None) -> List[float]:
    """`while step < train_steps` of :258-361 over an iterable of batches; returns the per-optimizer-step losses."""
    losses: List[float] = []
    it = iter(batches)
    while trainer.global_step < train_steps:
        out = trainer.step(next(it))
        if out["optimizer_step"]:
            losses.append(float(out["loss"]))
            if log is not None:
                log(dict(step=trainer.global_step, loss=losses[-1], lr=out["lr"]))
            if checkpointing_steps and logdir and trainer.global_step % checkpointing_steps == 0:
                trainer.save_checkpoint(logdir, scheduler)
    return losses
