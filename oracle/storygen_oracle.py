"""ORACLE — test infrastructure, not product code.

A plain-PyTorch fp32 CPU restatement of StoryGen's denoising hot path: the custom UNet forward (with feature
harvest / Visual-Language-Context consumption) and the per-step sampling loop.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it, and only as the checker / the
timed CPU baseline ("port") — never as a fallback for the HIP path.

Parity pinning: the reference ships NO tests or golden vectors for this path, and its leaf
arithmetic lives in the un-vendored `diffusers==0.13.1`.  This restatement is therefore pinned against outputs
of the reference's own `model/*.py` executed verbatim (on the clean-room shim in oracle/diffusers_shim) by
`oracle/make_golden.py` in the build container; the resulting vectors are committed under tests/golden/ and
`tests/test_oracle_golden.py` re-checks this file against them wherever the tests run.

Every function cites the reference lines it follows.  Functional style over a flat state dict (the reference's
checkpoint keys), so it is independent of the product's module classes.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------- leaf ops
def timestep_embedding(t: Tensor, dim: int, flip_sin_to_cos: bool, shift: float) -> Tensor:
    """diffusers `Timesteps` as used at unet_2d_condition.py:138,392: [cos|sin] of t*exp(-ln(1e4) k/(half-shift))."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / (half - shift))
    e = t[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(e), torch.sin(e)], -1) if flip_sin_to_cos else torch.cat([torch.sin(e), torch.cos(e)], -1)


def resnet_block(sd: SD, p: str, x: Tensor, emb: Tensor, groups: int, eps: float) -> Tensor:
    """diffusers ResnetBlock2D(time_embedding_norm='default', output_scale_factor=1) — SURVEY §8a row a10;
    instantiated at unet_2d_blocks.py:331-344 et al."""
    h = F.silu(F.group_norm(x, groups, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps))
    h = F.conv2d(h, sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    t = F.linear(F.silu(emb), sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"])
    h = h + t[:, :, None, None]
    h = F.silu(F.group_norm(h, groups, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps))
    h = F.conv2d(h, sd[f"{p}.conv2.weight"], sd[f"{p}.conv2.bias"], padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[f"{p}.conv_shortcut.weight"], sd[f"{p}.conv_shortcut.bias"])
    return x + h


def attention(sd: SD, p: str, x: Tensor, ctx: Optional[Tensor], heads: int, q_chunk: int = 1024) -> Tensor:
    """diffusers CrossAttention + CrossAttnProcessor: to_out(softmax(q k^T d^-1/2) v); heads = 8, d = C/8
    (attention.py:175-223 ctor).  Query-chunked to bound memory (row-wise softmax)."""
    src = x if ctx is None else ctx
    q = F.linear(x, sd[f"{p}.to_q.weight"])
    k = F.linear(src, sd[f"{p}.to_k.weight"])
    v = F.linear(src, sd[f"{p}.to_v.weight"])
    b, nq, c = q.shape
    d = c // heads
    q = q.view(b, nq, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    scale = d ** -0.5
    outs = []
    for s in range(0, nq, q_chunk):
        a = torch.softmax(torch.matmul(q[:, :, s:s + q_chunk], k.transpose(-1, -2)) * scale, dim=-1)
        outs.append(torch.matmul(a, v))
    o = torch.cat(outs, 2).transpose(1, 2).reshape(b, nq, c)
    return F.linear(o, sd[f"{p}.to_out.0.weight"], sd[f"{p}.to_out.0.bias"])


def transformer_block(sd: SD, p: str, h: Tensor, text: Tensor, image_ctx: Optional[Tensor], heads: int
                      ) -> Tuple[Tensor, Tensor]:
    """BasicTransformerBlock.forward, attention.py:236-302, incl. the doubled residual `(a2+h)+(a3+h)` (:277,291-293)
    and the harvested feature = state after the self-attention residual (:262-263)."""
    c = h.shape[-1]

    def ln(n, t):
        return F.layer_norm(t, (c,), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)

    h = attention(sd, f"{p}.attn1", ln("norm1", h), None, heads) + h                      # :250-262
    feature = h.clone()                                                                   # :263
    ht = attention(sd, f"{p}.attn2", ln("norm2", h), text, heads) + h                     # :266-277
    if image_ctx is not None:
        hi = attention(sd, f"{p}.attn3", ln("norm4", h), image_ctx, heads) + h            # :281-291
        h = ht + hi                                                                       # :293
    else:
        h = ht                                                                            # :295
    n3 = ln("norm3", h)                                                                   # :298
    proj = F.linear(n3, sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.0.proj.bias"])   # GEGLU :381-393
    val, gate = proj.chunk(2, dim=-1)
    ff = F.linear(val * F.gelu(gate), sd[f"{p}.ff.net.2.weight"], sd[f"{p}.ff.net.2.bias"])
    return ff + h, feature                                                                # :300


def transformer_2d(sd: SD, p: str, x: Tensor, text: Tensor, image_ctx: Optional[Tensor], heads: int, groups: int
                   ) -> Tuple[Tensor, Tensor]:
    """Transformer2DModel.forward, attention.py:85-128 (GroupNorm eps is 1e-6 here, :55)."""
    b, c, hh, ww = x.shape
    h = F.group_norm(x, groups, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    h = F.conv2d(h, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    h, feature = transformer_block(sd, f"{p}.transformer_blocks.0", h, text, image_ctx, heads)
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    h = F.conv2d(h, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    return h + x, feature


# ----------------------------------------------------------------------------------------------- the UNet
def unet_forward(sd: SD, cfg: dict, sample: Tensor, timestep, text: Tensor,
                 image_hidden_states: Optional[Dict[str, Tensor]] = None
                 ) -> Tuple[Tensor, Dict[str, Tensor]]:
    """UNet2DConditionModel.forward, unet_2d_condition.py:338-485, for the SD-1.5-style config.

    Returns (sample, img_dif_conditions).  When `image_hidden_states` is None the 16 features are harvested
    (blocks' first branch, unet_2d_blocks.py:383-396,606-620,273-284); otherwise they are consumed by attn3 and the
    returned dict is empty.  Feature keys are by block index."""
    boc = list(cfg["block_out_channels"])
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    ahd = cfg["attention_head_dim"]
    heads = list(ahd) if isinstance(ahd, (list, tuple)) else [ahd] * len(boc)
    lpb = cfg["layers_per_block"]
    harvest = image_hidden_states is None
    feats: Dict[str, Tensor] = {}

    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])               # :379-390
    if t.dim() == 0:
        t = t[None]
    t = t.expand(sample.shape[0])
    emb = timestep_embedding(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])        # :392
    emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])  # :398

    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)             # :411
    skips: List[Tensor] = [h]
    for i, typ in enumerate(cfg["down_block_types"]):                                     # :417-433
        for j in range(lpb):
            h = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", h, emb, groups, eps)
            if typ == "CrossAttnDownBlock2D":
                key = f"down_{i + 1}_{j + 1}"
                ctx = None if harvest else image_hidden_states[key]
                h, f = transformer_2d(sd, f"down_blocks.{i}.attentions.{j}", h, text, ctx, heads[i], groups)
                if harvest:
                    feats[key] = f
            skips.append(h)
        if i != len(boc) - 1:
            w = f"down_blocks.{i}.downsamplers.0.conv"
            h = F.conv2d(h, sd[f"{w}.weight"], sd[f"{w}.bias"], stride=2, padding=cfg["downsample_padding"])
            skips.append(h)

    h = resnet_block(sd, "mid_block.resnets.0", h, emb, groups, eps)                      # :436-445
    ctx = None if harvest else image_hidden_states["mid"]
    h, f = transformer_2d(sd, "mid_block.attentions.0", h, text, ctx, heads[-1], groups)
    if harvest:
        feats["mid"] = f
    h = resnet_block(sd, "mid_block.resnets.1", h, emb, groups, eps)

    rheads = list(reversed(heads))
    for i, typ in enumerate(cfg["up_block_types"]):                                       # :448-475
        for j in range(lpb + 1):
            h = torch.cat([h, skips.pop()], dim=1)                                        # unet_2d_blocks.py:609,626,716
            h = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", h, emb, groups, eps)
            if typ == "CrossAttnUpBlock2D":
                key = f"up_{i}_{j + 1}"
                ctx = None if harvest else image_hidden_states[key]
                h, f = transformer_2d(sd, f"up_blocks.{i}.attentions.{j}", h, text, ctx, rheads[i], groups)
                if harvest:
                    feats[key] = f
        if i != len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")                        # diffusers Upsample2D
            w = f"up_blocks.{i}.upsamplers.0.conv"
            h = F.conv2d(h, sd[f"{w}.weight"], sd[f"{w}.bias"], padding=1)

    h = F.silu(F.group_norm(h, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))  # :477-479
    h = F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)                # :480
    return h, feats


# ----------------------------------------------------------------------------------------------- scheduler
class DDIM:
    """DDIMScheduler pieces used by the loop (pipeline.py:366-367,420-424,461) for the shipped
    scheduler_config.json: scaled-linear betas 0.00085..0.012 (fp32), 1000 train steps, steps_offset=1,
    set_alpha_to_one=False, clip_sample=False, eta=0."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 set_alpha_to_one=False):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.n_train = num_train_timesteps
        self.steps_offset = steps_offset

    def timesteps(self, n: int) -> List[int]:
        ratio = self.n_train // n
        return [int(round(i * ratio)) + self.steps_offset for i in reversed(range(n))]

    def add_noise(self, x: Tensor, noise: Tensor, t: int) -> Tensor:
        a = self.alphas_cumprod[t]
        return a ** 0.5 * x + (1 - a) ** 0.5 * noise

    def step(self, eps: Tensor, t: int, x: Tensor, n: int) -> Tensor:
        prev = t - self.n_train // n
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps


class PNDM(DDIM):
    """PNDMScheduler (diffusers 0.13.1) with skip_prk_steps=True — the class
    /root/reference/ckpt/stable-diffusion-v1-5/scheduler/scheduler_config.json:2 names and /root/reference/model/pipeline.py:7-16
    accepts.  PARITY UNPINNED: diffusers is not installed here and the reference's shipped scripts only ever build a DDIMScheduler
    (inference.py:48), so this restates the published algorithm (`set_timesteps`, `step_plms`, `_get_prev_sample`) as written
    there — a stateful list of past epsilons — and is the checker for the device's table-driven form of the same rule."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ets: List[Tensor] = []
        self.counter = 0
        self.cur_sample: Optional[Tensor] = None

    def timesteps(self, n: int) -> List[int]:
        ratio = self.n_train // n
        base = [int(round(i * ratio)) + self.steps_offset for i in range(n)]
        self.ets, self.counter, self.cur_sample = [], 0, None
        return (base[:-1] + base[-2:-1] + base[-1:])[::-1]

    def _prev(self, x: Tensor, t: int, prev: int, e: Tensor) -> Tensor:
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        coeff = (a_p / a_t) ** 0.5
        denom = a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5
        return coeff * x - (a_p - a_t) * e / denom

    def step(self, eps: Tensor, t: int, x: Tensor, n: int) -> Tensor:
        ratio = self.n_train // n
        prev = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev, t = t, t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = x
        elif len(self.ets) == 1 and self.counter == 1:
            eps = (eps + self.ets[-1]) / 2
            x, self.cur_sample = self.cur_sample, None
        elif len(self.ets) == 2:
            eps = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            eps = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            eps = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        self.counter += 1
        return self._prev(x, t, prev, eps)


# ----------------------------------------------------------------------------------------------- the loop
def denoise_step(sd: SD, cfg: dict, sched: DDIM, latents: Tensor, t: int, n_steps: int, inputs: Dict[str, Tensor],
                 stage: str, guidance_scale: float, image_guidance_scale: float) -> Tensor:
    """One iteration of pipeline.py:412-461 with classifier-free guidance on.

    R reference passes (batch 3N: [zero, img, img] latents with [uncond, text_i, text_i]) harvest features, which
    are concatenated on the token axis (:440-443) and consumed by the main pass (batch 3N: latents x3 with
    [uncond, uncond, text], :448-453); then 3-way CFG (:457-458) and the DDIM update (:461)."""
    imgs, zero, noise = inputs["image_prompts"], inputs["zero_prompt"], inputs["noise"]
    n_ref = imgs.shape[0]
    ref_t = int(t) // 10 if t >= 0 else 0                                                 # :414-415 (t/10).long()
    feats_all = []
    for i in range(n_ref if stage != "no" else 0):                                        # stage 'no': no reference pass (:436-438)
        ti = ref_t * (n_ref - i) if stage == "auto-regressive" else ref_t                 # :419-424
        noisy_img = sched.add_noise(imgs[i], noise, ti)
        noisy_zero = sched.add_noise(zero, noise, ti)
        x = torch.cat([noisy_zero, noisy_img, noisy_img])                                 # :429
        e = torch.cat([inputs["prev_uncond"][i], inputs["prev_text"][i], inputs["prev_text"][i]])  # :430
        feats_all.append(unet_forward(sd, cfg, x, ti, e, None)[1])                        # :433-435
    ctx = {k: torch.cat([f[k] for f in feats_all], dim=1) for k in feats_all[0]} if feats_all else None   # :440-445
    e = torch.cat([inputs["uncond"], inputs["uncond"], inputs["text"]])                   # :448
    x = torch.cat([latents] * 3)                                                          # :450
    eps = unet_forward(sd, cfg, x, t, e, ctx)[0]                                          # :453
    eu, ei, ea = eps.chunk(3)
    eps = eu + image_guidance_scale * (ei - eu) + guidance_scale * (ea - ei)              # :457-458
    return sched.step(eps, t, latents, n_steps)                                           # :461


def sample_loop(sd: SD, cfg: dict, inputs: Dict[str, Tensor], n_steps: int, stage: str = "multi-image-condition",
                guidance_scale: float = 7.5, image_guidance_scale: float = 3.5, max_steps: Optional[int] = None,
                trace: Optional[list] = None, scheduler: str = "ddim") -> Tensor:
    """pipeline.py:366-367 + :411-469: all (or the first `max_steps`) steps of the DDIM (default) or PNDM loop; returns the
    latents."""
    sched = DDIM() if scheduler == "ddim" else PNDM()
    latents = inputs["latents"].clone()
    with torch.no_grad():
        for k, t in enumerate(sched.timesteps(n_steps)):
            if max_steps is not None and k >= max_steps:
                break
            latents = denoise_step(sd, cfg, sched, latents, t, n_steps, inputs, stage, guidance_scale,
                                   image_guidance_scale)
            if trace is not None:
                trace.append(latents.clone())
    return latents


# ----------------------------------------------------------------------------------------------- training step (config 4)
#: parameters of modules whose name ends with "attn3" — the only trainable ones in stage 2 (train_StorySalon_stage2.py:170-177)
TRAINABLE_SUFFIXES = (".attn3.to_q.weight", ".attn3.to_k.weight", ".attn3.to_v.weight", ".attn3.to_out.0.weight",
                      ".attn3.to_out.0.bias")


def trainable_suffixes(module: str = "attn3") -> Tuple[str, ...]:
    """Parameter-name suffixes of the modules `name.endswith(module)` selects: "attn3" in stage 2 / COCO
    (train_StorySalon_stage2.py:170-177), "attn1" in stage 1 (train_StorySalon_stage1.py:175-179)."""
    return tuple(f".{module}.{leaf}" for leaf in ("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight", "to_out.0.bias"))


def ddpm_add_noise(sched: DDIM, x: Tensor, noise: Tensor, t: Tensor) -> Tensor:
    """DDPMScheduler.add_noise with a per-sample timestep vector (train_StorySalon_stage2.py:303,311)."""
    a = sched.alphas_cumprod[t.long()].view(-1, 1, 1, 1)
    return a.sqrt() * x + (1 - a).sqrt() * noise


def train_step(sd: SD, cfg: dict, batch: Dict[str, Tensor], use_refs=(0, 1, 2), trainable: str = "attn3",
               ref_levels: str = "stage2") -> Tuple[Tensor, Dict[str, Tensor]]:
    """Loss and attn3 gradients of one stage-2 training step, train_StorySalon_stage2.py:291-327 after its CLIP / VAE
    plumbing: `batch` holds latents [b,4,h,w], ref_latents [3,b,4,h,w], noise, ref_noise, timesteps [b] (int64), text
    [b,77,c], prev_text [3,b,77,c] and the already 1/8-downsampled mask [b,4,h,w] (:268-270).  `use_refs` replaces the
    random draw of :306-310 (p < 0.3 -> (0,1,2); 0.3 <= p < 0.6 -> (1,2); else (2,)).  The reference-frame noise level
    is ref_t * (3 - i) with the literal 3 of :311, whatever the number of frames used.
    Stage 1 (train_StorySalon_stage1.py:262-291) is the same step with use_refs=() — no reference pass, the main pass runs with
    image_hidden_states=None — and trainable="attn1".  train_COCO.py:286-316 is the stage-2 step with all three frames, every
    frame at the noise level ref_t (ref_levels="coco": no `* (3 - i)`, :303-304) and an unmasked loss (pass a zero mask, :315)."""
    if ref_levels not in ("stage2", "coco"):
        raise ValueError(ref_levels)
    suffixes = trainable_suffixes(trainable)
    params = {k: (v.detach().clone().requires_grad_(True) if k.endswith(suffixes) else v) for k, v in sd.items()}
    sched = DDIM()
    t = batch["timesteps"].long()
    ref_t = (batch["timesteps"] / 10).long()                                             # :297-300
    noisy = ddpm_add_noise(sched, batch["latents"], batch["noise"], t)                    # :303
    feats = []
    for i in use_refs:                                                                    # :309-314
        ti = ref_t * (3 - i) if ref_levels == "stage2" else ref_t
        x = ddpm_add_noise(sched, batch["ref_latents"][i], batch["ref_noise"], ti)
        feats.append(unet_forward(params, cfg, x, ti, batch["prev_text"][i], None)[1])
    ctx = {k: torch.cat([f[k] for f in feats], dim=1) for k in feats[0]} if feats else None   # :316-318 (stage 1: None, :288)
    pred = unet_forward(params, cfg, noisy, t, batch["text"], ctx)[0]                     # :322
    keep = 1.0 - batch["mask"]
    loss = F.mse_loss(pred.float() * keep, batch["noise"].float() * keep, reduction="mean")   # :325
    names = [k for k in params if k.endswith(suffixes)]
    grads = torch.autograd.grad(loss, [params[k] for k in names])                         # :327
    return loss.detach(), dict(zip(names, grads))
