"""TEST INFRASTRUCTURE — explicit (autograd-free) backward pass of StoryGen's UNet for the stage-2 training step.

Why this exists: `storygen_oracle.train_step` restates /root/reference/train_StorySalon_stage2.py:291-327 and lets
torch.autograd do the differentiation, which pins WHAT the gradients are but not HOW a kernel library gets them.  A HIP
backward needs one kernel per formula below (GroupNorm / LayerNorm / SiLU / GEGLU backward, convolution dgrad as a
flipped-weight convolution, attention backward with recomputed probabilities, the attn3 weight gradients), so this file
writes every one of them out by hand, in the order and with the saved tensors the device engine will use, and
`tests/test_oracle_backward.py` checks (a) each formula against torch.autograd on random inputs and (b) the whole chain
against `train_step` (and so, through tests/golden/tiny_train.pt, against the reference's own autograd).

Only `tests/` may import this module (see oracle/README.md).  Nothing here is differentiated by autograd: every function runs
under torch.no_grad().

Gradient scope (train_StorySalon_stage2.py:170-177): only parameters of modules named `*attn3` train.  attn3 is not
evaluated in the reference passes (image_hidden_states is None there, attention.py:281), so no gradient reaches a
trainable parameter through the harvested features and the backward pass is the MAIN pass only: dgrad through every
layer from conv_out back to the first transformer block, weight gradients for the 16 x {to_q, to_k, to_v, to_out.0}.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import storygen_oracle as O

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------- leaf formulas
def silu_bwd(x: Tensor, dy: Tensor) -> Tensor:
    """d/dx [x sigmoid(x)] = s (1 + x (1 - s))."""
    s = torch.sigmoid(x)
    return dy * s * (1.0 + x * (1.0 - s))


def gelu_bwd(x: Tensor, dy: Tensor) -> Tensor:
    """Exact (erf) GELU, attention.py:385-388: d/dx [x Phi(x)] = Phi(x) + x phi(x)."""
    cdf = 0.5 * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))
    pdf = torch.exp(-0.5 * x * x) * (1.0 / math.sqrt(2.0 * math.pi))
    return dy * (cdf + x * pdf)


def _norm_bwd(xhat: Tensor, rstd: Tensor, g: Tensor) -> Tensor:
    """Shared by GroupNorm and LayerNorm; the last dim is the normalised one, g = dy * gamma."""
    return rstd * (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True))


def group_norm_bwd(x: Tensor, gamma: Tensor, dy: Tensor, groups: int, eps: float) -> Tensor:
    """dx of F.group_norm(x [B,C,H,W], groups, gamma, beta, eps) (the affine parameters are frozen)."""
    b, c = x.shape[:2]
    xg = x.reshape(b, groups, -1)
    mean = xg.mean(-1, keepdim=True)
    rstd = torch.rsqrt(xg.var(-1, unbiased=False, keepdim=True) + eps)
    g = (dy * gamma.view(1, c, 1, 1)).reshape(b, groups, -1)
    return _norm_bwd((xg - mean) * rstd, rstd, g).reshape(x.shape)


def layer_norm_bwd(x: Tensor, gamma: Tensor, dy: Tensor, eps: float = 1e-5) -> Tensor:
    mean = x.mean(-1, keepdim=True)
    rstd = torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + eps)
    return _norm_bwd((x - mean) * rstd, rstd, dy * gamma)


def linear_dgrad(dy: Tensor, w: Tensor) -> Tensor:
    """y = x W^T (+ b)  ->  dx = dy W.  On the device this is the forward GEMM kernel with the pre-transposed weight."""
    return dy @ w


def linear_wgrad(dy: Tensor, x: Tensor) -> Tensor:
    """dW[n, k] = sum_rows dy[row, n] x[row, k] (contraction over tokens)."""
    return dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1])


def conv_dgrad(dy: Tensor, w: Tensor, stride: int = 1) -> Tensor:
    """dx of F.conv2d(x, w [Cout,Cin,k,k], stride=stride, padding=k//2), written as the FORWARD convolution the device
    kernel can run: weights rotated by 180 degrees with in/out channels swapped; stride 2 first scatters dy onto the
    even positions of a zero image of the input's size (4x redundant arithmetic, on three small layers)."""
    k = w.shape[-1]
    wt = w.flip(2, 3).transpose(0, 1).contiguous()
    if stride == 1:
        return F.conv2d(dy, wt, padding=k // 2)
    assert stride == 2 and k == 3
    b, c, ho, wo = dy.shape
    z = dy.new_zeros(b, c, 2 * ho, 2 * wo)
    z[:, :, ::2, ::2] = dy
    return F.conv2d(z, wt, padding=1)


def upsample2x_bwd(du: Tensor) -> Tensor:
    """F.interpolate(x, scale_factor=2, mode="nearest") copies each pixel to a 2x2 block: dx = the block's sum."""
    b, c, h2, w2 = du.shape
    return du.reshape(b, c, h2 // 2, 2, w2 // 2, 2).sum(dim=(3, 5))


def attention_core(q: Tensor, k: Tensor, v: Tensor, heads: int) -> Tuple[Tensor, Tensor]:
    """softmax(q k^T / sqrt(d)) v per head; also returns the log-sum-exp rows the backward recomputes P from
    (what a flash-style forward kernel stores: one fp32 per query and head)."""
    b, nq, c = q.shape
    d = c // heads
    qh, kh, vh = (t.reshape(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    s = (qh @ kh.transpose(-1, -2)) * d ** -0.5
    lse = torch.logsumexp(s, dim=-1)
    o = torch.exp(s - lse[..., None]) @ vh
    return o.transpose(1, 2).reshape(b, nq, c), lse


def attention_core_bwd(q: Tensor, k: Tensor, v: Tensor, o: Tensor, lse: Tensor, do: Tensor, heads: int
                       ) -> Tuple[Tensor, Tensor, Tensor]:
    """Flash-attention backward: P = exp(S - lse) recomputed, D = rowsum(dO * O),
    dV = P^T dO, dP = dO V^T, dS = P * (dP - D), dQ = dS K / sqrt(d), dK = dS^T Q / sqrt(d)."""
    b, nq, c = q.shape
    d = c // heads
    scale = d ** -0.5
    qh, kh, vh, oh, doh = (t.reshape(b, -1, heads, d).transpose(1, 2) for t in (q, k, v, o, do))
    p = torch.exp((qh @ kh.transpose(-1, -2)) * scale - lse[..., None])
    delta = (doh * oh).sum(-1, keepdim=True)
    dv = p.transpose(-1, -2) @ doh
    ds = p * (doh @ vh.transpose(-1, -2) - delta)
    dq = (ds @ kh) * scale
    dk = (ds.transpose(-1, -2) @ qh) * scale
    back = lambda t: t.transpose(1, 2).reshape(b, -1, c)   # noqa: E731
    return back(dq), back(dk), back(dv)


# ----------------------------------------------------------------------------------------------- modules
def attention_module_bwd(sd: SD, p: str, x: Tensor, ctx: Optional[Tensor], heads: int, dy: Tensor, wgrad: bool
                         ) -> Tuple[Tensor, Dict[str, Tensor]]:
    """Backward of storygen_oracle.attention (CrossAttention: to_q/to_k/to_v without bias, to_out.0 with bias).
    Returns dx (w.r.t. the query-side input; for self-attention the K/V paths are included) and, if `wgrad`, the five
    parameter gradients keyed like the state dict.  No gradient is returned for `ctx` (text embeddings and harvested
    features are constants of the training step)."""
    wq, wk, wv = sd[f"{p}.to_q.weight"], sd[f"{p}.to_k.weight"], sd[f"{p}.to_v.weight"]
    wo = sd[f"{p}.to_out.0.weight"]
    c = x if ctx is None else ctx
    q, k, v = x @ wq.t(), c @ wk.t(), c @ wv.t()
    o, lse = attention_core(q, k, v, heads)
    do = linear_dgrad(dy, wo)
    dq, dk, dv = attention_core_bwd(q, k, v, o, lse, do, heads)
    dx = linear_dgrad(dq, wq)
    if ctx is None:
        dx = dx + linear_dgrad(dk, wk) + linear_dgrad(dv, wv)
    grads: Dict[str, Tensor] = {}
    if wgrad:
        grads = {f"{p}.to_q.weight": linear_wgrad(dq, x), f"{p}.to_k.weight": linear_wgrad(dk, c),
                 f"{p}.to_v.weight": linear_wgrad(dv, c), f"{p}.to_out.0.weight": linear_wgrad(dy, o),
                 f"{p}.to_out.0.bias": dy.reshape(-1, dy.shape[-1]).sum(0)}
    return dx, grads


def transformer_block_bwd(sd: SD, p: str, h: Tensor, text: Tensor, image_ctx: Tensor, heads: int, dout: Tensor
                          ) -> Tuple[Tensor, Dict[str, Tensor]]:
    """Backward of storygen_oracle.transformer_block (attention.py:236-302) in consume mode.  Saved: the block input h;
    everything else is recomputed here (the device engine keeps h1, h3 and the LayerNorm outputs instead)."""
    c = h.shape[-1]

    def ln(n, t):
        return F.layer_norm(t, (c,), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], 1e-5)

    def ln_bwd(n, t, d):
        return layer_norm_bwd(t, sd[f"{p}.{n}.weight"], d)

    # forward recompute
    n1 = ln("norm1", h)
    h1 = O.attention(sd, f"{p}.attn1", n1, None, heads) + h
    n2, n4 = ln("norm2", h1), ln("norm4", h1)
    ht = O.attention(sd, f"{p}.attn2", n2, text, heads) + h1
    hi = O.attention(sd, f"{p}.attn3", n4, image_ctx, heads) + h1
    h3 = ht + hi
    n3 = ln("norm3", h3)
    w1, w2 = sd[f"{p}.ff.net.0.proj.weight"], sd[f"{p}.ff.net.2.weight"]
    val, gate = F.linear(n3, w1, sd[f"{p}.ff.net.0.proj.bias"]).chunk(2, dim=-1)
    # backward: feed-forward (:298-300)
    du = linear_dgrad(dout, w2)
    dproj = torch.cat([du * F.gelu(gate), gelu_bwd(gate, du * val)], dim=-1)
    dh3 = dout + ln_bwd("norm3", h3, linear_dgrad(dproj, w1))
    # the two cross-attention branches both start from h1 and both add it back (:277,291-293): dh1 gets 2 x dh3 directly
    dn4, grads = attention_module_bwd(sd, f"{p}.attn3", n4, image_ctx, heads, dh3, wgrad=True)
    dn2, _ = attention_module_bwd(sd, f"{p}.attn2", n2, text, heads, dh3, wgrad=False)
    dh1 = 2.0 * dh3 + ln_bwd("norm4", h1, dn4) + ln_bwd("norm2", h1, dn2)
    # self-attention (:250-262)
    dn1, _ = attention_module_bwd(sd, f"{p}.attn1", n1, None, heads, dh1, wgrad=False)
    return dh1 + ln_bwd("norm1", h, dn1), grads


def transformer_2d_bwd(sd: SD, p: str, x: Tensor, text: Tensor, image_ctx: Tensor, heads: int, groups: int, dout: Tensor
                       ) -> Tuple[Tensor, Dict[str, Tensor]]:
    """Backward of storygen_oracle.transformer_2d (attention.py:85-128); saved: the module input x."""
    b, c, hh, ww = x.shape
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(b, hh * ww, c)        # noqa: E731
    img = lambda t: t.reshape(b, hh, ww, c).permute(0, 3, 1, 2)         # noqa: E731
    g = F.group_norm(x, groups, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"], 1e-6)
    h = tok(F.conv2d(g, sd[f"{p}.proj_in.weight"], sd[f"{p}.proj_in.bias"]))
    dtok = tok(conv_dgrad(dout, sd[f"{p}.proj_out.weight"]))
    dh, grads = transformer_block_bwd(sd, f"{p}.transformer_blocks.0", h, text, image_ctx, heads, dtok)
    dg = conv_dgrad(img(dh), sd[f"{p}.proj_in.weight"])
    return dout + group_norm_bwd(x, sd[f"{p}.norm.weight"], dg, groups, 1e-6), grads


def resnet_block_bwd(sd: SD, p: str, x: Tensor, emb: Tensor, groups: int, eps: float, dout: Tensor) -> Tensor:
    """Backward of storygen_oracle.resnet_block (diffusers 0.13.1 ResnetBlock2D, output_scale_factor 1); saved: x.
    The time-embedding branch is a per-(sample, channel) constant of the step: no gradient is propagated into it."""
    n1 = F.group_norm(x, groups, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps)
    c1 = F.conv2d(F.silu(n1), sd[f"{p}.conv1.weight"], sd[f"{p}.conv1.bias"], padding=1)
    c1 = c1 + F.linear(F.silu(emb), sd[f"{p}.time_emb_proj.weight"], sd[f"{p}.time_emb_proj.bias"])[:, :, None, None]
    n2 = F.group_norm(c1, groups, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps)
    dn2 = silu_bwd(n2, conv_dgrad(dout, sd[f"{p}.conv2.weight"]))
    dc1 = group_norm_bwd(c1, sd[f"{p}.norm2.weight"], dn2, groups, eps)
    dn1 = silu_bwd(n1, conv_dgrad(dc1, sd[f"{p}.conv1.weight"]))
    dx = group_norm_bwd(x, sd[f"{p}.norm1.weight"], dn1, groups, eps)
    sc = f"{p}.conv_shortcut.weight"
    return dx + (conv_dgrad(dout, sd[sc]) if sc in sd else dout)


# ----------------------------------------------------------------------------------------------- the UNet
def unet_forward_saving(sd: SD, cfg: dict, sample: Tensor, timestep, text: Tensor, ctx: Dict[str, Tensor]
                        ) -> Tuple[Tensor, list]:
    """storygen_oracle.unet_forward in consume mode, recording the tape the backward walks: one entry per module, with
    the module's input (what the device engine saves per layer)."""
    boc = list(cfg["block_out_channels"])
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    ahd = cfg["attention_head_dim"]
    heads = list(ahd) if isinstance(ahd, (list, tuple)) else [ahd] * len(boc)
    lpb = cfg["layers_per_block"]
    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
    t = t.reshape(-1).expand(sample.shape[0])
    emb = O.timestep_embedding(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    tape: list = []

    def resnet(p, x):
        tape.append(("resnet", p, x))
        return O.resnet_block(sd, p, x, emb, groups, eps)

    def xf(p, x, key, nh):
        tape.append(("xf", p, x, ctx[key], nh))
        return O.transformer_2d(sd, p, x, text, ctx[key], nh, groups)[0]

    h = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    tape.append(("conv_in",))
    skips: List[Tensor] = [h]
    for i, typ in enumerate(cfg["down_block_types"]):
        for j in range(lpb):
            h = resnet(f"down_blocks.{i}.resnets.{j}", h)
            if typ == "CrossAttnDownBlock2D":
                h = xf(f"down_blocks.{i}.attentions.{j}", h, f"down_{i + 1}_{j + 1}", heads[i])
            skips.append(h)
            tape.append(("skip_push",))
        if i != len(boc) - 1:
            w = f"down_blocks.{i}.downsamplers.0.conv"
            tape.append(("down", w))
            h = F.conv2d(h, sd[f"{w}.weight"], sd[f"{w}.bias"], stride=2, padding=cfg["downsample_padding"])
            skips.append(h)
            tape.append(("skip_push",))
    h = resnet("mid_block.resnets.0", h)
    h = xf("mid_block.attentions.0", h, "mid", heads[-1])
    h = resnet("mid_block.resnets.1", h)
    rheads = list(reversed(heads))
    for i, typ in enumerate(cfg["up_block_types"]):
        for j in range(lpb + 1):
            s = skips.pop()
            tape.append(("cat", h.shape[1]))
            h = torch.cat([h, s], dim=1)
            h = resnet(f"up_blocks.{i}.resnets.{j}", h)
            if typ == "CrossAttnUpBlock2D":
                h = xf(f"up_blocks.{i}.attentions.{j}", h, f"up_{i}_{j + 1}", rheads[i])
        if i != len(boc) - 1:
            w = f"up_blocks.{i}.upsamplers.0.conv"
            tape.append(("up", w))
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[f"{w}.weight"], sd[f"{w}.bias"], padding=1)
    tape.append(("out", h))
    h = F.silu(F.group_norm(h, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps))
    h = F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return h, [emb, text] + tape


def unet_backward(sd: SD, cfg: dict, saved: list, d_eps: Tensor) -> Dict[str, Tensor]:
    """Walks the tape of unet_forward_saving backwards from d(loss)/d(epsilon); returns the 80 attn3 gradients.
    Skip connections: the gradient of a concatenated skip is parked until the walk reaches the layer that pushed it."""
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    emb, text, tape = saved[0], saved[1], saved[2:]
    grads: Dict[str, Tensor] = {}
    pending: List[Tensor] = []          # skip gradients parked by the concats; the up path pops skips in reverse push order and
                                        # this walk visits the concats in reverse again, so the LAST parked = the LAST pushed
    dh: Optional[Tensor] = d_eps
    for rec in reversed(tape):
        kind = rec[0]
        if kind == "out":
            x = rec[1]
            n = F.group_norm(x, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps)
            dn = silu_bwd(n, conv_dgrad(dh, sd["conv_out.weight"]))
            dh = group_norm_bwd(x, sd["conv_norm_out.weight"], dn, groups, eps)
        elif kind == "up":
            dh = upsample2x_bwd(conv_dgrad(dh, sd[f"{rec[1]}.weight"]))
        elif kind == "cat":
            c_h = rec[1]
            pending.append(dh[:, c_h:])
            dh = dh[:, :c_h]
        elif kind == "resnet":
            dh = resnet_block_bwd(sd, rec[1], rec[2], emb, groups, eps, dh)
        elif kind == "xf":
            _, p, x, ictx, nh = rec
            dh, g = transformer_2d_bwd(sd, p, x, text, ictx, nh, groups, dh)
            grads.update(g)
            if len(grads) == 5 * sum(1 for r in tape if r[0] == "xf"):
                break                                  # the first transformer block of the network: nothing trainable below it
        elif kind == "skip_push":
            # the tensor pushed here also continued down the network: its gradient = what came back along the main
            # path (dh) + what its concat consumer parked
            dh = dh + pending.pop()
        elif kind == "down":
            dh = conv_dgrad(dh, sd[f"{rec[1]}.weight"], stride=2)
        elif kind == "conv_in":
            break
    return grads


def train_step_explicit(sd: SD, cfg: dict, batch: Dict[str, Tensor], use_refs=(0, 1, 2)) -> Tuple[Tensor, Dict[str, Tensor]]:
    """storygen_oracle.train_step with the hand-written backward: same loss, same 80 gradients, no autograd."""
    with torch.no_grad():
        sched = O.DDIM()
        t = batch["timesteps"].long()
        ref_t = (batch["timesteps"] / 10).long()
        noisy = O.ddpm_add_noise(sched, batch["latents"], batch["noise"], t)
        feats = []
        for i in use_refs:
            ti = ref_t * (3 - i)
            x = O.ddpm_add_noise(sched, batch["ref_latents"][i], batch["ref_noise"], ti)
            feats.append(O.unet_forward(sd, cfg, x, ti, batch["prev_text"][i], None)[1])
        ctx = {k: torch.cat([f[k] for f in feats], dim=1) for k in feats[0]}
        pred, saved = unet_forward_saving(sd, cfg, noisy, t, batch["text"], ctx)
        keep = 1.0 - batch["mask"]
        diff = pred.float() * keep - batch["noise"].float() * keep
        loss = (diff * diff).mean()
        d_pred = 2.0 * diff * keep / diff.numel()
        return loss, unet_backward(sd, cfg, saved, d_pred)
