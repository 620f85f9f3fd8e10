"""HipCrossAttnProcessor — the HIP attention path as a diffusers attention processor (SURVEY §8b's operator-level plug-in).

The reference's three attention modules per block (`attn1`, `attn2`, `attn3`, /root/reference/model/attention.py:175-223)
are `diffusers.models.cross_attention.CrossAttention` objects, whose arithmetic is delegated to a processor object called as
`processor(attn, hidden_states, encoder_hidden_states=None, attention_mask=None)` (diffusers 0.13.1
`CrossAttention.forward` -> `CrossAttnProcessor.__call__`; call sites attention.py:255-260,271-276,285-290).  Installing

    for m in unet.modules():
        if isinstance(m, CrossAttention):
            m.set_processor(HipCrossAttnProcessor())

on the reference's own (PyTorch) UNet swaps just the projections + softmax(QK^T)V core of every attention for the HIP
kernels (sg_gemm_f16 for q / k / V^T / out, sg_attn_fwd_f16 for the core) and leaves the rest of the model on torch: the
smallest possible adoption step.  Same contract as the default processor: returns `to_out[1](to_out[0](attention))` with the
input's dtype and shape; `attention_mask` is not supported (the StoryGen path never passes one) and raises.
Inputs must be fp16 tensors on the HIP device; the projection weights are read from the module at call time (no cache)."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops


class HipCrossAttnProcessor:
    def __init__(self, workspace_mb: int = 64):
        self._ws: Optional[torch.Tensor] = None
        self._ws_mb = workspace_mb

    def _workspace(self, dev) -> torch.Tensor:
        if self._ws is None or self._ws.device != dev:
            self._ws = torch.empty(self._ws_mb << 20, dtype=torch.uint8, device=dev)
        return self._ws

    @torch.no_grad()
    def __call__(self, attn, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        if attention_mask is not None:
            raise NotImplementedError("HipCrossAttnProcessor: attention_mask is not supported (StoryGen never passes one)")
        h = hidden_states
        if h.dtype != torch.float16 or not h.is_cuda:
            raise TypeError("HipCrossAttnProcessor needs fp16 hidden states on the HIP device (there is no CPU path)")
        ctx = h if encoder_hidden_states is None else encoder_hidden_states.to(h.dtype)
        B, Nq, Cq = h.shape
        Nk, Ck = ctx.shape[1], ctx.shape[2]
        inner = attn.to_q.weight.shape[0]
        heads = attn.heads
        if Nk % 8:                                   # V^T rows must be 16-byte aligned: pad the context with zero tokens, mask by nk
            pad = 8 - Nk % 8
            ctx = torch.cat([ctx, ctx.new_zeros(B, pad, Ck)], dim=1)
        Nkp = ctx.shape[1]
        dev, ws = h.device, self._workspace(h.device)
        f16 = dict(dtype=torch.float16, device=dev)
        wq, wk, wv = (w.detach().to(torch.float16).contiguous() for w in (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight))
        bq, bk, bv = (None if lin.bias is None else lin.bias.detach().to(torch.float16).contiguous()
                      for lin in (attn.to_q, attn.to_k, attn.to_v))
        h2, c2 = h.reshape(B * Nq, Cq).contiguous(), ctx.reshape(B * Nkp, Ck).contiguous()
        q, k = torch.empty(B * Nq, inner, **f16), torch.empty(B * Nkp, inner, **f16)
        vt = torch.empty(inner, B * Nkp, **f16)
        ops.gemm(h2, wq, q, bias=bq, workspace=ws)
        ops.gemm(c2, wk, k, bias=bk, workspace=ws)
        ops.gemm(wv, c2, vt, workspace=ws)                                   # V^T[C, B*Nk] = Wv . ctx^T (the kernel's operand layout)
        if bv is not None:
            vt += bv[:, None]
        out = torch.empty(B, Nq, inner, **f16)
        ops.attention(q.view(B, Nq, inner), k.view(B, Nkp, inner), vt.view(inner, B, Nkp).permute(1, 0, 2), out, heads,
                      float(attn.scale), nk=Nk)
        lin = attn.to_out[0]
        y = torch.empty(B * Nq, lin.weight.shape[0], **f16)
        ops.gemm(out.view(B * Nq, inner), lin.weight.detach().to(torch.float16).contiguous(), y,
                 bias=None if lin.bias is None else lin.bias.detach().to(torch.float16).contiguous(), workspace=ws)
        return attn.to_out[1](y.view(B, Nq, -1))
