"""Training-step building blocks on the HIP kernels: forward-with-saved-tensors and backward of one BasicTransformerBlock
and one ResnetBlock2D (BASELINE config 4, /root/reference/train_StorySalon_stage2.py:322-327).

STATUS: validated on MI355X in round 2.  The composition layer between the backward kernels (csrc/backward.hip,
csrc/attention_bwd.hip, the GroupNorm backward in csrc/norm.hip) and storygen_amd/train.py; tests/test_backward_gpu.py checks
both blocks against oracle/storygen_backward.py.  The op order follows that oracle line by line; tensors are allocated per call (no graph
capture yet) — this is a correctness-first layer.

Conventions: activations are [M, C] row-major with M = B * tokens; the residual stream and its gradients are fp32, MFMA
operands fp16 (same precision plan as the inference engine).  A linear layer y = x W^T has
  dgrad  dx = dy W        -> ops.gemm(dy, W^T stored as a [K, N] "weight")
  wgrad  dW = dy^T x      -> ops.gemm(dy^T, x^T) on ops.transpose outputs (contraction over the M tokens).
Only attn3's parameters train (train_StorySalon_stage2.py:170-177): their five gradients per block are returned in fp32.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops
from .repack import conv1x1_nk, conv3x3_krsc, interleave_geglu

F16, F32 = torch.float16, torch.float32


def _t(w: torch.Tensor) -> torch.Tensor:
    return w.t().contiguous()


def _e(*shape, dev, dtype=F16) -> torch.Tensor:
    return torch.empty(*shape, dtype=dtype, device=dev)


def _cast16(x32: torch.Tensor) -> torch.Tensor:
    """fp32 [M, C] -> fp16 copy (one bandwidth pass; GEMM A operands must be fp16)."""
    out = _e(*x32.shape, dev=x32.device)
    ops.copy_rows(out.unsqueeze(0), x32.unsqueeze(0))
    return out


def _tr(x: torch.Tensor) -> torch.Tensor:
    """[M, C] (fp16 / fp32) -> fp16 [C, M]."""
    return ops.transpose(x, _e(x.shape[1], x.shape[0], dev=x.device))


def _tr_batched(x: torch.Tensor) -> torch.Tensor:
    """[B, N, C] fp16 -> [B, C, Np] fp16 with Np = N rounded up to 8 (zero padded: the attention kernels read whole
    16-byte chunks of the key / query axis)."""
    B, N, C = x.shape
    Np = (N + 7) // 8 * 8
    if Np != N:
        xp = torch.zeros(B, Np, C, dtype=x.dtype, device=x.device)
        xp[:, :N] = x
        x = xp
    return ops.transpose_batched(x, _e(B, C, Np, dev=x.device))


class TransformerBlockTrain:
    """One BasicTransformerBlock (model/attention.py:131-302) in consume mode: attn1 (self), attn2 (text), attn3 (image
    context), GEGLU feed-forward.  `forward` keeps what `backward` needs."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, heads: int, device, trainable: str = "attn3"):
        """trainable: the module whose five weight gradients `backward` returns — "attn3" (stage 2 / COCO,
        train_StorySalon_stage2.py:170-177) or "attn1" (stage 1, train_StorySalon_stage1.py:175-179)."""
        if trainable not in ("attn1", "attn3"):
            raise NotImplementedError(f"weight gradients are built for attn1 and attn3, not {trainable!r}")
        self.dev, self.heads, p, self.trainable = torch.device(device), heads, prefix, trainable
        g = lambda k: sd[f"{p}.{k}"].detach().to(self.dev, F16).contiguous()     # noqa: E731
        self.ln = {n: (g(f"{n}.weight"), g(f"{n}.bias")) for n in ("norm1", "norm2", "norm3", "norm4")}
        self.w = {}
        for a in ("attn1", "attn2", "attn3"):
            for m in ("to_q", "to_k", "to_v"):
                self.w[f"{a}.{m}"] = g(f"{a}.{m}.weight")
            self.w[f"{a}.to_out"], self.w[f"{a}.b_out"] = g(f"{a}.to_out.0.weight"), g(f"{a}.to_out.0.bias")
        self.w_ff1, self.b_ff1 = interleave_geglu(g("ff.net.0.proj.weight"), g("ff.net.0.proj.bias"))
        self.w_ff2, self.b_ff2 = g("ff.net.2.weight"), g("ff.net.2.bias")
        # dgrad "weights": dx = dy W  ==  gemm(dy, W^T as an [in, out] matrix)
        self.wt = {k: _t(v) for k, v in self.w.items() if not k.endswith("b_out")}
        self.w_ff1_t, self.w_ff2_t = _t(self.w_ff1), _t(self.w_ff2)
        self.C = self.w["attn1.to_q"].shape[0]
        self.scale = (self.C // heads) ** -0.5
        self.saved: Optional[dict] = None

    def set_trainable(self, params: Dict[str, torch.Tensor]) -> None:
        """Refresh the device copies of the trainable module after an optimizer step.  `params`: to_q.weight, to_k.weight,
        to_v.weight, to_out.0.weight, to_out.0.bias (any dtype / device)."""
        a = self.trainable

        def put(table, key, value):
            # IN PLACE when the buffer exists: a captured training graph (UNetTrainer.train_step_graph) keeps reading these addresses
            if key in table and table[key].shape == value.shape:
                table[key].copy_(value)
            else:
                table[key] = value.contiguous()

        cp = lambda t: t.detach().to(self.dev, F16)                                  # noqa: E731
        for m in ("to_q", "to_k", "to_v"):
            put(self.w, f"{a}.{m}", cp(params[f"{m}.weight"]))
            put(self.wt, f"{a}.{m}", _t(self.w[f"{a}.{m}"]))
        put(self.w, f"{a}.to_out", cp(params["to_out.0.weight"]))
        put(self.w, f"{a}.b_out", cp(params["to_out.0.bias"]))
        put(self.wt, f"{a}.to_out", _t(self.w[f"{a}.to_out"]))

    set_attn3 = set_trainable          # round-1 name

    # ------------------------------------------------------------------------------------------------ forward
    def _attend(self, name: str, x16: torch.Tensor, kv16: torch.Tensor, B: int) -> dict:
        """q from x16 [B*Nq, C], k / v from kv16 [B*Nk, Ck]; returns the tensors the backward needs."""
        C, H, dev = self.C, self.heads, self.dev
        Mq, Mk = x16.shape[0], kv16.shape[0]
        Nq, Nk = Mq // B, Mk // B
        q, k, v = _e(Mq, C, dev=dev), _e(Mk, C, dev=dev), _e(Mk, C, dev=dev)
        ops.gemm(x16, self.w[f"{name}.to_q"], q)
        ops.gemm(kv16, self.w[f"{name}.to_k"], k)
        ops.gemm(kv16, self.w[f"{name}.to_v"], v)
        vt = _tr_batched(v.view(B, Nk, C))                                   # the forward kernel wants V^T [B, C, Nk]
        o = _e(Mq, C, dev=dev)
        lse = _e(B, H, Nq, dev=dev, dtype=F32)
        ops.attention_lse(q.view(B, Nq, C), k.view(B, Nk, C), vt, o.view(B, Nq, C), lse, H, self.scale, nk=Nk)
        return dict(q=q, k=k, v=v, o=o, lse=lse, Nq=Nq, Nk=Nk)

    def forward(self, h: torch.Tensor, text16: torch.Tensor, ctx16: Optional[torch.Tensor], B: int) -> torch.Tensor:
        """h fp32 [B*N, C]; text16 fp16 [B*S, 768]; ctx16 fp16 [B*Nc, C] (harvested features) or None = no image context
        (attention.py:279: the attn3 branch is skipped, stage 1).  Returns fp32 [B*N, C]."""
        C, dev, M = self.C, self.dev, h.shape[0]
        n1 = _e(M, C, dev=dev)
        ops.layernorm(h, *self.ln["norm1"], n1)
        a1 = self._attend("attn1", n1, n1, B)
        h1 = _e(M, C, dev=dev, dtype=F32)
        ops.gemm(a1["o"], self.w["attn1.to_out"], h1, bias=self.w["attn1.b_out"], res1=h)           # :250-262
        n2, n4, a3 = _e(M, C, dev=dev), None, None
        if ctx16 is not None:
            n4 = _e(M, C, dev=dev)
            ops.layernorm(h1, *self.ln["norm2"], n2, 1e-5, *self.ln["norm4"], n4)
        else:
            ops.layernorm(h1, *self.ln["norm2"], n2)
        a2 = self._attend("attn2", n2, text16, B)
        t = _e(M, C, dev=dev, dtype=F32)
        ops.gemm(a2["o"], self.w["attn2.to_out"], t, bias=self.w["attn2.b_out"], res1=h1)           # ht :266-277
        h3 = t                                                                                      # no image context: :295
        if ctx16 is not None:
            a3 = self._attend("attn3", n4, ctx16, B)
            h3 = _e(M, C, dev=dev, dtype=F32)
            ops.gemm(a3["o"], self.w["attn3.to_out"], h3, bias=self.w["attn3.b_out"], res1=t, res2=h1)  # ht + hi :281-293
        n3 = _e(M, C, dev=dev)
        ops.layernorm(h3, *self.ln["norm3"], n3)
        ffi = _e(M, 4 * C, dev=dev)
        ops.gemm(n3, self.w_ff1, ffi, bias=self.b_ff1, epilogue=ops.EPI_GEGLU)
        out = _e(M, C, dev=dev, dtype=F32)
        ops.gemm(ffi, self.w_ff2, out, bias=self.b_ff2, res1=h3)                                    # :298-300
        self.saved = dict(h=h, h1=h1, h3=h3, n1=n1, n2=n2, n4=n4, n3=n3, a1=a1, a2=a2, a3=a3, text=text16, ctx=ctx16, B=B)
        return out

    # ------------------------------------------------------------------------------------------------ backward
    def _attend_bwd(self, name: str, a: dict, do: torch.Tensor, B: int, need_kv: bool):
        """dq (token-major [Mq, C]) and, if need_kv, dK^T / dV^T ([B, C, Nk], keys contiguous)."""
        C, H, dev = self.C, self.heads, self.dev
        Nq, Nk = a["Nq"], a["Nk"]
        q3, k3, v3 = a["q"].view(B, Nq, C), a["k"].view(B, Nk, C), a["v"].view(B, Nk, C)
        do3 = do.view(B, Nq, C)
        ld2 = _e(B, H, Nq, 2, dev=dev, dtype=F32)
        ops.attention_bwd_prep(a["o"].view(B, Nq, C), do3, a["lse"], ld2, H)
        dq = _e(B * Nq, C, dev=dev)
        ops.attention_bwd_dq(q3, k3, _tr_batched(k3), v3, do3, ld2, dq.view(B, Nq, C), H, self.scale)
        if not need_kv:
            return dq, None, None
        dkt, dvt = _e(B, C, Nk, dev=dev), _e(B, C, Nk, dev=dev)
        ops.attention_bwd_dkv(q3, _tr_batched(q3), k3, v3, do3, _tr_batched(do3), ld2, dkt, dvt, H, self.scale)
        return dq, dkt, dvt

    def backward(self, dout: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """dout fp32 [M, C] -> (dh fp32 [M, C], {parameter name inside the trainable module: fp32 gradient}); oracle:
        transformer_block_bwd."""
        s, C, dev = self.saved, self.C, self.dev
        B, M = s["B"], dout.shape[0]
        # feed-forward :298-300
        dout16 = _cast16(dout)
        du = _e(M, 4 * C, dev=dev)
        ops.gemm(dout16, self.w_ff2_t, du)
        proj = _e(M, 8 * C, dev=dev)
        ops.gemm(s["n3"], self.w_ff1, proj, bias=self.b_ff1)                        # recomputed pre-activation (interleaved)
        dproj = ops.geglu_bwd(proj, du, _e(M, 8 * C, dev=dev))
        dn3 = _e(M, C, dev=dev)
        ops.gemm(dproj, self.w_ff1_t, dn3)
        dh3 = _e(M, C, dev=dev, dtype=F32)
        ops.layernorm_bwd(s["h3"], dn3, self.ln["norm3"][0], dh3, res=dout)
        dh3_16 = _cast16(dh3)
        grads: Dict[str, torch.Tensor] = {}
        dn4 = None
        if s["a3"] is not None:
            # image cross-attention (attn3): dgrad + (stage 2) the five weight gradients
            do3 = _e(M, C, dev=dev)
            ops.gemm(dh3_16, self.wt["attn3.to_out"], do3)
            want3 = self.trainable == "attn3"
            dq3, dk3t, dv3t = self._attend_bwd("attn3", s["a3"], do3, B, need_kv=want3)
            dn4 = _e(M, C, dev=dev)
            ops.gemm(dq3, self.wt["attn3.to_q"], dn4)
            if want3:
                grads = self._attn3_wgrads(s, dh3, dh3_16, dq3, dk3t, dv3t, B)
        # text cross-attention (attn2): only dq matters (text K/V and every attn2 weight are constants)
        do2 = _e(M, C, dev=dev)
        ops.gemm(dh3_16, self.wt["attn2.to_out"], do2)
        dq2, _, _ = self._attend_bwd("attn2", s["a2"], do2, B, need_kv=False)
        dn2 = _e(M, C, dev=dev)
        ops.gemm(dq2, self.wt["attn2.to_q"], dn2)
        # both branches add h1 back: dh1 = 2 dh3 + dLN2 + dLN4 (one kernel: the two LayerNorms share their input)
        dh1 = _e(M, C, dev=dev, dtype=F32)
        if dn4 is not None:
            ops.layernorm_bwd(s["h1"], dn2, self.ln["norm2"][0], dh1, dy2=dn4, g2=self.ln["norm4"][0], res=dh3, res_scale=2.0)
        else:                                                                       # no image branch: h3 = attn2(LN2(h1)) + h1
            ops.layernorm_bwd(s["h1"], dn2, self.ln["norm2"][0], dh1, res=dh3)
        dh1_16 = _cast16(dh1)
        # self-attention (attn1)
        do1 = _e(M, C, dev=dev)
        ops.gemm(dh1_16, self.wt["attn1.to_out"], do1)
        a1 = s["a1"]
        dq1, dk1t, dv1t = self._attend_bwd("attn1", a1, do1, B, need_kv=True)
        N = a1["Nk"]
        dk1, dv1 = _e(M, C, dev=dev), _e(M, C, dev=dev)
        for b in range(B):                                                          # [C, N] -> token-major [N, C]
            ops.transpose(dk1t[b], dk1[b * N:(b + 1) * N])
            ops.transpose(dv1t[b], dv1[b * N:(b + 1) * N])
        dn1 = _e(M, C, dev=dev, dtype=F32)
        ops.gemm(dq1, self.wt["attn1.to_q"], dn1)
        ops.gemm(dk1, self.wt["attn1.to_k"], dn1, res1=dn1)
        ops.gemm(dv1, self.wt["attn1.to_v"], dn1, res1=dn1)
        dh = _e(M, C, dev=dev, dtype=F32)
        ops.layernorm_bwd(s["h"], dn1, self.ln["norm1"][0], dh, res=dh1)
        if self.trainable == "attn1":
            grads = self._attn1_wgrads(s, dh1_16, dq1, dk1, dv1)
        return dh, grads

    def _attn1_wgrads(self, s, dh1_16, dq1, dk1, dv1) -> Dict[str, torch.Tensor]:
        """Stage 1: q, k and v of the self-attention are all projections of n1 = LayerNorm1(h), so with the token-major dq / dk / dv the
        backward walk already has, dW = d{q,k,v}^T n1; to_out as in _attn3_wgrads."""
        C, dev = self.C, self.dev
        f32 = lambda *sh: _e(*sh, dev=dev, dtype=F32)                                # noqa: E731
        n1_t = _tr(s["n1"])                                                          # [C, M]
        dh1_t = _tr(dh1_16)
        g = {"to_q.weight": ops.gemm(_tr(dq1), n1_t, f32(C, C)), "to_k.weight": ops.gemm(_tr(dk1), n1_t, f32(C, C)),
             "to_v.weight": ops.gemm(_tr(dv1), n1_t, f32(C, C)), "to_out.0.weight": ops.gemm(dh1_t, _tr(s["a1"]["o"]), f32(C, C))}
        ones = torch.ones(8, dh1_t.shape[1], dtype=F16, device=dev)
        g["to_out.0.bias"] = ops.gemm(dh1_t, ones, f32(C, 8))[:, 0].contiguous()
        return g

    def _attn3_wgrads(self, s, dh3, dh3_16, dq3, dk3t, dv3t, B) -> Dict[str, torch.Tensor]:
        C, dev = self.C, self.dev
        a3 = s["a3"]
        Nk = a3["Nk"]
        f32 = lambda *sh: _e(*sh, dev=dev, dtype=F32)                                # noqa: E731
        dh3_t = _tr(dh3_16)                                                          # [C, M]
        g = {}
        g["to_q.weight"] = ops.gemm(_tr(dq3), _tr(s["n4"]), f32(C, C))              # dq^T n4
        g["to_out.0.weight"] = ops.gemm(dh3_t, _tr(a3["o"]), f32(C, C))              # dy^T o
        ones = torch.ones(8, dh3_t.shape[1], dtype=F16, device=dev)                 # column sums as a GEMM against ones
        g["to_out.0.bias"] = ops.gemm(dh3_t, ones, f32(C, 8))[:, 0].contiguous()
        ctx_t = _tr_batched(s["ctx"].view(B, Nk, C))                                 # [B, C, Nk]
        dwk, dwv = f32(C, C), f32(C, C)
        for b in range(B):                                                          # sum over the batch, contraction over keys
            ops.gemm(dk3t[b], ctx_t[b][:, :Nk], dwk, res1=dwk if b else None)
            ops.gemm(dv3t[b], ctx_t[b][:, :Nk], dwv, res1=dwv if b else None)
        g["to_k.weight"], g["to_v.weight"] = dwk, dwv
        return g


class ResnetBlockTrain:
    """diffusers ResnetBlock2D with the time embedding as a per-(sample, channel) constant."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, groups: int, eps: float, device):
        self.dev, self.groups, self.eps, p = torch.device(device), groups, eps, prefix
        g = lambda k: sd[f"{p}.{k}"].detach().to(self.dev, F16).contiguous()     # noqa: E731
        self.n1, self.n2 = (g("norm1.weight"), g("norm1.bias")), (g("norm2.weight"), g("norm2.bias"))
        w1, w2 = sd[f"{p}.conv1.weight"].to(self.dev, F16), sd[f"{p}.conv2.weight"].to(self.dev, F16)
        self.w1, self.w2 = conv3x3_krsc(w1), conv3x3_krsc(w2)
        self.b1, self.b2 = g("conv1.bias"), g("conv2.bias")
        # dgrad weights: rotated by 180 degrees, in / out channels swapped (oracle conv_dgrad)
        self.w1_d = conv3x3_krsc(w1.flip(2, 3).transpose(0, 1).contiguous())
        self.w2_d = conv3x3_krsc(w2.flip(2, 3).transpose(0, 1).contiguous())
        self.cin, self.cout = w1.shape[1], w1.shape[0]
        self.wsc = self.wsc_t = self.bsc = None
        if f"{p}.conv_shortcut.weight" in sd:
            self.wsc = conv1x1_nk(sd[f"{p}.conv_shortcut.weight"].to(self.dev, F16))
            self.wsc_t, self.bsc = _t(self.wsc), g("conv_shortcut.bias")
        self.saved: Optional[dict] = None

    def forward(self, x: torch.Tensor, temb_proj: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
        """x fp32 [B*H*W, Cin]; temb_proj fp32 [B, Cout] = time_emb_proj(silu(emb)) (bias included).  Returns fp32 [B*H*W, Cout]."""
        dev, hw, M = self.dev, H * W, x.shape[0]
        ws = _e(ops.groupnorm_workspace_bytes(B, self.groups), dev=dev, dtype=torch.uint8)
        p_in = torch.zeros(B, H + 2, W + 2, self.cin, dtype=F16, device=dev)
        ops.groupnorm(x.view(B, hw, self.cin), *self.n1, p_in, self.groups, self.eps, True, ws)
        c1 = _e(M, self.cout, dev=dev, dtype=F32)
        rb = (temb_proj + self.b1.float()[None]).contiguous()
        ops.conv3x3(p_in, self.w1, c1.view(B, H, W, self.cout), rowbias=rb, x_padded=True)
        p_mid = torch.zeros(B, H + 2, W + 2, self.cout, dtype=F16, device=dev)
        ops.groupnorm(c1.view(B, hw, self.cout), *self.n2, p_mid, self.groups, self.eps, True, ws)
        res = x
        if self.wsc is not None:
            res = _e(M, self.cout, dev=dev, dtype=F32)
            ops.gemm(_cast16(x), self.wsc, res, bias=self.bsc)
        out = _e(M, self.cout, dev=dev, dtype=F32)
        ops.conv3x3(p_mid, self.w2, out.view(B, H, W, self.cout), bias=self.b2, res1=res.view(B, H, W, self.cout), x_padded=True)
        self.saved = dict(x=x, c1=c1, B=B, H=H, W=W)
        return out

    def backward(self, dout: torch.Tensor) -> torch.Tensor:
        """dout fp32 [M, Cout] -> dx fp32 [M, Cin]; oracle: resnet_block_bwd."""
        s, dev = self.saved, self.dev
        B, H, W = s["B"], s["H"], s["W"]
        hw, M = H * W, dout.shape[0]
        ws = _e(ops.groupnorm_bwd_workspace_bytes(B, self.groups), dev=dev, dtype=torch.uint8)
        pad_o = torch.zeros(B, H + 2, W + 2, self.cout, dtype=F16, device=dev)
        ops.pad_cast(dout.view(B, H, W, self.cout), pad_o)
        da2 = _e(M, self.cout, dev=dev)                                              # gradient w.r.t. silu(norm2(c1))
        ops.conv3x3(pad_o, self.w2_d, da2.view(B, H, W, self.cout), x_padded=True)
        ops.groupnorm_bwd(s["c1"].view(B, hw, self.cout), da2.view(B, hw, self.cout), *self.n2, pad_o, self.groups, self.eps, True, ws)
        da1 = _e(M, self.cin, dev=dev)                                               # gradient w.r.t. silu(norm1(x))
        ops.conv3x3(pad_o, self.w1_d, da1.view(B, H, W, self.cin), x_padded=True)
        if self.wsc is None:
            res = dout
        else:
            res = _e(M, self.cin, dev=dev, dtype=F32)
            ops.gemm(_cast16(dout), self.wsc_t, res)
        dx = _e(M, self.cin, dev=dev, dtype=F32)
        ops.groupnorm_bwd(s["x"].view(B, hw, self.cin), da1.view(B, hw, self.cin), *self.n1, dx.view(B, hw, self.cin), self.groups,
                          self.eps, True, ws, res=res.view(B, hw, self.cin))
        return dx
