"""The two networks either side of the denoising loop on the HIP kernels:

  ClipTextEngine   CLIPTextModel.forward        /root/reference/model/pipeline.py:137,183; train_StorySalon_stage2.py:283-302
  VaeEngine        AutoencoderKL.encode/decode  /root/reference/model/pipeline.py:198-205,392,401; train_StorySalon_stage2.py:281-288

Both are eager launch sequences over the UNet's kernels (sg_gemm_f16, sg_conv3x3_nhwc_f16, sg_groupnorm_nhwc_f16, sg_layernorm_f16,
sg_conv_in/out_f16, sg_pad_cast_f16) plus the five small kernels of csrc/encoders.hip; they run once per call of the pipeline, not per
denoising step, so they are not captured into graphs.  Activations are channels-last; the residual stream is fp32, every MFMA operand
fp16 — the UNet engine's conventions.  Parity: tests/test_encoders_gpu.py against oracle/encoders_oracle.py (CLIP pinned to
transformers, the VAE restated from diffusers 0.13.1).  There is no CPU path: importing this module loads libstorygen_hip.so.

Algebraic folds (exact in real arithmetic, done once in fp32 on the host):
  * encoder.conv_out followed by quant_conv (1x1, no padding in between) is ONE 3x3 convolution with weights Wq.Wc and bias
    Wq.bc + bq, run as two 4-channel sg_conv_out_f16 launches (mean | logvar);
  * post_quant_conv (1x1) followed by decoder.conv_in is ONE 3x3 convolution over the latent plus a constant-one channel that
    carries post_quant_conv's bias (the zero padding of conv_in applies to that channel too, so the border taps come out right);
  * AttentionBlock.value's bias moves into proj_attn's (softmax rows sum to one): bp' = Wp.bv + bp.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import ops, repack

F16, F32 = torch.float16, torch.float32
SD = Dict[str, torch.Tensor]


def _count(sd: SD, prefix: str) -> int:
    idx = {int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)}
    return 1 + max(idx) if idx else 0


# ================================================================================================================== CLIP
class ClipTextEngine:
    """CLIP text transformer (pre-LN, causal, quick_gelu) — transformers 4.27.4 CLIPTextTransformer."""

    def __init__(self, state_dict: SD, device, heads: int = 12, eps: float = 1e-5, hidden_act: str = "quick_gelu"):
        sd = {k[len("text_model."):] if k.startswith("text_model.") else k: v for k, v in state_dict.items()}
        if hidden_act not in ("quick_gelu", "gelu"):
            raise ValueError(f"ClipTextEngine: unsupported hidden_act {hidden_act!r}")
        self.dev = torch.device(device)
        self.heads, self.eps = heads, eps
        self.act = ops.ACT_QUICK_GELU if hidden_act == "quick_gelu" else ops.ACT_GELU

        def d32(name):
            return sd[name].detach().to(self.dev, F32).contiguous()

        def d16(t):
            return t.detach().to(self.dev, F32).to(F16).contiguous()

        self.tok, self.pos = d32("embeddings.token_embedding.weight"), d32("embeddings.position_embedding.weight")
        self.vocab, self.C = self.tok.shape
        if self.C % heads or self.C // heads > 64 or self.C % 8:
            raise ValueError(f"ClipTextEngine: hidden size {self.C} / {heads} heads is outside sg_attn_small_f16 (head dim <= 64)")
        self.layers = []
        for i in range(_count(sd, "encoder.layers.")):
            p = f"encoder.layers.{i}."
            a = p + "self_attn."
            self.layers.append(dict(
                ln1=(d16(sd[p + "layer_norm1.weight"]), d16(sd[p + "layer_norm1.bias"])),
                ln2=(d16(sd[p + "layer_norm2.weight"]), d16(sd[p + "layer_norm2.bias"])),
                wqkv=d16(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0)),
                bqkv=d16(torch.cat([sd[a + "q_proj.bias"], sd[a + "k_proj.bias"], sd[a + "v_proj.bias"]], 0)),
                wo=d16(sd[a + "out_proj.weight"]), bo=d16(sd[a + "out_proj.bias"]),
                w1=d16(sd[p + "mlp.fc1.weight"]), b1=d16(sd[p + "mlp.fc1.bias"]),
                w2=d16(sd[p + "mlp.fc2.weight"]), b2=d16(sd[p + "mlp.fc2.bias"])))
        self.lnf = (d16(sd["final_layer_norm.weight"]), d16(sd["final_layer_norm.bias"]))
        self.inner = self.layers[0]["w1"].shape[0]
        self.ws = ops.new_workspace(64 << 20, self.dev)

    def __call__(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """input_ids [B,T] -> (last_hidden_state fp32 [B,T,C], pooled fp32 [B,C]); T <= 128 and <= the position table."""
        if input_ids.dim() != 2:
            raise ValueError("ClipTextEngine: input_ids must be [B,T]")
        B, T = input_ids.shape
        if T > min(128, self.pos.shape[0]):
            raise ValueError(f"ClipTextEngine: sequence length {T} exceeds {min(128, self.pos.shape[0])}")
        ids_host = input_ids.detach().to("cpu", torch.int64)
        if int(ids_host.min()) < 0 or int(ids_host.max()) >= self.vocab:
            raise IndexError(f"ClipTextEngine: token id outside [0, {self.vocab})")
        ids = ids_host.to(self.dev).reshape(-1).contiguous()
        key_bias = None
        if attention_mask is not None:
            key_bias = ((1.0 - attention_mask.detach().to(self.dev, F32)) * torch.finfo(F32).min).contiguous()
        C, M, H = self.C, B * T, self.heads
        dev = self.dev
        x = torch.empty(M, C, dtype=F32, device=dev)
        x2 = torch.empty_like(x)
        h16 = torch.empty(M, C, dtype=F16, device=dev)
        qkv = torch.empty(M, 3 * C, dtype=F16, device=dev)
        a16 = torch.empty(M, C, dtype=F16, device=dev)
        u16 = torch.empty(M, self.inner, dtype=F16, device=dev)
        ops.embed_tokens(ids, self.tok, self.pos, x, T)
        q3 = qkv.view(B, T, 3 * C)
        scale = (C // H) ** -0.5
        for L in self.layers:
            ops.layernorm(x, L["ln1"][0], L["ln1"][1], h16, self.eps)
            ops.gemm(h16, L["wqkv"], qkv, bias=L["bqkv"], workspace=self.ws)
            ops.attention_small(q3[:, :, :C], q3[:, :, C:2 * C], q3[:, :, 2 * C:], a16.view(B, T, C), H, scale, True, key_bias)
            ops.gemm(a16, L["wo"], x2, bias=L["bo"], res1=x, workspace=self.ws)
            ops.layernorm(x2, L["ln2"][0], L["ln2"][1], h16, self.eps)
            ops.gemm(h16, L["w1"], u16, bias=L["b1"], workspace=self.ws)
            ops.act_rows(u16, self.act)
            ops.gemm(u16, L["w2"], x, bias=L["b2"], res1=x2, workspace=self.ws)
        ops.layernorm(x, self.lnf[0], self.lnf[1], h16, self.eps)
        hidden = h16.view(B, T, C).float()
        pooled = hidden[torch.arange(B, device=dev), ids_host.argmax(dim=-1).to(dev)]
        return hidden, pooled


# =================================================================================================================== VAE
class _Resnet:
    def __init__(self, sd: SD, p: str, d16):
        self.g1, self.b1 = d16(sd[p + "norm1.weight"]), d16(sd[p + "norm1.bias"])
        self.g2, self.b2 = d16(sd[p + "norm2.weight"]), d16(sd[p + "norm2.bias"])
        self.w1, self.c1 = d16(repack.conv3x3_krsc(sd[p + "conv1.weight"].float())), d16(sd[p + "conv1.bias"])
        self.w2, self.c2 = d16(repack.conv3x3_krsc(sd[p + "conv2.weight"].float())), d16(sd[p + "conv2.bias"])
        self.cin, self.cout = self.w1.shape[3], self.w1.shape[0]
        self.ws_, self.bs_ = None, None
        if p + "conv_shortcut.weight" in sd:
            self.ws_, self.bs_ = d16(repack.conv1x1_nk(sd[p + "conv_shortcut.weight"].float())), d16(sd[p + "conv_shortcut.bias"])


class _Attn:
    def __init__(self, sd: SD, p: str, d16):
        f = lambda n: sd[p + n].detach().float()   # noqa: E731
        self.g, self.b = d16(f("group_norm.weight")), d16(f("group_norm.bias"))
        self.wq, self.bq = d16(f("query.weight")), d16(f("query.bias"))
        self.wk, self.bk = d16(f("key.weight")), d16(f("key.bias"))
        self.wv = d16(f("value.weight"))
        self.wp = d16(f("proj_attn.weight"))
        self.bp = d16(f("proj_attn.weight") @ f("value.bias") + f("proj_attn.bias"))     # value bias folded (module docstring)
        self.C = self.wq.shape[0]


class VaeEngine:
    """AutoencoderKL (diffusers 0.13.1) forward passes: Encoder + quant_conv -> (mean, logvar); post_quant_conv + Decoder."""

    def __init__(self, state_dict: SD, device, groups: int = 32, eps: float = 1e-6):
        sd = state_dict
        self.dev = torch.device(device)
        self.groups, self.eps = groups, eps

        def d16(t):
            return t.detach().to(self.dev, F32).to(F16).contiguous()

        f = lambda n: sd[n].detach().float()   # noqa: E731
        # ---- encoder
        w = f("encoder.conv_in.weight")
        self.e_in_w, self.e_in_b = d16(repack.conv_in_kn(w)), d16(f("encoder.conv_in.bias"))
        self.in_channels = w.shape[1]
        self.e_down = []
        for i in range(_count(sd, "encoder.down_blocks.")):
            p = f"encoder.down_blocks.{i}."
            res = [_Resnet(sd, f"{p}resnets.{j}.", d16) for j in range(_count(sd, p + "resnets."))]
            down = None
            if p + "downsamplers.0.conv.weight" in sd:
                down = (d16(repack.conv3x3_krsc(f(p + "downsamplers.0.conv.weight"))), d16(f(p + "downsamplers.0.conv.bias")))
            self.e_down.append((res, down))
        self.e_mid = (_Resnet(sd, "encoder.mid_block.resnets.0.", d16), _Attn(sd, "encoder.mid_block.attentions.0.", d16),
                      _Resnet(sd, "encoder.mid_block.resnets.1.", d16))
        self.e_out_g, self.e_out_b = d16(f("encoder.conv_norm_out.weight")), d16(f("encoder.conv_norm_out.bias"))
        wq = f("quant_conv.weight").reshape(f("quant_conv.weight").shape[0], -1)               # [2L, 2L]
        wc = torch.einsum("ij,jcyx->icyx", wq, f("encoder.conv_out.weight"))                    # quant_conv folded into conv_out
        bc = wq @ f("encoder.conv_out.bias") + f("quant_conv.bias")
        self.latent = wc.shape[0] // 2
        if self.latent > 4:
            raise ValueError("VaeEngine: sg_conv_out_f16 handles at most 4 output channels per launch (latent_channels <= 4)")
        L = self.latent
        self.e_mean = (d16(repack.conv3x3_krsc(wc[:L])), self._bias16(bc[:L]))
        self.e_logvar = (d16(repack.conv3x3_krsc(wc[L:])), self._bias16(bc[L:]))
        # ---- decoder
        wp = f("post_quant_conv.weight").reshape(L, L)
        wi = f("decoder.conv_in.weight")                                                         # [C, L, 3, 3]
        wfold = torch.cat([torch.einsum("ojyx,ji->oiyx", wi, wp), torch.einsum("ojyx,j->oyx", wi, f("post_quant_conv.bias"))[:, None]], 1)
        self.d_in_w, self.d_in_b = d16(repack.conv_in_kn(wfold)), d16(f("decoder.conv_in.bias"))
        self.d_mid = (_Resnet(sd, "decoder.mid_block.resnets.0.", d16), _Attn(sd, "decoder.mid_block.attentions.0.", d16),
                      _Resnet(sd, "decoder.mid_block.resnets.1.", d16))
        self.d_up = []
        for i in range(_count(sd, "decoder.up_blocks.")):
            p = f"decoder.up_blocks.{i}."
            res = [_Resnet(sd, f"{p}resnets.{j}.", d16) for j in range(_count(sd, p + "resnets."))]
            up = None
            if p + "upsamplers.0.conv.weight" in sd:
                up = (d16(repack.conv3x3_krsc(f(p + "upsamplers.0.conv.weight"))), d16(f(p + "upsamplers.0.conv.bias")))
            self.d_up.append((res, up))
        self.d_out_g, self.d_out_b = d16(f("decoder.conv_norm_out.weight")), d16(f("decoder.conv_norm_out.bias"))
        wo = f("decoder.conv_out.weight")
        self.out_channels = wo.shape[0]
        if self.out_channels > 4:
            raise ValueError("VaeEngine: out_channels <= 4")
        self.d_out_w, self.d_out_bias = d16(repack.conv3x3_krsc(wo)), self._bias16(f("decoder.conv_out.bias"))
        self.downs = sum(1 for _, d in self.e_down if d is not None)
        self.ws = ops.new_workspace(256 << 20, self.dev)
        self._pads: Dict[tuple, torch.Tensor] = {}
        self._gnws: Dict[int, torch.Tensor] = {}

    def _bias16(self, b: torch.Tensor) -> torch.Tensor:
        """A bias of fewer than 8 entries in a 16-byte aligned, 8-entry fp16 buffer."""
        out = torch.zeros(8, dtype=F16, device=self.dev)
        out[:b.numel()] = b.to(self.dev, F32).to(F16)
        return out

    # ---------------------------------------------------------------------------------------------------- buffers
    def _pad(self, B: int, H: int, W: int, Cc: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(zero-bordered fp16 [B,H+2,W+2,C], the same storage shifted by one padded row and one pixel).  The border is never
        written, so one buffer per shape serves every layer; the slack after the last image backs the shifted view."""
        key = (B, H, W, Cc)
        if key not in self._pads:
            n, off = B * (H + 2) * (W + 2) * Cc, (W + 3) * Cc
            self._pads[key] = torch.zeros(n + off, dtype=F16, device=self.dev)
        flat = self._pads[key]
        n, off = B * (H + 2) * (W + 2) * Cc, (W + 3) * Cc
        return flat[:n].view(B, H + 2, W + 2, Cc), flat[off:off + n].view(B, H + 2, W + 2, Cc)

    def _gn(self, x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, out: torch.Tensor, silu: bool, xcopy=None) -> None:
        B = x.shape[0]
        if B not in self._gnws:
            self._gnws[B] = torch.empty(ops.groupnorm_workspace_bytes(B, self.groups), dtype=torch.uint8, device=self.dev)
        ops.groupnorm(x, g, b, out, self.groups, self.eps, silu, self._gnws[B], xcopy=xcopy)

    # ----------------------------------------------------------------------------------------------------- blocks
    def _resnet(self, r: _Resnet, x: torch.Tensor) -> torch.Tensor:
        """x fp32 [B,H,W,Cin] -> fp32 [B,H,W,Cout]: conv2(silu(norm2(conv1(silu(norm1(x)))))) + shortcut(x)."""
        B, H, W, _ = x.shape
        dev = self.dev
        pad1, _ = self._pad(B, H, W, r.cin)
        xc = torch.empty(B, H * W, r.cin, dtype=F16, device=dev) if r.ws_ is not None else None
        self._gn(x.view(B, H * W, r.cin), r.g1, r.b1, pad1, True, xcopy=xc)
        h = torch.empty(B, H, W, r.cout, dtype=F16, device=dev)
        ops.conv3x3(pad1, r.w1, h, bias=r.c1, x_padded=True, workspace=self.ws)
        pad2, _ = self._pad(B, H, W, r.cout)
        self._gn(h.view(B, H * W, r.cout), r.g2, r.b2, pad2, True)
        res = x
        if r.ws_ is not None:
            res = torch.empty(B, H, W, r.cout, dtype=F32, device=dev)
            ops.gemm(xc.view(B * H * W, r.cin), r.ws_, res.view(B * H * W, r.cout), bias=r.bs_, workspace=self.ws)
        out = torch.empty(B, H, W, r.cout, dtype=F32, device=dev)
        ops.conv3x3(pad2, r.w2, out, bias=r.c2, res1=res, x_padded=True, workspace=self.ws)
        return out

    def _attention(self, a: _Attn, x: torch.Tensor) -> torch.Tensor:
        """AttentionBlock with one head: per image, S = Q K^T (fp32), P = softmax(S / sqrt(C)) (fp16), O = P V, x + proj(O)."""
        B, H, W, Cc = x.shape
        N = H * W
        N8 = (N + 7) & ~7
        dev = self.dev
        h = torch.empty(B, N, Cc, dtype=F16, device=dev)
        self._gn(x.view(B, N, Cc), a.g, a.b, h, False)
        q = torch.empty(N, Cc, dtype=F16, device=dev)
        k = torch.zeros(N8, Cc, dtype=F16, device=dev)                  # rows >= N stay zero: their scores are never read
        vt = torch.zeros(Cc, N8, dtype=F16, device=dev)                 # columns >= N stay zero: they meet the zero tail of P
        s = torch.empty(N, N8, dtype=F32, device=dev)
        pr = torch.empty(N, N8, dtype=F16, device=dev)
        o = torch.empty(N, Cc, dtype=F16, device=dev)
        out = torch.empty(B, H, W, Cc, dtype=F32, device=dev)
        scale = 1.0 / math.sqrt(Cc)                                       # (1/sqrt(sqrt(C)))^2: q and k are each scaled once
        hp = torch.zeros(N8, Cc, dtype=F16, device=dev) if N8 != N else None      # V^T = Wv h^T needs a GEMM N that is a multiple of 8
        for b in range(B):
            hb = h[b]
            ops.gemm(hb, a.wq, q, bias=a.bq, workspace=self.ws)
            ops.gemm(hb, a.wk, k[:N], bias=a.bk, workspace=self.ws)
            if hp is not None:
                ops.copy_rows(hp[:N].unsqueeze(0), hb.unsqueeze(0))
            ops.gemm(a.wv, hb if hp is None else hp, vt, workspace=self.ws)
            ops.gemm(q, k, s, workspace=self.ws)
            ops.softmax_rows(s[:, :N], pr, scale)
            ops.gemm(pr, vt, o, workspace=self.ws)
            ops.gemm(o, a.wp, out[b].view(N, Cc), bias=a.bp, res1=x[b].view(N, Cc), workspace=self.ws)
        return out

    def _mid(self, mid, x: torch.Tensor) -> torch.Tensor:
        return self._resnet(mid[2], self._attention(mid[1], self._resnet(mid[0], x)))

    # ------------------------------------------------------------------------------------------------- public API
    def encode(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x [B, in_channels, H, W] (H, W multiples of 2^downsamples) -> (mean, logvar), fp32 NCHW [B, latent, H/8, W/8]."""
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"VaeEngine.encode: expected [B,{self.in_channels},H,W], got {tuple(x.shape)}")
        B, _, H, W = x.shape
        if H % (1 << self.downs) or W % (1 << self.downs):
            raise ValueError(f"VaeEngine.encode: H and W must be multiples of {1 << self.downs}")
        dev = self.dev
        x = x.detach().to(dev, F32).contiguous()
        h = torch.empty(B, H, W, self.e_in_b.numel(), dtype=F32, device=dev)
        ops.conv_in(x, self.e_in_w, self.e_in_b, h)
        for res, down in self.e_down:
            for r in res:
                h = self._resnet(r, h)
            if down is not None:
                # Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then a stride-2 3x3 convolution = the padded-input convolution read
                # from one row and one pixel further in
                Bh, Hh, Wh, Ch = h.shape
                pad, shifted = self._pad(Bh, Hh, Wh, Ch)
                ops.pad_cast(h, pad)
                h = torch.empty(Bh, Hh // 2, Wh // 2, Ch, dtype=F32, device=dev)
                ops.conv3x3(shifted, down[0], h, stride=2, bias=down[1], x_padded=True, workspace=self.ws)
        h = self._mid(self.e_mid, h)
        Bh, Hh, Wh, Ch = h.shape
        y = torch.empty(Bh, Hh * Wh, Ch, dtype=F16, device=dev)
        self._gn(h.view(Bh, Hh * Wh, Ch), self.e_out_g, self.e_out_b, y, True)
        mean = torch.empty(Bh, self.latent, Hh, Wh, dtype=F32, device=dev)
        logvar = torch.empty_like(mean)
        ops.conv_out(y.view(Bh, Hh, Wh, Ch), self.e_mean[0], self.e_mean[1], mean)
        ops.conv_out(y.view(Bh, Hh, Wh, Ch), self.e_logvar[0], self.e_logvar[1], logvar)
        return mean, logvar

    def sample(self, mean: torch.Tensor, logvar: torch.Tensor, noise: Optional[torch.Tensor], scale: float = 1.0) -> torch.Tensor:
        """DiagonalGaussianDistribution.sample() (noise = the standard-normal draw; None = mode()) times `scale`."""
        out = torch.empty_like(mean)
        if noise is not None:
            noise = noise.detach().to(self.dev, F32).contiguous()
        return ops.gaussian_sample(mean, logvar, noise, out, scale)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, latent, h, w] -> image fp32 NCHW [B, out_channels, 8h, 8w] (AutoencoderKL.decode(z).sample)."""
        if z.dim() != 4 or z.shape[1] != self.latent:
            raise ValueError(f"VaeEngine.decode: expected [B,{self.latent},h,w], got {tuple(z.shape)}")
        dev = self.dev
        B, L, H, W = z.shape
        z1 = torch.ones(B, L + 1, H, W, dtype=F32, device=dev)           # the constant-one channel carries post_quant_conv's bias
        z1[:, :L] = z.detach().to(dev, F32)
        h = torch.empty(B, H, W, self.d_in_b.numel(), dtype=F32, device=dev)
        ops.conv_in(z1, self.d_in_w, self.d_in_b, h)
        h = self._mid(self.d_mid, h)
        for res, up in self.d_up:
            for r in res:
                h = self._resnet(r, h)
            if up is not None:
                Bh, Hh, Wh, Ch = h.shape
                pad, _ = self._pad(Bh, Hh, Wh, Ch)
                ops.pad_cast(h, pad)
                h = torch.empty(Bh, 2 * Hh, 2 * Wh, Ch, dtype=F32, device=dev)
                ops.conv3x3(pad, up[0], h, upsample2x=True, bias=up[1], x_padded=True, workspace=self.ws)
        Bh, Hh, Wh, Ch = h.shape
        y = torch.empty(Bh, Hh * Wh, Ch, dtype=F16, device=dev)
        self._gn(h.view(Bh, Hh * Wh, Ch), self.d_out_g, self.d_out_b, y, True)
        img = torch.empty(Bh, self.out_channels, Hh, Wh, dtype=F32, device=dev)
        ops.conv_out(y.view(Bh, Hh, Wh, Ch), self.d_out_w, self.d_out_bias, img)
        return img


# ============================================================================================== parameter name / shape maps
def vae_param_shapes(block_out_channels=(128, 256, 512, 512), layers_per_block: int = 2, in_channels: int = 3, out_channels: int = 3,
                     latent_channels: int = 4) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of AutoencoderKL's state dict (diffusers 0.13.1) for the DownEncoderBlock2D / UpDecoderBlock2D layout."""
    out: Dict[str, Tuple[int, ...]] = {}

    def conv(n, co, ci, k):
        out[n + ".weight"], out[n + ".bias"] = (co, ci, k, k), (co,)

    def vec(n, c):
        out[n + ".weight"], out[n + ".bias"] = (c,), (c,)

    def resnet(p, ci, co):
        vec(p + "norm1", ci), conv(p + "conv1", co, ci, 3), vec(p + "norm2", co), conv(p + "conv2", co, co, 3)
        if ci != co:
            conv(p + "conv_shortcut", co, ci, 1)

    def mid(p, c):
        resnet(p + "resnets.0.", c, c)
        vec(p + "attentions.0.group_norm", c)
        for n in ("query", "key", "value", "proj_attn"):
            out[f"{p}attentions.0.{n}.weight"], out[f"{p}attentions.0.{n}.bias"] = (c, c), (c,)
        resnet(p + "resnets.1.", c, c)

    boc = tuple(block_out_channels)
    conv("encoder.conv_in", boc[0], in_channels, 3)
    ci = boc[0]
    for i, co in enumerate(boc):
        for j in range(layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", ci, co)
            ci = co
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    mid("encoder.mid_block.", boc[-1])
    vec("encoder.conv_norm_out", boc[-1]), conv("encoder.conv_out", 2 * latent_channels, boc[-1], 3)
    rev = boc[::-1]
    conv("decoder.conv_in", rev[0], latent_channels, 3)
    mid("decoder.mid_block.", rev[0])
    ci = rev[0]
    for i, co in enumerate(rev):
        for j in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", ci, co)
            ci = co
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    vec("decoder.conv_norm_out", rev[-1]), conv("decoder.conv_out", out_channels, rev[-1], 3)
    conv("quant_conv", 2 * latent_channels, 2 * latent_channels, 1)
    conv("post_quant_conv", latent_channels, latent_channels, 1)
    return out


def clip_text_param_shapes(vocab_size: int = 49408, hidden_size: int = 768, intermediate_size: int = 3072, num_hidden_layers: int = 12,
                           max_position_embeddings: int = 77) -> Dict[str, Tuple[int, ...]]:
    """Names (with transformers 4.x's `text_model.` prefix) and shapes of CLIPTextModel's parameters."""
    C, I = hidden_size, intermediate_size
    out = {"text_model.embeddings.token_embedding.weight": (vocab_size, C),
           "text_model.embeddings.position_embedding.weight": (max_position_embeddings, C)}
    for i in range(num_hidden_layers):
        p = f"text_model.encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            out[f"{p}self_attn.{n}.weight"], out[f"{p}self_attn.{n}.bias"] = (C, C), (C,)
        out[p + "layer_norm1.weight"], out[p + "layer_norm1.bias"] = (C,), (C,)
        out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = (I, C), (I,)
        out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = (C, I), (C,)
        out[p + "layer_norm2.weight"], out[p + "layer_norm2.bias"] = (C,), (C,)
    out["text_model.final_layer_norm.weight"], out["text_model.final_layer_norm.bias"] = (C,), (C,)
    return out


def init_state(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, embed_std: float = 0.02) -> SD:
    """PyTorch-default-like random initialisation of a name -> shape map (uniform +-1/sqrt(fan_in) for weights and biases of
    convolutions / linears, identity norms, N(0, embed_std) embedding tables) — there are no checkpoints on this machine."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}
    for name, shape in shapes.items():
        base = name.rsplit(".", 1)[0]
        if "embedding" in name:
            sd[name] = embed_std * torch.randn(shape, generator=g)
        elif len(shape) == 1 and (base + ".weight") in shapes and len(shapes[base + ".weight"]) == 1:      # a norm
            sd[name] = torch.ones(shape) if name.endswith(".weight") else torch.zeros(shape)
        else:
            wshape = shapes[base + ".weight"]
            bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return sd
