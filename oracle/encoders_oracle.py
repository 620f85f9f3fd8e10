"""TEST INFRASTRUCTURE — CPU restatement (torch fp32, functional, driven by plain state dicts) of the two networks on either side
of the denoising loop: the CLIP text encoder and the AutoencoderKL.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (storygen_amd/encoders.py) never does.

The reference calls them at model/pipeline.py:137 and :183 (`self.text_encoder(ids, attention_mask=...)[0]`), :392 and :401
(`self.vae.encode(x).latent_dist.sample()`) and :198-205 (`self.vae.decode(latents / 0.18215).sample`), and
train_StorySalon_stage2.py:281-302.  The arithmetic lives in third-party packages that are absent from /root/reference:

  * transformers==4.27.4 `CLIPTextModel` (environment.yaml) — PINNED: `clip_text_forward` is checked against the installed
    transformers' CLIPTextModel on random weights by oracle/make_golden_encoders.py, which also writes the committed fixture
    tests/golden/clip_text_tiny.pt (tests/test_oracle_golden.py::test_clip_text_oracle_matches_transformers_golden).
  * diffusers==0.13.1 `AutoencoderKL` (models/autoencoder_kl.py, vae.py, unet_2d_blocks.py, resnet.py, attention.py of that
    release) — PARITY UNPINNED: diffusers is not installable here, so `vae_encode_moments` / `vae_decode` restate its published
    algorithm (state-dict names included: `mid_block.attentions.0.{group_norm,query,key,value,proj_attn}`) and are anchored
    only by structural checks (shapes, the reference's VAE config, encode/decode consistency).
"""
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------------------------------ CLIP
def _strip(sd: SD, prefix: str) -> SD:
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def clip_text_state(sd: SD) -> SD:
    """transformers 4.x names everything `text_model.*`; newer releases drop the prefix.  Returns the un-prefixed dict."""
    return _strip(sd, "text_model.") if any(k.startswith("text_model.") for k in sd) else dict(sd)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def clip_text_forward(sd: SD, input_ids: torch.Tensor, heads: int, eps: float = 1e-5, hidden_act: str = "quick_gelu",
                      attention_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """CLIPTextTransformer.forward (transformers 4.27.4 modeling_clip.py): token + position embeddings, `layers` pre-LN
    blocks with a causal additive mask (and the optional padding mask), final_layer_norm; pooled = the hidden state at the
    position of the largest token id (the EOS token).  Returns (last_hidden_state [B,T,C], pooled [B,C])."""
    sd = {k: v.float() for k, v in clip_text_state(sd).items()}
    B, T = input_ids.shape
    x = sd["embeddings.token_embedding.weight"][input_ids] + sd["embeddings.position_embedding.weight"][:T][None]
    C = x.shape[-1]
    D = C // heads
    mask = torch.full((T, T), float("-inf")).triu(1)[None, None]
    if attention_mask is not None:
        mask = mask + (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    act = quick_gelu if hidden_act == "quick_gelu" else (lambda v: F.gelu(v))
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (C,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]) * D ** -0.5
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, T, heads, D).transpose(1, 2) for t in (q, k, v))
        w = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)
        a = (w @ v).transpose(1, 2).reshape(B, T, C)
        x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (C,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
        h = F.linear(act(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = x + h
    x = F.layer_norm(x, (C,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], eps)
    pooled = x[torch.arange(B), input_ids.argmax(dim=-1)]
    return x, pooled


# ------------------------------------------------------------------------------------------------------------------- VAE
def _gn(x, sd, name, groups, eps):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def _conv(x, sd, name, stride=1, padding=1):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _resnet(x, sd, p, groups, eps=1e-6):
    """ResnetBlock2D with temb_channels=None, dropout 0, output_scale_factor 1 (diffusers 0.13.1 resnet.py)."""
    h = _conv(F.silu(_gn(x, sd, p + "norm1", groups, eps)), sd, p + "conv1")
    h = _conv(F.silu(_gn(h, sd, p + "norm2", groups, eps)), sd, p + "conv2")
    if p + "conv_shortcut.weight" in sd:
        x = _conv(x, sd, p + "conv_shortcut", padding=0)
    return x + h


def _attention_block(x, sd, p, groups, eps=1e-6):
    """AttentionBlock (diffusers 0.13.1 models/attention.py): one head when num_head_channels is None;
    q and k are each scaled by 1/sqrt(sqrt(C/heads)); softmax in fp32; (proj_attn(out) + residual) / rescale_output_factor(=1)."""
    B, C, H, W = x.shape
    h = _gn(x, sd, p + "group_norm", groups, eps).view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, sd[p + "query.weight"], sd[p + "query.bias"])
    k = F.linear(h, sd[p + "key.weight"], sd[p + "key.bias"])
    v = F.linear(h, sd[p + "value.weight"], sd[p + "value.bias"])
    scale = 1.0 / math.sqrt(math.sqrt(C / 1))
    w = torch.softmax((q * scale) @ (k * scale).transpose(-1, -2), dim=-1)
    h = F.linear(w @ v, sd[p + "proj_attn.weight"], sd[p + "proj_attn.bias"])
    return h.transpose(1, 2).reshape(B, C, H, W) + x


def _mid(x, sd, p, groups):
    x = _resnet(x, sd, p + "resnets.0.", groups)
    x = _attention_block(x, sd, p + "attentions.0.", groups)
    return _resnet(x, sd, p + "resnets.1.", groups)


def _count(sd, prefix):
    idx = {int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)}
    return 1 + max(idx) if idx else 0


def vae_encode_moments(sd: SD, x: torch.Tensor, groups: int = 32) -> torch.Tensor:
    """AutoencoderKL.encode up to the moments: Encoder (conv_in, DownEncoderBlock2D x n [resnets, Downsample2D(padding=0):
    F.pad (0,1,0,1) + 3x3 stride-2 conv], UNetMidBlock2D, GroupNorm(eps 1e-6) + SiLU + conv_out) then quant_conv.
    Returns [B, 2*latent, h, w] = (mean | logvar)."""
    sd = {k: v.float() for k, v in sd.items()}
    h = _conv(x.float(), sd, "encoder.conv_in")
    for i in range(_count(sd, "encoder.down_blocks.")):
        p = f"encoder.down_blocks.{i}."
        for j in range(_count(sd, p + "resnets.")):
            h = _resnet(h, sd, f"{p}resnets.{j}.", groups)
        if p + "downsamplers.0.conv.weight" in sd:
            h = _conv(F.pad(h, (0, 1, 0, 1)), sd, p + "downsamplers.0.conv", stride=2, padding=0)
    h = _mid(h, sd, "encoder.mid_block.", groups)
    h = _conv(F.silu(_gn(h, sd, "encoder.conv_norm_out", groups, 1e-6)), sd, "encoder.conv_out")
    return _conv(h, sd, "quant_conv", padding=0)


def gaussian_sample(moments: torch.Tensor, noise: Optional[torch.Tensor]) -> torch.Tensor:
    """DiagonalGaussianDistribution(moments).sample() with the given standard-normal draw (None = .mode())."""
    mean, logvar = moments.chunk(2, dim=1)
    if noise is None:
        return mean
    return mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise


def vae_decode(sd: SD, z: torch.Tensor, groups: int = 32) -> torch.Tensor:
    """AutoencoderKL.decode: post_quant_conv, Decoder (conv_in, UNetMidBlock2D, UpDecoderBlock2D x n [resnets,
    Upsample2D: nearest x2 + 3x3 conv], GroupNorm(eps 1e-6) + SiLU + conv_out)."""
    sd = {k: v.float() for k, v in sd.items()}
    h = _conv(z.float(), sd, "post_quant_conv", padding=0)
    h = _conv(h, sd, "decoder.conv_in")
    h = _mid(h, sd, "decoder.mid_block.", groups)
    for i in range(_count(sd, "decoder.up_blocks.")):
        p = f"decoder.up_blocks.{i}."
        for j in range(_count(sd, p + "resnets.")):
            h = _resnet(h, sd, f"{p}resnets.{j}.", groups)
        if p + "upsamplers.0.conv.weight" in sd:
            h = _conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd, p + "upsamplers.0.conv")
    return _conv(F.silu(_gn(h, sd, "decoder.conv_norm_out", groups, 1e-6)), sd, "decoder.conv_out")


# ------------------------------------------------------------------------------------------- random-init state dicts
def vae_random_state(block_out=(128, 256, 512, 512), layers_per_block=2, in_channels=3, out_channels=3, latent_channels=4,
                     seed=0) -> SD:
    """A state dict with AutoencoderKL's names and shapes (ckpt/stable-diffusion-v1-5/vae/config.json by default) and
    PyTorch-default-like init (uniform +-1/sqrt(fan_in) weights, unit/zero norms)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {}

    def conv(name, co, ci, k):
        b = 1.0 / math.sqrt(ci * k * k)
        sd[name + ".weight"] = (torch.rand(co, ci, k, k, generator=g) * 2 - 1) * b
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * b

    def lin(name, co, ci):
        b = 1.0 / math.sqrt(ci)
        sd[name + ".weight"] = (torch.rand(co, ci, generator=g) * 2 - 1) * b
        sd[name + ".bias"] = (torch.rand(co, generator=g) * 2 - 1) * b

    def norm(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    def resnet(p, ci, co):
        norm(p + "norm1", ci), conv(p + "conv1", co, ci, 3), norm(p + "norm2", co), conv(p + "conv2", co, co, 3)
        if ci != co:
            conv(p + "conv_shortcut", co, ci, 1)

    def mid(p, c):
        resnet(p + "resnets.0.", c, c)
        norm(p + "attentions.0.group_norm", c)
        for n in ("query", "key", "value", "proj_attn"):
            lin(p + "attentions.0." + n, c, c)
        resnet(p + "resnets.1.", c, c)

    conv("encoder.conv_in", block_out[0], in_channels, 3)
    ci = block_out[0]
    for i, co in enumerate(block_out):
        for j in range(layers_per_block):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", ci, co)
            ci = co
        if i != len(block_out) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    mid("encoder.mid_block.", block_out[-1])
    norm("encoder.conv_norm_out", block_out[-1]), conv("encoder.conv_out", 2 * latent_channels, block_out[-1], 3)
    conv("quant_conv", 2 * latent_channels, 2 * latent_channels, 1)
    conv("post_quant_conv", latent_channels, latent_channels, 1)
    rev = list(reversed(block_out))
    conv("decoder.conv_in", rev[0], latent_channels, 3)
    mid("decoder.mid_block.", rev[0])
    ci = rev[0]
    for i, co in enumerate(rev):
        for j in range(layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", ci, co)
            ci = co
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", rev[-1]), conv("decoder.conv_out", out_channels, rev[-1], 3)
    return sd


def clip_text_random_state(vocab=49408, hidden=768, intermediate=3072, layers=12, positions=77, seed=0) -> SD:
    """CLIPTextModel names/shapes (un-prefixed) with transformers-like init (normal 0.02 embeddings, scaled normal projections)."""
    g = torch.Generator().manual_seed(seed)
    sd: SD = {"embeddings.token_embedding.weight": 0.02 * torch.randn(vocab, hidden, generator=g),
              "embeddings.position_embedding.weight": 0.02 * torch.randn(positions, hidden, generator=g)}

    def lin(name, co, ci, std):
        sd[name + ".weight"] = std * torch.randn(co, ci, generator=g)
        sd[name + ".bias"] = 0.02 * torch.randn(co, generator=g)

    def norm(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)

    for i in range(layers):
        p = f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(p + "self_attn." + n, hidden, hidden, hidden ** -0.5)
        norm(p + "layer_norm1", hidden), norm(p + "layer_norm2", hidden)
        lin(p + "mlp.fc1", intermediate, hidden, hidden ** -0.5)
        lin(p + "mlp.fc2", hidden, intermediate, intermediate ** -0.5)
    norm("final_layer_norm", hidden)
    return sd
