"""Static description of the StoryGen UNet topology: layer specs, state-dict keys/shapes, feature keys.

Mirrors what the reference builds imperatively in
  /root/reference/model/unet_2d_condition.py:83-270   (ctor: conv_in, time embedding, down/mid/up, conv_out)
  /root/reference/model/unet_2d_blocks.py:300-372,439-489,518-586,663-709,197-267   (block ctors)
  /root/reference/model/attention.py:26-83,131-234,305-350,373-383                   (transformer ctor)
but as plain data, so that the drop-in nn.Module (storygen_amd/model), the HIP engine
(storygen_amd/engine.py) and the tests all agree on names, shapes and routing.

Feature ("img_dif_condition") keys are assigned by *block index* — `down_{i+1}_{j+1}`, `mid`,
`up_{i}_{j+1}` — exactly the names the producer side uses (unet_2d_condition.py:428-429,445,468-470).
The reference's consumer side re-derives the block number from the latent height
(unet_2d_blocks.py:380-381,600-601), which coincides with the block index only for heights 64..94
(SURVEY F5); keying by index reproduces it at 64x64 and is the only consistent choice at 96x96.
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

DEFAULT_CONFIG = dict(
    sample_size=None, in_channels=4, out_channels=4, flip_sin_to_cos=True, freq_shift=0,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    mid_block_type="UNetMidBlock2DCrossAttn",
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    only_cross_attention=False, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5,
    cross_attention_dim=1280, attention_head_dim=8, use_linear_projection=False, class_embed_type=None,
    num_class_embeds=None, upcast_attention=False, resnet_time_scale_shift="default",
    time_embedding_type="positional", conv_in_kernel=3, conv_out_kernel=3,
)

#: the SD-1.5 UNet config shipped with the reference (ckpt/stable-diffusion-v1-5/unet/config.json:1-36)
SD15_CONFIG = dict(DEFAULT_CONFIG, cross_attention_dim=768, sample_size=512)


def load_config(path_or_dict, subfolder: Optional[str] = None) -> dict:
    """Config dict from a diffusers-style folder / json file / dict, filled with the ctor defaults."""
    if isinstance(path_or_dict, dict):
        raw = path_or_dict
    else:
        p = path_or_dict
        if os.path.isdir(p):
            p = os.path.join(p, subfolder or "", "config.json")
        with open(p) as f:
            raw = json.load(f)
    cfg = dict(DEFAULT_CONFIG)
    cfg.update({k: v for k, v in raw.items() if k in DEFAULT_CONFIG})
    for k in ("down_block_types", "up_block_types", "block_out_channels"):
        cfg[k] = tuple(cfg[k])
    return cfg


@dataclass
class ResnetSpec:
    prefix: str
    cin: int
    cout: int

    @property
    def has_shortcut(self) -> bool:
        return self.cin != self.cout


@dataclass
class XfSpec:
    """One Transformer2DModel (GN -> proj_in -> BasicTransformerBlock -> proj_out -> +x)."""
    prefix: str
    channels: int
    heads: int
    feature_key: str

    @property
    def dim_head(self) -> int:
        return self.channels // self.heads


@dataclass
class BlockSpec:
    kind: str                                  # "down" | "mid" | "up"
    index: int
    resnets: List[ResnetSpec] = field(default_factory=list)
    attns: List[Optional[XfSpec]] = field(default_factory=list)   # aligned with resnets (None = no attention)
    sampler_prefix: Optional[str] = None       # downsamplers.0.conv / upsamplers.0.conv
    channels: int = 0                          # output channels
    skip_channels: Tuple[int, ...] = ()        # up blocks: channels of the popped skip per layer


@dataclass
class UNetArch:
    config: dict
    down: List[BlockSpec]
    mid: BlockSpec
    up: List[BlockSpec]
    temb_dim: int
    feature_keys: List[str]
    feature_channels: Dict[str, int]
    feature_level: Dict[str, int]              # downsampling level (0 => full latent res, 3 => /8)

    @property
    def resnets(self) -> List[ResnetSpec]:
        out = []
        for b in self.down + [self.mid] + self.up:
            out.extend(b.resnets)
        return out


def build_arch(config: dict) -> UNetArch:
    cfg = load_config(config)
    boc = tuple(cfg["block_out_channels"])
    nblk = len(boc)
    if cfg["time_embedding_type"] != "positional":
        raise ValueError("only time_embedding_type='positional' is on the StoryGen path")
    if cfg["use_linear_projection"]:
        raise ValueError("use_linear_projection=True is not exercised by the StoryGen checkpoints")
    if cfg["class_embed_type"] is not None or cfg["num_class_embeds"] is not None:
        raise ValueError("class embeddings are not exercised by the StoryGen checkpoints")
    if cfg["resnet_time_scale_shift"] != "default":
        raise ValueError("resnet_time_scale_shift must be 'default'")
    ahd = cfg["attention_head_dim"]
    heads = tuple(ahd) if isinstance(ahd, (tuple, list)) else (ahd,) * nblk
    lpb = cfg["layers_per_block"]
    temb_dim = boc[0] * 4
    fkeys: List[str] = []
    fch: Dict[str, int] = {}
    flev: Dict[str, int] = {}

    down: List[BlockSpec] = []
    out_c = boc[0]
    for i, typ in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        has_attn = typ == "CrossAttnDownBlock2D"
        if typ not in ("CrossAttnDownBlock2D", "DownBlock2D"):
            raise ValueError(f"{typ} does not exist.")
        blk = BlockSpec("down", i, channels=out_c)
        for j in range(lpb):
            blk.resnets.append(ResnetSpec(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c))
            if has_attn:
                key = f"down_{i + 1}_{j + 1}"
                blk.attns.append(XfSpec(f"down_blocks.{i}.attentions.{j}", out_c, heads[i], key))
                fkeys.append(key), fch.__setitem__(key, out_c), flev.__setitem__(key, i)
            else:
                blk.attns.append(None)
        if i != nblk - 1:
            blk.sampler_prefix = f"down_blocks.{i}.downsamplers.0.conv"
        down.append(blk)

    if cfg["mid_block_type"] != "UNetMidBlock2DCrossAttn":
        raise ValueError(f"unknown mid_block_type : {cfg['mid_block_type']}")
    mc = boc[-1]
    mid = BlockSpec("mid", 0, channels=mc)
    mid.resnets = [ResnetSpec("mid_block.resnets.0", mc, mc), ResnetSpec("mid_block.resnets.1", mc, mc)]
    mid.attns = [XfSpec("mid_block.attentions.0", mc, heads[-1], "mid")]
    fkeys.append("mid"), fch.__setitem__("mid", mc), flev.__setitem__("mid", nblk - 1)

    up: List[BlockSpec] = []
    rboc = list(reversed(boc))
    rheads = list(reversed(heads))
    out_c = rboc[0]
    for i, typ in enumerate(cfg["up_block_types"]):
        prev_c, out_c = out_c, rboc[i]
        in_c = rboc[min(i + 1, nblk - 1)]
        has_attn = typ == "CrossAttnUpBlock2D"
        if typ not in ("CrossAttnUpBlock2D", "UpBlock2D"):
            raise ValueError(f"{typ} does not exist.")
        blk = BlockSpec("up", i, channels=out_c)
        skips = []
        for j in range(lpb + 1):
            skip_c = in_c if j == lpb else out_c
            res_in = prev_c if j == 0 else out_c
            skips.append(skip_c)
            blk.resnets.append(ResnetSpec(f"up_blocks.{i}.resnets.{j}", res_in + skip_c, out_c))
            if has_attn:
                key = f"up_{i}_{j + 1}"
                blk.attns.append(XfSpec(f"up_blocks.{i}.attentions.{j}", out_c, rheads[i], key))
                fkeys.append(key), fch.__setitem__(key, out_c), flev.__setitem__(key, nblk - 1 - i)
            else:
                blk.attns.append(None)
        blk.skip_channels = tuple(skips)
        if i != nblk - 1:
            blk.sampler_prefix = f"up_blocks.{i}.upsamplers.0.conv"
        up.append(blk)

    return UNetArch(cfg, down, mid, up, temb_dim, fkeys, fch, flev)


def _resnet_shapes(r: ResnetSpec, temb: int, out: "OrderedDict[str, tuple]") -> None:
    p = r.prefix
    out[f"{p}.norm1.weight"] = (r.cin,)
    out[f"{p}.norm1.bias"] = (r.cin,)
    out[f"{p}.conv1.weight"] = (r.cout, r.cin, 3, 3)
    out[f"{p}.conv1.bias"] = (r.cout,)
    out[f"{p}.time_emb_proj.weight"] = (r.cout, temb)
    out[f"{p}.time_emb_proj.bias"] = (r.cout,)
    out[f"{p}.norm2.weight"] = (r.cout,)
    out[f"{p}.norm2.bias"] = (r.cout,)
    out[f"{p}.conv2.weight"] = (r.cout, r.cout, 3, 3)
    out[f"{p}.conv2.bias"] = (r.cout,)
    if r.has_shortcut:
        out[f"{p}.conv_shortcut.weight"] = (r.cout, r.cin, 1, 1)
        out[f"{p}.conv_shortcut.bias"] = (r.cout,)


def _xf_shapes(x: XfSpec, cad: int, out: "OrderedDict[str, tuple]") -> None:
    p, c = x.prefix, x.channels
    out[f"{p}.norm.weight"] = (c,)
    out[f"{p}.norm.bias"] = (c,)
    out[f"{p}.proj_in.weight"] = (c, c, 1, 1)
    out[f"{p}.proj_in.bias"] = (c,)
    t = f"{p}.transformer_blocks.0"
    for name, kdim in (("attn1", c), ("attn2", cad), ("attn3", c)):
        out[f"{t}.{name}.to_q.weight"] = (c, c)
        out[f"{t}.{name}.to_k.weight"] = (c, kdim)
        out[f"{t}.{name}.to_v.weight"] = (c, kdim)
        out[f"{t}.{name}.to_out.0.weight"] = (c, c)
        out[f"{t}.{name}.to_out.0.bias"] = (c,)
    for n in ("norm1", "norm2", "norm3", "norm4"):
        out[f"{t}.{n}.weight"] = (c,)
        out[f"{t}.{n}.bias"] = (c,)
    out[f"{t}.ff.net.0.proj.weight"] = (8 * c, c)
    out[f"{t}.ff.net.0.proj.bias"] = (8 * c,)
    out[f"{t}.ff.net.2.weight"] = (c, 4 * c)
    out[f"{t}.ff.net.2.bias"] = (c,)
    out[f"{p}.proj_out.weight"] = (c, c, 1, 1)
    out[f"{p}.proj_out.bias"] = (c,)


def param_shapes(arch: UNetArch) -> "OrderedDict[str, tuple]":
    """State-dict keys -> shapes; the checkpoint contract of SURVEY §8b (PyTorch layouts)."""
    cfg = arch.config
    boc = cfg["block_out_channels"]
    cad = cfg["cross_attention_dim"]
    k_in, k_out = cfg["conv_in_kernel"], cfg["conv_out_kernel"]
    out: "OrderedDict[str, tuple]" = OrderedDict()
    out["conv_in.weight"] = (boc[0], cfg["in_channels"], k_in, k_in)
    out["conv_in.bias"] = (boc[0],)
    out["time_embedding.linear_1.weight"] = (arch.temb_dim, boc[0])
    out["time_embedding.linear_1.bias"] = (arch.temb_dim,)
    out["time_embedding.linear_2.weight"] = (arch.temb_dim, arch.temb_dim)
    out["time_embedding.linear_2.bias"] = (arch.temb_dim,)
    for blk in arch.down + [arch.mid] + arch.up:
        for j, r in enumerate(blk.resnets):
            _resnet_shapes(r, arch.temb_dim, out)
        for a in blk.attns:
            if a is not None:
                _xf_shapes(a, cad, out)
        if blk.sampler_prefix:
            c = blk.channels
            out[f"{blk.sampler_prefix}.weight"] = (c, c, 3, 3)
            out[f"{blk.sampler_prefix}.bias"] = (c,)
    out["conv_norm_out.weight"] = (boc[0],)
    out["conv_norm_out.bias"] = (boc[0],)
    out["conv_out.weight"] = (cfg["out_channels"], boc[0], k_out, k_out)
    out["conv_out.bias"] = (cfg["out_channels"],)
    return out


def feature_shapes(arch: UNetArch, height: int, width: int) -> "OrderedDict[str, Tuple[int, int]]":
    """feature key -> (tokens, channels) for one prior frame at the given latent size."""
    out = OrderedDict()
    for k in arch.feature_keys:
        lv = arch.feature_level[k]
        out[k] = ((height >> lv) * (width >> lv), arch.feature_channels[k])
    return out
