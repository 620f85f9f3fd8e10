"""Drop-in for `model.unet_2d_condition.UNet2DConditionModel` of the reference
(/root/reference/model/unet_2d_condition.py:35-510), computing on the HIP engine.

What is mirrored:
  * constructor keywords and defaults (:83-117), `.config` (attribute and item access), `.sample_size`, `.in_channels`,
    `.dtype`, `.device`, and the ValueErrors for unknown block types / unsupported switches (:130,141,211);
  * the module tree's NAMES — `named_modules()` yields `...transformer_blocks.0.attn3` etc., which is how
    train_StorySalon_stage2.py:170-177 selects the trainable modules — and `state_dict()` keys / shapes / PyTorch
    layouts, so checkpoints written by the reference load unchanged (`from_pretrained`, `load_state_dict`,
    `load_SDM_state_dict` :487-510) and `save_pretrained` writes the diffusers folder layout;
  * `forward(sample, timestep, encoder_hidden_states, image_hidden_states=None, class_labels=None,
    cross_attention_kwargs=None, return_dict=True)` -> `UNet2DConditionOutput(sample, img_dif_conditions)` or the
    2-tuple (:338-485): without `image_hidden_states` the 16 features are harvested and returned, with it they are
    consumed by attn3 and the returned dict is empty.  Feature keys are by block index;
  * the memory knobs `set_attention_slice`, `enable/disable_xformers_memory_efficient_attention`,
    `enable_gradient_checkpointing` are accepted and do nothing: the flash-style kernel never materialises scores.

The parameters live in ordinary nn.Parameters (PyTorch layouts); the engine keeps repacked fp16 copies that are
refreshed whenever a parameter's version counter moves.  There is no CPU or eager-PyTorch path: `forward` on a model
that is not on a HIP device raises.  Backward (BASELINE config 4) is not implemented yet: calling `forward` with
autograd recording on trainable parameters raises instead of silently returning a graph-less tensor.
"""
from __future__ import annotations

import json
import math
import os
from collections import OrderedDict
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from ..arch import DEFAULT_CONFIG, build_arch, feature_shapes, load_config, param_shapes

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"
SAFETENSORS_NAME = "diffusion_pytorch_model.safetensors"


class FrozenConfig(OrderedDict):
    """diffusers' FrozenDict behaviour as the callers use it: `config.sample_size` and `config["sample_size"]`."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        raise AttributeError("config is frozen")

    def __setitem__(self, name, value):
        if getattr(self, "_FrozenConfig__frozen", False):
            raise TypeError("config is frozen")
        super().__setitem__(name, value)

    def freeze(self):
        object.__setattr__(self, "_FrozenConfig__frozen", True)
        return self


class UNet2DConditionOutput(tuple):
    """(sample, img_dif_conditions) with attribute and tuple access, like the reference's BaseOutput dataclass
    (unet_2d_condition.py:24-32)."""

    def __new__(cls, sample, img_dif_conditions):
        return super().__new__(cls, (sample, img_dif_conditions))

    sample = property(lambda self: self[0])
    img_dif_conditions = property(lambda self: self[1])


class _Node(nn.Module):
    """A name in the module tree.  Leaves hold `weight` / `bias`; the arithmetic happens in the HIP engine."""

    def forward(self, *a, **k):
        raise RuntimeError("submodules of the HIP UNet are parameter containers; call the UNet itself")


def _init_param(name: str, shape: Tuple[int, ...], weight_shape: Tuple[int, ...]) -> torch.Tensor:
    """PyTorch's default initialisation of the layer the reference would have built (nn.Conv2d / nn.Linear:
    U(+-1/sqrt(fan_in)) for weight and bias; norms: ones / zeros)."""
    leaf = name.rsplit(".", 2)[-2]
    if leaf.startswith("norm") or leaf == "conv_norm_out":
        return torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
    fan_in = 1
    for d in weight_shape[1:]:
        fan_in *= d
    bound = 1.0 / math.sqrt(fan_in)
    return torch.empty(shape).uniform_(-bound, bound)


class UNet2DConditionModel(nn.Module):
    config_name = CONFIG_NAME
    _supports_gradient_checkpointing = True

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                                      "CrossAttnDownBlock2D", "DownBlock2D"),
                 mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
                 up_block_types: Tuple[str, ...] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D",
                                                    "CrossAttnUpBlock2D"),
                 only_cross_attention: Union[bool, Tuple[bool, ...]] = False,
                 block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu",
                 norm_num_groups: Optional[int] = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 1280,
                 attention_head_dim: Union[int, Tuple[int, ...]] = 8, use_linear_projection: bool = False,
                 class_embed_type: Optional[str] = None, num_class_embeds: Optional[int] = None,
                 upcast_attention: bool = False, resnet_time_scale_shift: str = "default",
                 time_embedding_type: str = "positional", conv_in_kernel: int = 3, conv_out_kernel: int = 3):
        super().__init__()
        cfg = {k: v for k, v in locals().items() if k in DEFAULT_CONFIG}
        if only_cross_attention not in (False, (False,) * len(block_out_channels)):
            raise ValueError("only_cross_attention is not exercised by the StoryGen checkpoints")
        if act_fn not in ("silu", "swish") or downsample_padding != 1 or mid_block_scale_factor != 1 \
                or norm_num_groups is None or conv_in_kernel != 3 or conv_out_kernel != 3:
            raise ValueError("config differs from the SD-1.5-style UNet the HIP engine implements")
        self._arch = build_arch(cfg)               # raises ValueError for unknown block types / unsupported switches
        object.__setattr__(self, "_config", FrozenConfig(load_config(cfg)).freeze())
        self.sample_size = sample_size
        self.in_channels = in_channels
        shapes = param_shapes(self._arch)
        for name, shape in shapes.items():
            wshape = shapes[name[: -len("bias")] + "weight"] if name.endswith("bias") else shape
            self._register(name, nn.Parameter(_init_param(name, shape, wshape)))
        self._engines: Dict[tuple, Any] = {}
        self._weights = None
        self._weights_tag = None
        self._weights_dev = None
        self._trainers = {}

    # ------------------------------------------------------------------------------------------ module tree
    def _register(self, dotted: str, p: nn.Parameter):
        node: nn.Module = self
        *path, leaf = dotted.split(".")
        for part in path:
            nxt = node._modules.get(part)
            if nxt is None:
                nxt = _Node()
                node.add_module(part, nxt)
            node = nxt
        node.register_parameter(leaf, p)

    @property
    def config(self) -> FrozenConfig:
        return self._config

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    # ------------------------------------------------------------------------------------------ (de)serialisation
    @classmethod
    def from_config(cls, config, subfolder: Optional[str] = None, **kwargs) -> "UNet2DConditionModel":
        """`config`: a dict, a json file or a diffusers folder (+subfolder), as train_StorySalon_stage1.py:146 uses it."""
        raw = config if isinstance(config, dict) else load_config(config, subfolder)
        raw = {k: v for k, v in dict(raw).items() if k in DEFAULT_CONFIG}
        raw.update({k: v for k, v in kwargs.items() if k in DEFAULT_CONFIG})
        return cls(**raw)

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: Optional[str] = None,
                        torch_dtype: Optional[torch.dtype] = None, **kwargs) -> "UNet2DConditionModel":
        """inference.py:47 / train_StorySalon_stage2.py:146: `<path>/<subfolder>/config.json` + the weights file."""
        folder = os.path.join(pretrained_model_path, subfolder or "")
        model = cls.from_config(os.path.join(folder, CONFIG_NAME))
        st = os.path.join(folder, SAFETENSORS_NAME)
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            path = os.path.join(folder, WEIGHTS_NAME)
            if not os.path.exists(path):
                raise EnvironmentError(f"no {WEIGHTS_NAME} or {SAFETENSORS_NAME} under {folder}")
            sd = torch.load(path, map_location="cpu", weights_only=True)
        model.load_state_dict(sd)
        if torch_dtype is not None:
            model = model.to(torch_dtype)
        return model.eval()

    def save_pretrained(self, save_directory: str, safe_serialization: bool = False, **kwargs):
        """diffusers folder layout (train_StorySalon_stage2.py:348-357 via the pipeline's save_pretrained)."""
        os.makedirs(save_directory, exist_ok=True)
        cfg = OrderedDict(_class_name=type(self).__name__, _diffusers_version="0.13.1")
        cfg.update(self.config)
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = OrderedDict((k, v.detach().cpu()) for k, v in self.state_dict().items())
        if safe_serialization:
            from safetensors.torch import save_file
            save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(save_directory, SAFETENSORS_NAME))
        else:
            torch.save(sd, os.path.join(save_directory, WEIGHTS_NAME))

    def load_SDM_state_dict(self, state_dict_SDM, **kwargs):
        """Adopt a plain SD-1.5 UNet checkpoint (unet_2d_condition.py:487-510): unknown keys raise KeyError, shape
        mismatches are dropped (with a message), and every key SD-1.5 lacks (`attn3.*`, `norm4.*`) is filled from the
        same block's `attn1` / `norm1`."""
        own = self.state_dict()
        sdm = dict(state_dict_SDM)
        for k, v in list(sdm.items()):
            if k not in own:
                raise KeyError(f"SDM state_dict key {k} does not exist in model")
            if v.shape != own[k].shape:
                print(f"state_dict shape mismatch, SDM {v.shape}, our {own[k].shape}")
                del sdm[k]
        for k in own:
            if k not in sdm:
                src = k.replace("attn3", "attn1").replace("norm4", "norm1")
                sdm[k] = sdm[src]
                print(f"state_dict key {k} is initialized with self attention.")
        own.update(sdm)
        return self.load_state_dict(own, **kwargs)

    # ------------------------------------------------------------------------------------------ no-op knobs
    def set_attention_slice(self, slice_size):
        n = len(self._arch.feature_keys) * 3       # sliceable attention layers: attn1/2/3 per transformer block
        if isinstance(slice_size, list) and len(slice_size) != n:
            raise ValueError(f"You have provided {len(slice_size)}, but {self.config} has {n} different attention layers.")
        heads = self.config["attention_head_dim"]
        hmax = max(heads) if isinstance(heads, (list, tuple)) else heads
        for s in (slice_size if isinstance(slice_size, list) else [slice_size]):
            if isinstance(s, int) and s > hmax:
                raise ValueError(f"size {s} has to be smaller or equal to {hmax}.")

    def set_use_memory_efficient_attention_xformers(self, valid: bool, attention_op=None):
        pass

    def enable_xformers_memory_efficient_attention(self, attention_op=None):
        pass

    def disable_xformers_memory_efficient_attention(self):
        pass

    def _set_gradient_checkpointing(self, module, value=False):
        pass

    def enable_gradient_checkpointing(self):
        pass

    def disable_gradient_checkpointing(self):
        pass

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _engine_weights(self):
        """The repacked device weights, kept current with the module's parameters.  Staleness is tracked per parameter
        ((data_ptr, _version) pairs); new values are copied INTO the existing repacked tensors, so engines, samplers and
        captured hipGraphs built on them stay valid: an optimizer step on the attn3 modules (stage 2) re-packs just those,
        anything else (load_state_dict, ...) re-packs the checkpoint in place.  Only a device change re-allocates."""
        from ..engine import EngineWeights
        named = list(self.named_parameters())
        tags = {n: (p.data_ptr(), p._version) for n, p in named}
        if self._weights is None or self._weights_dev != self.device:
            self._weights = EngineWeights(self._arch, self.state_dict(), self.device)
            self._weights_dev = self.device
            self._engines.clear()
            self._trainers.clear()
        elif tags != self._weights_tag:
            changed = [n for n, tg in tags.items() if self._weights_tag.get(n) != tg]
            if all(".attn3." in n for n in changed):
                prefixes = {n.split(".transformer_blocks.")[0] for n in changed}
                self._weights.refresh_attn3_(self.state_dict(), prefixes)
            elif all(".attn1." in n for n in changed):                       # stage 1
                prefixes = {n.split(".transformer_blocks.")[0] for n in changed}
                self._weights.refresh_attn1_(self.state_dict(), prefixes)
            else:
                self._weights.reload_(self.state_dict())
                self._trainers.clear()             # the trainer keeps its own repacked copies of the frozen layers
        self._weights_tag = tags
        return self._weights

    def _engine(self, B: int, H: int, W: int, R: int, S: int):
        from ..engine import UNetEngine
        wts = self._engine_weights()
        key = (B, H, W, R, S)
        eng = self._engines.get(key)
        if eng is None:
            if len(self._engines) >= 4:            # bound the activation memory held by stale shapes
                self._engines.pop(next(iter(self._engines)))
            eng = self._engines[key] = UNetEngine(self._arch, None, self.device, B, H, W, R, S, weights=wts)
        return eng

    def _forward_train(self, sample, timestep, encoder_hidden_states, image_hidden_states, return_dict, module: str = "attn3"):
        """Main pass under autograd: gradients for the parameters of the trainable module (`attn3` with image context — stage 2 / COCO;
        `attn1` without — stage 1) through storygen_amd.train.MainPassFunction."""
        from ..train import MainPassFunction, UNetTrainer
        B, _, H, W = sample.shape
        shapes = feature_shapes(self._arch, H, W)
        k0 = self._arch.feature_keys[0]
        R = 0 if image_hidden_states is None else image_hidden_states[k0].shape[1] // shapes[k0][0]
        # The reference's stage-2 loop draws 1-3 prior frames at random per step (train_StorySalon_stage2.py:306-313), so R changes on
        # most steps: the trainer is keyed on (B, H, W, module) only — forward_main / backward_main never read its n_ref (the context
        # tensors carry their own length) — and is NOT rebuilt when R changes.  _engine_weights() runs on every call so that a
        # load_state_dict between two training forwards invalidates the cached trainers (their frozen-layer copies would be stale).
        wts = self._engine_weights()                     # (clears self._trainers itself when it re-packs the whole checkpoint)
        key = (B, H, W, module)
        tr = self._trainers.get(key)                     # kept apart from the inference engines' LRU
        if tr is None:
            if len(self._trainers) >= 2:
                self._trainers.pop(next(iter(self._trainers)))
            tr = self._trainers[key] = UNetTrainer(self._arch, self.state_dict(), self.device, B, H, W, n_ref=R,
                                                  ref_engine=object(),      # reference passes go through forward(None)
                                                  weights=wts, trainable=module)
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        t = t.to(self.device, torch.float32).reshape(-1)
        t = t.expand(B) if t.numel() == 1 else t
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad]
        keys = list(shapes) if image_hidden_states is not None else []
        feats = [image_hidden_states[k].to(self.device, torch.float16).reshape(-1, shapes[k][1]).contiguous() for k in keys]
        text16 = encoder_hidden_states.to(self.device, torch.float16).reshape(-1, encoder_hidden_states.shape[-1]).contiguous()
        pred = MainPassFunction.apply(tr, [n for n, _ in named], keys, sample.to(self.device, torch.float32).contiguous(), t, text16,
                                      *feats, *[p for _, p in named])
        out = pred.to(sample.dtype)
        return UNet2DConditionOutput(out, {}) if return_dict else (out, {})

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor, image_hidden_states: Optional[Dict[str, torch.Tensor]] = None,
                class_labels: Optional[torch.Tensor] = None, cross_attention_kwargs: Optional[Dict[str, Any]] = None,
                return_dict: bool = True):
        if class_labels is not None:
            raise ValueError("class_labels should be provided only when num_class_embeds > 0 (unsupported here)")
        if cross_attention_kwargs:
            raise ValueError("cross_attention_kwargs are not supported by the HIP attention kernels")
        if self.device.type != "cuda":
            raise RuntimeError("the HIP UNet has no CPU path: move the model to a HIP device (model.to('cuda'))")
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if training:
            mods = {("attn1" if ".attn1." in n else "attn3" if ".attn3." in n else n) for n, p in self.named_parameters() if p.requires_grad}
            if mods == {"attn3"}:                       # stage 2 / COCO, train_StorySalon_stage2.py:170-177
                if image_hidden_states is not None:
                    return self._forward_train(sample, timestep, encoder_hidden_states, image_hidden_states, return_dict, "attn3")
                # harvest pass: attn3 is not evaluated, so no gradient reaches a trainable parameter through it
                with torch.no_grad():
                    return self.forward(sample, timestep, encoder_hidden_states, None, return_dict=return_dict)
            if mods == {"attn1"} and image_hidden_states is None:      # stage 1, train_StorySalon_stage1.py:175-179,288
                return self._forward_train(sample, timestep, encoder_hidden_states, None, return_dict, "attn1")
            raise NotImplementedError("the HIP backward covers: the attn3 modules trainable (stage 2, train_StorySalon_stage2.py:170-177) or the "
                                      "attn1 modules trainable with image_hidden_states=None (stage 1, train_StorySalon_stage1.py:175,288); "
                                      f"trainable here: {sorted(mods)[:3]}, image context: {image_hidden_states is not None}")
        B, _, H, W = sample.shape
        S = encoder_hidden_states.shape[1]
        arch = self._arch
        shapes = feature_shapes(arch, H, W)
        if image_hidden_states is None:
            R = 1
        else:
            if set(image_hidden_states) != set(shapes):
                raise KeyError(f"image_hidden_states must have the keys {list(shapes)}")
            k0 = arch.feature_keys[0]
            R, rem = divmod(image_hidden_states[k0].shape[1], shapes[k0][0])
            if rem or R < 1:
                raise ValueError("image_hidden_states token count is not a multiple of the feature map size")
        eng = self._engine(B, H, W, R, S)
        t = timestep                                                                       # :379-390
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float32)
        t = t.to(torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        eng.x_in.copy_(sample)
        eng.t_in.copy_(t)
        eng.text_in.copy_(encoder_hidden_states)
        out_dtype = sample.dtype
        if image_hidden_states is None:
            eps = eng.forward(harvest_slot=0)
            feats = OrderedDict((k, v.to(out_dtype, copy=True)) for k, v in eng.features(0).items())   # the .clone() of :428
        else:
            for k, v in image_hidden_states.items():
                eng.ctx[k].copy_(v)
            eps = eng.forward(consume=True)
            feats = {}
        out = eps.to(out_dtype, copy=True)
        if not return_dict:
            return (out, feats)
        return UNet2DConditionOutput(out, feats)
