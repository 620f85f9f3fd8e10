"""Drop-in `AutoencoderKL` and `CLIPTextModel` on the HIP engines of storygen_amd/encoders.py.

They mirror what the reference's scripts and pipeline touch, nothing more:

  AutoencoderKL.from_pretrained(path, subfolder="vae")            inference.py:46, train_StorySalon_stage2.py:142
  vae.encode(x).latent_dist.sample() / vae.decode(z).sample        model/pipeline.py:198-205,392,401; train_StorySalon_stage2.py:281-288
  vae.requires_grad_(False), vae.to(device, dtype=...)             train_StorySalon_stage2.py:167,234
  CLIPTextModel.from_pretrained(path, subfolder="text_encoder")    inference.py:45, train_StorySalon_stage2.py:141
  text_encoder(input_ids, attention_mask=mask)[0]                  model/pipeline.py:137,183; train_StorySalon_stage2.py:283,302
  text_encoder.config.use_attention_mask                           model/pipeline.py:128-135

Weights keep the third-party packages' names and layouts (diffusers 0.13.1 AutoencoderKL, transformers CLIPTextModel), so their
checkpoints load unchanged; `from_torch(module)` adopts an already-constructed torch module of either package.  Inference only (the
reference freezes both).  There is no CPU path: calling encode / decode / the text encoder needs the HIP device."""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Dict, Optional

import torch

from ..encoders import ClipTextEngine, VaeEngine, clip_text_param_shapes, init_state, vae_param_shapes
from .unet_2d_condition import CONFIG_NAME, SAFETENSORS_NAME, WEIGHTS_NAME, FrozenConfig

SD = Dict[str, torch.Tensor]


def _load_weights(folder: str, names) -> SD:
    for n in names:
        p = os.path.join(folder, n)
        if os.path.exists(p):
            if n.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    raise EnvironmentError(f"none of {list(names)} under {folder}")


class _HipModule:
    """State-dict holder with the few nn.Module methods the reference's scripts call on the frozen encoders."""
    _engine = None

    def _adopt(self, sd: SD, shapes: Dict[str, tuple], strict: bool = True):
        sd = dict(sd)
        missing = [k for k in shapes if k not in sd]
        extra = [k for k in sd if k not in shapes]
        bad = [k for k in shapes if k in sd and tuple(sd[k].shape) != tuple(shapes[k])]
        if bad or (strict and (missing or extra)):
            raise RuntimeError(f"{type(self).__name__}.load_state_dict: missing {missing[:4]}, unexpected {extra[:4]}, wrong shape {bad[:4]}")
        for k in shapes:
            if k in sd:
                self._sd[k] = sd[k].detach().to(self._sd[k].device, self._sd[k].dtype).clone()
        self._engine = None

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self._sd)

    def parameters(self):
        return iter(self._sd.values())

    def named_parameters(self):
        return iter(self._sd.items())

    def requires_grad_(self, flag: bool = False):
        if flag:
            raise NotImplementedError(f"{type(self).__name__} is inference-only (the reference freezes it)")
        return self

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError(f"{type(self).__name__} is inference-only")
        return self

    def to(self, *args, **kwargs):
        device, dtype = kwargs.get("device"), kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        for k, v in self._sd.items():
            self._sd[k] = v.to(device=device if device is not None else v.device, dtype=dtype if dtype is not None else v.dtype)
        self._engine = None
        return self

    def half(self):
        return self.to(torch.float16)

    def float(self):
        return self.to(torch.float32)

    @property
    def device(self) -> torch.device:
        return next(iter(self._sd.values())).device

    @property
    def dtype(self) -> torch.dtype:
        return next(iter(self._sd.values())).dtype

    @property
    def config(self) -> FrozenConfig:
        return self._config

    def _save(self, save_directory: str, class_name: str, weights_name: str, safetensors_name: str, safe_serialization: bool, extra=None):
        os.makedirs(save_directory, exist_ok=True)
        cfg = OrderedDict(extra or {})
        cfg["_class_name"] = class_name
        cfg.update(self._config)
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = OrderedDict((k, v.detach().cpu().contiguous()) for k, v in self._sd.items())
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, safetensors_name))
        else:
            torch.save(sd, os.path.join(save_directory, weights_name))


# =================================================================================================================== VAE
class DiagonalGaussianDistribution:
    """diffusers' class of the same name as the pipeline uses it: `.sample(generator=None)`, `.mode()`, `.mean`, `.logvar`, `.std`."""

    def __init__(self, mean: torch.Tensor, logvar: torch.Tensor, engine: VaeEngine, dtype: torch.dtype):
        self._mean, self._logvar, self._engine, self._dtype = mean, logvar, engine, dtype

    @property
    def mean(self):
        return self._mean.to(self._dtype)

    @property
    def logvar(self):
        return self._logvar.clamp(-30.0, 20.0).to(self._dtype)

    @property
    def std(self):
        return torch.exp(0.5 * self._logvar.clamp(-30.0, 20.0)).to(self._dtype)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        dev = generator.device if generator is not None else self._mean.device
        noise = torch.randn(self._mean.shape, generator=generator, device=dev, dtype=torch.float32)
        return self._engine.sample(self._mean, self._logvar, noise).to(self._dtype)

    def mode(self) -> torch.Tensor:
        return self._mean.to(self._dtype)


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist

    def __getitem__(self, i):
        return (self.latent_dist,)[i]


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, i):
        return (self.sample,)[i]


VAE_DEFAULTS = OrderedDict(in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",), up_block_types=("UpDecoderBlock2D",),
                           block_out_channels=(64,), layers_per_block=1, act_fn="silu", latent_channels=4, norm_num_groups=32,
                           sample_size=32, scaling_factor=0.18215)


class AutoencoderKL(_HipModule):
    def __init__(self, seed: int = 0, **config):
        unknown = [k for k in config if k not in VAE_DEFAULTS]
        if unknown:
            raise TypeError(f"AutoencoderKL: unknown config keys {unknown}")
        cfg = OrderedDict(VAE_DEFAULTS)
        cfg.update(config)
        for k in ("down_block_types", "up_block_types", "block_out_channels"):
            cfg[k] = tuple(cfg[k])
        if any(t != "DownEncoderBlock2D" for t in cfg["down_block_types"]) or any(t != "UpDecoderBlock2D" for t in cfg["up_block_types"]):
            raise NotImplementedError("AutoencoderKL: only DownEncoderBlock2D / UpDecoderBlock2D")
        if cfg["act_fn"] not in ("silu", "swish") or len(cfg["down_block_types"]) != len(cfg["block_out_channels"]):
            raise NotImplementedError("AutoencoderKL: act_fn must be silu and one block type per block_out_channels entry")
        self._config = FrozenConfig(cfg).freeze()
        self._shapes = vae_param_shapes(cfg["block_out_channels"], cfg["layers_per_block"], cfg["in_channels"], cfg["out_channels"],
                                        cfg["latent_channels"])
        self._sd: SD = init_state(self._shapes, seed)

    # -------------------------------------------------------------------------------------------- (de)serialisation
    @classmethod
    def from_config(cls, config, subfolder: Optional[str] = None) -> "AutoencoderKL":
        if not isinstance(config, dict):
            p = os.path.join(config, subfolder or "", CONFIG_NAME) if os.path.isdir(config) else config
            with open(p) as f:
                config = json.load(f)
        return cls(**{k: v for k, v in config.items() if k in VAE_DEFAULTS})

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: Optional[str] = None, torch_dtype: Optional[torch.dtype] = None,
                        **kwargs) -> "AutoencoderKL":
        folder = os.path.join(pretrained_model_path, subfolder or "")
        model = cls.from_config(os.path.join(folder, CONFIG_NAME))
        model.load_state_dict(_load_weights(folder, (SAFETENSORS_NAME, WEIGHTS_NAME)))
        return model.to(torch_dtype) if torch_dtype is not None else model

    @classmethod
    def from_torch(cls, module) -> "AutoencoderKL":
        """Adopt a constructed diffusers AutoencoderKL (its `.config` and `.state_dict()`)."""
        model = cls(**{k: v for k, v in dict(module.config).items() if k in VAE_DEFAULTS})
        model.load_state_dict({k: v for k, v in module.state_dict().items()})
        p = next(iter(module.state_dict().values()))
        return model.to(p.device, p.dtype)

    def load_state_dict(self, state_dict: SD, strict: bool = True):
        self._adopt(state_dict, self._shapes, strict)

    def save_pretrained(self, save_directory: str, safe_serialization: bool = False, **kwargs):
        self._save(save_directory, "AutoencoderKL", WEIGHTS_NAME, SAFETENSORS_NAME, safe_serialization, {"_diffusers_version": "0.13.1"})

    def enable_slicing(self):      # model/pipeline.py exposes enable_vae_slicing; batches are already processed image by image
        pass

    def disable_slicing(self):
        pass

    # ------------------------------------------------------------------------------------------------------ forward
    def _eng(self) -> VaeEngine:
        if self._engine is None:
            if self.device.type != "cuda":
                raise RuntimeError("AutoencoderKL: move the model to the HIP device first (.to('cuda')); there is no CPU path")
            self._engine = VaeEngine(self._sd, self.device, groups=self._config["norm_num_groups"])
        return self._engine

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        eng = self._eng()
        mean, logvar = eng.encode(x)
        dist = DiagonalGaussianDistribution(mean, logvar, eng, x.dtype if x.is_floating_point() else torch.float32)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        img = self._eng().decode(z).to(z.dtype)
        return DecoderOutput(img) if return_dict else (img,)

    def forward(self, sample: torch.Tensor, sample_posterior: bool = False, return_dict: bool = True,
                generator: Optional[torch.Generator] = None):
        dist = self.encode(sample).latent_dist
        return self.decode(dist.sample(generator) if sample_posterior else dist.mode(), return_dict)

    __call__ = forward


# ================================================================================================================== CLIP
CLIP_TEXT_DEFAULTS = OrderedDict(vocab_size=49408, hidden_size=512, intermediate_size=2048, projection_dim=512, num_hidden_layers=12,
                                 num_attention_heads=8, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                                 attention_dropout=0.0, initializer_range=0.02, initializer_factor=1.0, pad_token_id=1, bos_token_id=0,
                                 eos_token_id=2, model_type="clip_text_model")
CLIP_WEIGHTS = ("model.safetensors", "pytorch_model.bin")


class CLIPTextModelOutput(tuple):
    """BaseModelOutputWithPooling as the pipeline reads it: `out[0]` / `.last_hidden_state`, `out[1]` / `.pooler_output`."""

    def __new__(cls, last_hidden_state, pooler_output):
        self = super().__new__(cls, (last_hidden_state, pooler_output))
        self.last_hidden_state, self.pooler_output = last_hidden_state, pooler_output
        return self


class CLIPTextModel(_HipModule):
    def __init__(self, config: Optional[dict] = None, seed: int = 0, **kwargs):
        cfg = OrderedDict(CLIP_TEXT_DEFAULTS)
        raw = dict(config or {})
        raw.update(kwargs)
        if "text_config" in raw and isinstance(raw["text_config"], dict):      # a full CLIPModel config (ckpt/.../CLIP/config.json)
            raw = dict(raw["text_config"])
        extra = OrderedDict((k, v) for k, v in raw.items() if k not in CLIP_TEXT_DEFAULTS and not k.startswith("_"))
        cfg.update({k: v for k, v in raw.items() if k in CLIP_TEXT_DEFAULTS})
        cfg.update(extra)                                                         # e.g. use_attention_mask, read by the pipeline
        if cfg["hidden_act"] not in ("quick_gelu", "gelu"):
            raise NotImplementedError(f"CLIPTextModel: hidden_act {cfg['hidden_act']!r}")
        self._config = FrozenConfig(cfg).freeze()
        self._shapes = clip_text_param_shapes(cfg["vocab_size"], cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"],
                                              cfg["max_position_embeddings"])
        self._sd: SD = init_state(self._shapes, seed, embed_std=cfg["initializer_range"])

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: Optional[str] = None, torch_dtype: Optional[torch.dtype] = None,
                        **kwargs) -> "CLIPTextModel":
        folder = os.path.join(pretrained_model_path, subfolder or "")
        with open(os.path.join(folder, CONFIG_NAME)) as f:
            model = cls(json.load(f))
        model.load_state_dict(_load_weights(folder, CLIP_WEIGHTS), strict=False)
        return model.to(torch_dtype) if torch_dtype is not None else model

    @classmethod
    def from_torch(cls, module) -> "CLIPTextModel":
        """Adopt a constructed transformers CLIPTextModel."""
        model = cls(module.config.to_dict() if hasattr(module.config, "to_dict") else dict(module.config))
        model.load_state_dict(module.state_dict(), strict=False)
        p = next(iter(module.state_dict().values()))
        return model.to(p.device, p.dtype)

    def load_state_dict(self, state_dict: SD, strict: bool = True):
        """Accepts transformers 4.x names (`text_model.*`), the un-prefixed names of newer releases, and ignores the
        non-parameter `position_ids` buffer old checkpoints carry."""
        sd = {}
        for k, v in state_dict.items():
            if k.endswith("position_ids"):
                continue
            sd[k if k.startswith("text_model.") else "text_model." + k] = v
        missing = [k for k in self._shapes if k not in sd]
        if missing:
            raise RuntimeError(f"CLIPTextModel.load_state_dict: missing {missing[:4]} (+{max(0, len(missing) - 4)} more)")
        self._adopt(sd, self._shapes, strict)

    def save_pretrained(self, save_directory: str, safe_serialization: bool = False, **kwargs):
        self._save(save_directory, "CLIPTextModel", CLIP_WEIGHTS[1], CLIP_WEIGHTS[0], safe_serialization,
                   {"architectures": ["CLIPTextModel"]})

    def _eng(self) -> ClipTextEngine:
        if self._engine is None:
            if self.device.type != "cuda":
                raise RuntimeError("CLIPTextModel: move the model to the HIP device first (.to('cuda')); there is no CPU path")
            c = self._config
            self._engine = ClipTextEngine(self._sd, self.device, heads=c["num_attention_heads"], eps=c["layer_norm_eps"],
                                          hidden_act=c["hidden_act"])
        return self._engine

    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, position_ids=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        if position_ids is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("CLIPTextModel: position_ids / output_attentions / output_hidden_states are not supported")
        hidden, pooled = self._eng()(input_ids, attention_mask)
        return CLIPTextModelOutput(hidden.to(self.dtype), pooled.to(self.dtype))

    __call__ = forward
