"""ORACLE tooling — runs the reference's own `model/*.py` *verbatim* (build container only).

/root/reference is imported unmodified on top of oracle/diffusers_shim; the CLIP text encoder, tokenizer and
VAE (outside the hot path) are replaced by table look-ups so that
`StableDiffusionPipeline.__call__` (/root/reference/model/pipeline.py:273-484) executes its real loop
(:411-469) on our synthetic inputs.  Nothing here travels to the GPU box; only the vectors it produces do
(tests/golden/, written by oracle/make_golden.py).
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch

REFERENCE_ROOT = os.environ.get("STORYGEN_REFERENCE", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusers_shim")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model"))


def import_reference():
    """Returns (UNet2DConditionModel, StableDiffusionPipeline, DDIMScheduler) from the reference tree."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    for p in (REFERENCE_ROOT, _SHIM):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for name in list(sys.modules):
        if name == "model" or name.startswith("model."):
            mod = sys.modules[name]
            if not (getattr(mod, "__file__", None) or REFERENCE_ROOT).startswith(REFERENCE_ROOT):
                del sys.modules[name]
    from model.unet_2d_condition import UNet2DConditionModel  # type: ignore
    from model.pipeline import StableDiffusionPipeline  # type: ignore
    from diffusers import DDIMScheduler  # the shim
    return UNet2DConditionModel, StableDiffusionPipeline, DDIMScheduler


class _TableTokenizer:
    """prompt string -> row index; `_encode_prompt` (pipeline.py:87-196) only needs `.input_ids` of equal shape."""
    model_max_length = 77

    def __init__(self, prompts: List[str]):
        self.index = {p: i for i, p in enumerate(prompts)}

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None):
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        ids = torch.zeros(len(prompts), self.model_max_length, dtype=torch.long)
        for r, p in enumerate(prompts):
            ids[r, 0] = self.index[p]
        return SimpleNamespace(input_ids=ids, attention_mask=torch.ones_like(ids))


class _TableTextEncoder(torch.nn.Module):
    def __init__(self, table: torch.Tensor):
        super().__init__()
        self.table = torch.nn.Parameter(table, requires_grad=False)
        self.config = SimpleNamespace()

    def forward(self, input_ids, attention_mask=None):
        return (self.table[input_ids[:, 0]],)


class _QueueVAE:
    """`vae.encode(x).latent_dist.sample()` pops the next preset latent (already divided by 0.18215 so that the
    pipeline's `* 0.18215`, pipeline.py:393,402, restores it); call order is zero image then the R refs (:390-404)."""

    def __init__(self, latents: List[torch.Tensor]):
        self.queue = list(latents)
        self.config = SimpleNamespace(block_out_channels=(128, 256, 512, 512))

    def encode(self, x):
        lat = self.queue.pop(0) / 0.18215
        return SimpleNamespace(latent_dist=SimpleNamespace(sample=lambda: lat))

    def decode(self, z):
        return SimpleNamespace(sample=torch.zeros(z.shape[0], 3, 8, 8))


def build_reference_unet(cfg: dict, state_dict: Dict[str, torch.Tensor]):
    UNet, _, _ = import_reference()
    import inspect
    params = inspect.signature(UNet.__init__).parameters
    unet = UNet(**{k: v for k, v in cfg.items() if k in params})
    ref_sd = unet.state_dict()
    assert set(ref_sd) == set(state_dict), (
        f"state-dict key mismatch: only-ref={sorted(set(ref_sd) - set(state_dict))[:5]} "
        f"only-ours={sorted(set(state_dict) - set(ref_sd))[:5]}")
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(state_dict[k].shape), (k, v.shape, state_dict[k].shape)
    unet.load_state_dict(state_dict)
    return unet.eval()


def run_reference_pipeline(unet, inputs: Dict[str, torch.Tensor], n_steps: int, stage: str,
                           guidance_scale: float, image_guidance_scale: float, max_steps: Optional[int] = None,
                           noise_seed: int = 1234) -> List[torch.Tensor]:
    """Latents after every executed step of the reference pipeline (captured through its `callback`).

    `inputs["noise"]` must equal `torch.manual_seed(noise_seed); torch.randn(shape)` because the pipeline draws it
    from the global generator (pipeline.py:409); make_golden.py builds it that way."""
    _, Pipeline, DDIM = import_reference()
    n_ref = inputs["image_prompts"].shape[0]
    n = inputs["latents"].shape[0]
    assert n == 1, "table stand-ins are written for one prompt"
    prompts = ["", "main"] + [f"prev{i}" for i in range(n_ref)]
    table = torch.stack([inputs["uncond"][0], inputs["text"][0]] + [inputs["prev_text"][i][0] for i in range(n_ref)])
    sched = DDIM.from_pretrained(os.path.join(REFERENCE_ROOT, "ckpt/stable-diffusion-v1-5"), subfolder="scheduler")
    vae = _QueueVAE([inputs["zero_prompt"]] + [inputs["image_prompts"][i] for i in range(n_ref)])
    pipe = Pipeline(vae=vae, text_encoder=_TableTextEncoder(table), tokenizer=_TableTokenizer(prompts), unet=unet,
                    scheduler=sched)
    trace: List[torch.Tensor] = []

    class _Stop(Exception):
        pass

    def cb(i, t, latents):
        trace.append(latents.detach().clone())
        if max_steps is not None and len(trace) >= max_steps:
            raise _Stop()

    h, w = inputs["latents"].shape[-2:]
    image_prompt = torch.zeros(1, n_ref, 3, 8 * h, 8 * w)
    torch.manual_seed(noise_seed)
    try:
        pipe(stage=stage, prompt="main", image_prompt=image_prompt, prev_prompt=[f"prev{i}" for i in range(n_ref)],
             height=8 * h, width=8 * w, num_inference_steps=n_steps, guidance_scale=guidance_scale,
             image_guidance_scale=image_guidance_scale, latents=inputs["latents"].clone(), output_type="np",
             callback=cb, callback_steps=1)
    except _Stop:
        pass
    return trace
