#!/usr/bin/env python
"""End-to-end latency of one `StableDiffusionPipeline.__call__` in the reference's default inference setting (inference.py:58-64,103-115,
127-131: 40 DDIM steps, 512x512, guidance 7.0 / 3.5, 3 prior frames) with every network on the HIP kernels: CLIP text encoder on the
prompts, VAE encode of the prior frames, the denoising loop, VAE decode.  Random weights of the reference's configs (no checkpoints here),
a stand-in tokenizer (token ids are irrelevant for timing).  Prints one JSON line; non-contract (bench.py is the contract)."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd.arch import SD15_CONFIG  # noqa: E402
from storygen_amd.model import AutoencoderKL, CLIPTextModel, StableDiffusionPipeline, UNet2DConditionModel  # noqa: E402
from storygen_amd.scheduler import DDIMSchedule  # noqa: E402


class Tok:
    model_max_length = 77

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None):
        n = 1 if isinstance(prompt, str) else len(prompt)
        g = torch.Generator().manual_seed(n)
        ids = torch.randint(1, 49000, (n, 77), generator=g)
        return SimpleNamespace(input_ids=ids, attention_mask=torch.ones_like(ids))


def main():
    dev, f16 = torch.device("cuda:0"), torch.float16
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    unet = UNet2DConditionModel.from_config(SD15_CONFIG).to(dev, f16).eval()
    vae = AutoencoderKL(block_out_channels=(128, 256, 512, 512), down_block_types=("DownEncoderBlock2D",) * 4,
                        up_block_types=("UpDecoderBlock2D",) * 4, layers_per_block=2).to(dev, f16)
    clip = CLIPTextModel(dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12)).to(dev, f16)
    pipe = StableDiffusionPipeline(vae=vae, text_encoder=clip, tokenizer=Tok(), unet=unet, scheduler=DDIMSchedule())
    pipe.set_progress_bar_config(disable=True)
    frames = torch.rand(1, 3, 3, 512, 512)

    def call():
        return pipe(stage="multi-image-condition", prompt="a", image_prompt=frames, prev_prompt=["b", "c", "d"], height=512, width=512,
                    num_inference_steps=steps, guidance_scale=7.0, image_guidance_scale=3.5, output_type="np").images

    call()                                            # builds the sampler, captures the graphs
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        img = call()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({"workload": f"one pipeline call: {steps} DDIM steps, 512x512, 3 prior frames, CFG, HIP CLIP + VAE + UNet, fp16",
                      "seconds_per_image_median": round(ts[1], 4), "seconds_min": round(ts[0], 4), "ms_per_step_incl_everything": round(ts[1] / steps * 1e3, 2),
                      "image_shape": list(img.shape), "finite": bool(torch.isfinite(torch.as_tensor(img)).all())}))


if __name__ == "__main__":
    main()
