#!/usr/bin/env python
"""Timing of the networks either side of the loop on the HIP kernels, at the sizes one pipeline call uses them:
CLIP text encoder on [uncond, prompt] + 3 previous prompts (model/pipeline.py:359-362), VAE encode of the zero image and 3 prior frames
at 512x512 (:390-404), VAE decode of one 64x64 latent (:198-205).  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel
table (profiles/r02j_*).  Random weights of the reference's configurations."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd.encoders import ClipTextEngine, VaeEngine, clip_text_param_shapes, init_state, vae_param_shapes  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    vae = VaeEngine(init_state(vae_param_shapes(), 0), dev)
    clip = ClipTextEngine(init_state(clip_text_param_shapes(), 1), dev, heads=12)
    ids = torch.randint(0, 49407, (5, 77))
    frames = torch.rand(4, 3, 512, 512, device=dev)
    z = torch.randn(1, 4, 64, 64, device=dev)
    out = dict(clip_text_5x77_ms=round(timed(lambda: clip(ids)), 3),
               vae_encode_4x512x512_ms=round(timed(lambda: vae.encode(frames)), 3),
               vae_encode_1x512x512_ms=round(timed(lambda: vae.encode(frames[:1])), 3),
               vae_decode_1x64x64_ms=round(timed(lambda: vae.decode(z)), 3))
    # convolution + attention work of AutoencoderKL at 512x512 (2*MAC): decode 1.24 TFLOP, encode 0.57 TFLOP per image
    out["vae_decode_tflops"] = round(1.24 / out["vae_decode_1x64x64_ms"] * 1e3, 1)
    out["vae_encode_tflops"] = round(4 * 0.566 / out["vae_encode_4x512x512_ms"] * 1e3, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
