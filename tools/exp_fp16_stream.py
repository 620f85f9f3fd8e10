#!/usr/bin/env python
"""Experiment: does a fp16 residual stream INSIDE the transformer blocks (engine.FP16_BLOCK_STREAM) still meet
the 1e-3 latent bar at full depth, and what does it buy?  Runs all 50 DDIM steps of BASELINE config 2 against the reference-made golden
(tests/golden/sd15_64_r3_full.pt) with the flag off and on, and times 20 steps of each.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storygen_amd import engine  # noqa: E402
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.sampler import StoryGenSampler  # noqa: E402
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    dev = torch.device("cuda:0")
    gold = torch.load(os.path.join(ROOT, "tests", "golden", "sd15_64_r3_full.pt"), weights_only=False)
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, 0)                      # the weights every golden was made with (tests/test_unet_gpu.py::sd15)
    R, hw = gold["n_ref"], gold["hw"]
    inputs = synthetic_inputs(1, R, hw, hw, gold["seed"], arch.config["cross_attention_dim"])
    want = gold["stages"]["multi-image-condition"]["latents"]
    out = {}
    if "ref" in sys.argv:      # round 6: fp16 stream in the REFERENCE engine only (batch 20, group schedule), main pass fp32 as shipped
        variants = (("fp32_stream", None), ("ref_fp16_block", (True, False)), ("ref_fp16_resnet", (False, True)), ("ref_fp16_both", (True, True)))
    else:
        variants = (("fp32_stream", False, False), ("fp16_block_stream", True, False), ("fp16_whole_stream", True, True))
    for var in variants:
        name = var[0]
        if "ref" in sys.argv:
            smp = StoryGenSampler(arch, sd, dev, 1, hw, hw, R, ref_ahead=5, ref_fp16_stream=var[1])
        else:
            engine.FP16_BLOCK_STREAM, engine.FP16_RESNET_STREAM = var[1], var[2]
            smp = StoryGenSampler(arch, sd, dev, 1, hw, hw, R)
        smp.prepare(inputs, gold["n_steps"], "multi-image-condition", *gold["guidance"])
        trace = []
        smp.run(trace=trace)
        torch.cuda.synchronize()
        errs = [rel(a.cpu(), b) for a, b in zip(trace, want)]
        smp.prepare(inputs, gold["n_steps"], "multi-image-condition", *gold["guidance"])
        for _ in range(5):
            smp.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            smp.step()
        torch.cuda.synchronize()
        out[name] = dict(err_step_1=errs[0], err_step_10=errs[9], err_step_25=errs[24], err_step_50=errs[49],
                                                                   err_max=max(errs), ms_per_step=round((time.perf_counter() - t0) / 20 * 1e3, 3))
        del smp
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
