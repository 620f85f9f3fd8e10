#!/usr/bin/env python
"""How long do the two halves of a denoising step take ALONE, and together?  (development tool, round 3)

The step's graph runs the main pass of step k beside the reference pass of step k + 1 on two streams.  This replays, as separate
hipGraphs, the main pass alone, the reference pass alone, both back to back on one stream, and both concurrently on two streams:
the gap between max(main, ref) and the concurrent time is what a better schedule could still recover."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.sampler import StoryGenSampler  # noqa: E402
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict  # noqa: E402


def main():
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, 0)
    inputs = synthetic_inputs(1, 3, 64, 64, 0, 768)
    smp = StoryGenSampler(arch, sd, "cuda:0", 1, 64, 64, 3, split_graphs=True)
    smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
    gm, gr = smp.g_main[0], smp.g_ref[0]
    s2 = torch.cuda.Stream()

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def both():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            gr.replay()
        gm.replay()
        torch.cuda.current_stream().wait_stream(s2)

    def seq():
        gr.replay()
        gm.replay()

    print(f"main pass alone      {timed(gm.replay):7.2f} ms")
    print(f"reference pass alone {timed(gr.replay):7.2f} ms")
    print(f"back to back         {timed(seq):7.2f} ms")
    print(f"two streams          {timed(both):7.2f} ms")


if __name__ == "__main__":
    main()
