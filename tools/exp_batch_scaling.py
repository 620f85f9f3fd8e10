#!/usr/bin/env python
"""How does the time of one UNet pass scale with its batch?  (round-3 schedule study: would ONE batch-7 pass — the 3 main samples
and the 4 reference samples of the next step in the same launches — beat the two overlapped passes of the product schedule?)
Times a whole harvesting pass (no early exit) as a hipGraph at batch 1, 2, 3, 4, 7, and the main (consuming) pass at batch 3."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.engine import EngineWeights, HarvestPlan, UNetEngine  # noqa: E402
from storygen_amd.synth import synthetic_state_dict  # noqa: E402


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    arch = build_arch(SD15_CONFIG)
    wts = EngineWeights(arch, synthetic_state_dict(arch, 0), dev)
    sink = UNetEngine(arch, None, dev, 7, 64, 64, 3, 77, weights=wts)          # owns context buffers to harvest into
    for B in (1, 2, 3, 4, 7):
        eng = UNetEngine(arch, None, dev, B, 64, 64, 0, 77, weights=wts)
        eng.set_inputs(torch.randn(B, 4, 64, 64), 50.0, torch.randn(B, 77, 768))
        eng.cache_text_kv()
        plan = HarvestPlan(sink.ctx, [(b, 0, b, 0, 1) for b in range(B)], None)
        side = torch.cuda.Stream(device=dev)
        for mode, kw in (("whole harvesting pass", dict(harvest=plan)), ("harvest-only (early exit)", dict(harvest=plan, harvest_only=True))):
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                eng.forward(text_cache=True, **kw)
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eng.forward(text_cache=True, side=side, **kw)
            print(f"batch {B}: {mode:28s} {timed(g.replay):6.2f} ms", flush=True)
            del g
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
