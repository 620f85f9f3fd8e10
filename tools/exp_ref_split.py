#!/usr/bin/env python
"""TIMING-ONLY experiment (round 3): would the step get faster if the reference pass ran as TWO half-batch engines on two more
graph branches (plus the attn3 K / V^T projections on a fourth)?  The harvested values are not wired up correctly here — the
kernels, shapes and dependencies inside each chain are the real ones, the cross-chain dependencies (K / V after both harvests)
are dropped — so this only answers "how much overlap do more, smaller chains buy"."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.engine import HarvestPlan, UNetEngine, _pair  # noqa: E402
from storygen_amd.sampler import StoryGenSampler  # noqa: E402
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
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
    sd = synthetic_state_dict(arch, 0)
    inputs = synthetic_inputs(1, 3, 64, 64, 0, 768)
    smp = StoryGenSampler(arch, sd, dev, 1, 64, 64, 3)
    smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
    print(f"product schedule (main || batched reference pass): {timed(smp.step, 20):6.2f} ms / step")
    R = 3
    ctx, kv = smp.ctx_sets[1], smp.kv_sets[1]

    def make(nsplit):
        per = 4 // nsplit
        engs = [UNetEngine(arch, None, dev, per, 64, 64, 0, 77, weights=smp.weights) for _ in range(nsplit)]
        plans = []
        for e_i, eng in enumerate(engs):
            eng.set_inputs(torch.randn(per, 4, 64, 64), 50.0, torch.randn(per, 77, 768))
            eng.cache_text_kv()
            hops = []
            for loc in range(per):
                u = e_i * per + loc
                hops.append((loc, 0, 0, 0, R) if u == 0 else (loc, 0, 1, u - 1, 1))
            plans.append(HarvestPlan(ctx, hops, None))
        return engs, plans

    def kv_chain():
        ws, wp = smp.main.ws_split, smp.main.ws_pair
        for key in arch.feature_keys:
            xf = smp.main.xfs[arch.feature_prefix[key]] if hasattr(arch, "feature_prefix") else None
            if xf is None:
                xf = next(x for x in smp.main.xfs.values() if x.spec.feature_key == key)
            c = ctx[key]
            c2d = c.view(c.shape[0] * c.shape[1], c.shape[2])
            ki, vti = kv[key]
            _pair(((c2d, xf.w_k3, ki), dict(workspace=ws)), ((xf.w_v3, c2d, vti), dict(workspace=wp)))

    for nsplit in (1, 2, 4):
        engs, plans = make(nsplit)
        streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit + 1)]

        def body():
            cur = torch.cuda.current_stream(dev)
            for s in streams:
                s.wait_stream(cur)
            for eng, plan, s in zip(engs, plans, streams):
                with torch.cuda.stream(s):
                    eng.forward(harvest=plan, harvest_only=True, text_cache=True)
            with torch.cuda.stream(streams[-1]):
                kv_chain()
            smp._main_pass(0)
            for s in streams:
                cur.wait_stream(s)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        print(f"reference pass as {nsplit} engine(s) of {4 // nsplit} sample(s) + K/V chain on its own branch: {timed(g.replay, 20):6.2f} ms / step")
        del g, engs, plans
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
