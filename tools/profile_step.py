#!/usr/bin/env python
"""Per-(kernel family, shape) timing table of one denoising step (BASELINE config 2), measured in situ with HIP
events around every MFMA-class launch of an eager step, plus the eager/graph step times.  Development tool."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import ops  # noqa: E402

ops.apply_env_options()      # SG_* development variables -> sg_debug_set_option
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.sampler import StoryGenSampler  # noqa: E402
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict  # noqa: E402


def main():
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, 0)
    inputs = synthetic_inputs(1, 3, 64, 64, 0, 768)
    G = int(sys.argv[sys.argv.index("--ref-ahead") + 1]) if "--ref-ahead" in sys.argv else 1
    smp = StoryGenSampler(arch, sd, "cuda:0", 1, 64, 64, 3, use_graph=True, ref_ahead=G)
    smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
    for _ in range(2 * G):
        smp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4 * G):
        smp.step()
    torch.cuda.synchronize()
    print(f"graph step (ref_ahead {G}): {(time.perf_counter() - t0) / (4 * G) * 1e3:.2f} ms")
    sink, aux = [], []
    ops.PROFILE_SINK, ops.AUX_SINK = sink, aux
    smp.side_main = smp.side_ref = None   # sequential: per-kernel durations without co-running neighbours
    smp.params.copy_(smp.table[8])        # (every parameter row of a group gets this one: timing, not a trajectory)
    torch.cuda._sleep(200_000_000)   # ~100 ms GPU spin so the (slower) eager host stays ahead of the GPU: no launch gaps
    t0 = time.perf_counter()
    smp._step_body()
    torch.cuda.synchronize()
    print(f"eager instrumented {'group of ' + str(G) + ' steps' if G > 1 else 'step'}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    ops.PROFILE_SINK = ops.AUX_SINK = None
    rows = defaultdict(lambda: [0, 0.0, 0.0])
    for fam, flops, a, b, shape in sink:
        r = rows[(fam, shape)]
        r[0] += 1
        r[1] += a.elapsed_time(b)
        r[2] += flops
    tot = sum(r[1] for r in rows.values())
    print(f"MFMA-class total {tot:.2f} ms")
    print(f"{'family':14s} {'shape':34s} {'n':>4s} {'ms':>8s} {'%':>6s} {'us/launch':>10s} {'TFLOP/s':>8s}")
    for (fam, shape), (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{fam:14s} {shape:34s} {n:4d} {ms:8.3f} {100 * ms / tot:6.1f} {1e3 * ms / n:10.1f} {fl / ms / 1e9:8.1f}")
    rows = defaultdict(lambda: [0, 0.0, 0.0])
    for fam, nbytes, a, b, shape in aux:
        r = rows[(fam, shape)]
        r[0] += 1
        r[1] += a.elapsed_time(b)
        r[2] += nbytes
    tot_aux = sum(r[1] for r in rows.values())
    print(f"bandwidth-class total {tot_aux:.2f} ms (groupnorm / layernorm / copy_rows; event-bracketed, so each includes ~2 us of "
          f"launch gap)")
    for fam in ("groupnorm", "layernorm", "copy_rows"):
        print(f"  {fam}: {sum(r[1] for (f, _), r in rows.items() if f == fam):.3f} ms in {sum(r[0] for (f, _), r in rows.items() if f == fam)} launches")
    print(f"{'family':14s} {'shape':34s} {'n':>4s} {'ms':>8s} {'%':>6s} {'us/launch':>10s} {'GB/s':>8s}")
    for (fam, shape), (n, ms, nb) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"{fam:14s} {shape:34s} {n:4d} {ms:8.3f} {100 * ms / tot_aux:6.1f} {1e3 * ms / n:10.1f} {nb / ms / 1e6:8.0f}")


if __name__ == "__main__":
    main()
