"""ORACLE tooling — golden latents for the N = 2 ordering test (two story frames per GPU, R = 2 prior frames, 32x32 latent, SD-1.5 UNet).

What it pins: the row-major unit order of the batched reference pass, the per-unit noise expansion (noise[n(u)]) and the later steps
of the trajectory for N > 1 — on the default schedule and on the group schedule (tests/test_unet_gpu.py::
test_loop_two_samples_three_steps_vs_oracle_golden_32x32).  Until round 6 that test ran the oracle live (~4 CPU-minutes) and sat
behind SG_SLOW_TESTS, i.e. never under the driver; with the oracle's output on file it costs seconds.
The reference's own UNet cannot run a 32x32 latent (feature keys by latent height): this is the portable restatement
oracle/storygen_oracle.py (pinned to the reference at 64x64 by tests/test_oracle_golden.py), CPU fp32, both stages, 3 steps of the
51-evaluation table the test prepares.

Usage:  python oracle/make_golden_n2.py        (build container only; writes tests/golden/sd15_32_n2_r2.pt)
"""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict  # noqa: E402
from oracle import storygen_oracle as O  # noqa: E402

SEED_W, SEED_IN, HW, N, R, T, STEPS = 0, 9, 32, 2, 2, 51, 3


def main():
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, SEED_W)
    inputs = synthetic_inputs(N, R, HW, HW, SEED_IN, arch.config["cross_attention_dim"])
    out = dict(case="sd15_32_n2_r2", hw=HW, n_samples=N, n_ref=R, table_steps=T, steps=STEPS, seed_weights=SEED_W, seed_inputs=SEED_IN,
               guidance=(7.5, 3.5), latents={}, seconds={}, made_by="oracle/make_golden_n2.py",
               oracle="oracle.storygen_oracle.sample_loop", torch=torch.__version__, threads=torch.get_num_threads())
    with torch.no_grad():
        for stage in ("multi-image-condition", "auto-regressive"):
            t0 = time.time()
            tr = []
            O.sample_loop(sd, arch.config, inputs, T, stage, 7.5, 3.5, max_steps=STEPS, trace=tr)
            out["latents"][stage] = [t.clone() for t in tr]
            out["seconds"][stage] = time.time() - t0
            print(stage, f"{time.time() - t0:.0f}s", [float(t.norm()) for t in tr], flush=True)
    path = os.path.join(ROOT, "tests", "golden", "sd15_32_n2_r2.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
