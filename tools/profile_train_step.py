#!/usr/bin/env python
"""Per-(kernel family, shape) timing table of one stage-2 TRAINING step (BASELINE config 4: bs 4, 3 reference frames), measured in situ
with HIP events around every instrumented launch of an eager step (storygen_amd.train.UNetTrainer.train_step).  Development tool.
Usage: python tools/profile_train_step.py [path/to/another/libstorygen_hip.so]"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch  # noqa: E402
from storygen_amd import ops  # noqa: E402
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch  # noqa: E402
from storygen_amd.train import UNetTrainer  # noqa: E402


def table(sink, unit, scale):
    rows = defaultdict(lambda: [0, 0.0, 0.0])
    for fam, qty, a, b, shape in sink:
        r = rows[(fam, shape)]
        r[0] += 1
        r[1] += a.elapsed_time(b)
        r[2] += qty
    tot = sum(r[1] for r in rows.values())
    fams = defaultdict(lambda: [0, 0.0])
    for (fam, _), r in rows.items():
        fams[fam][0] += r[0]; fams[fam][1] += r[1]
    print(f"total {tot:.2f} ms; " + "; ".join(f"{f} {v[1]:.2f} ms / {v[0]}" for f, v in sorted(fams.items(), key=lambda kv: -kv[1][1])))
    print(f"{'family':14s} {'shape':48s} {'n':>4s} {'ms':>8s} {'%':>6s} {'us/launch':>10s} {unit:>8s}")
    for (fam, shape), (n, ms, q) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{fam:14s} {shape:48s} {n:4d} {ms:8.3f} {100 * ms / tot:6.1f} {1e3 * ms / n:10.1f} {q / ms / scale:8.1f}")


def main():
    dev = torch.device("cuda", 0)
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, 0)
    batch = synthetic_train_batch(4, 64, arch.config["cross_attention_dim"], 0)
    tr = UNetTrainer(arch, sd, dev, 4, 64, 64, n_ref=3)
    for _ in range(2):
        tr.train_step(batch)
    torch.cuda.synchronize()
    sink, aux = [], []
    ops.PROFILE_SINK, ops.AUX_SINK = sink, aux
    torch.cuda._sleep(400_000_000)
    t0 = time.perf_counter()
    tr.train_step(batch)
    torch.cuda.synchronize()
    print(f"eager instrumented training step: {(time.perf_counter() - t0) * 1e3:.2f} ms (incl. the spin)")
    ops.PROFILE_SINK = ops.AUX_SINK = None
    print("MFMA-class launches")
    table(sink, "TFLOP/s", 1e9)
    print("bandwidth-class launches (instrumented ones only)")
    table(aux, "GB/s", 1e6)


if __name__ == "__main__":
    main()
