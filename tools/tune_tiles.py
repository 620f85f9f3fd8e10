#!/usr/bin/env python
"""Per-shape tile / split-K tuning of sg_gemm_f16 and sg_conv3x3_nhwc_f16 for BASELINE config 2 on the local MI355X.

1. census: one eager denoising step with ops.TUNE_SINK on -> every distinct (shape, epilogue) signature + launch count;
2. for each signature: rebuild synthetic operands of that shape, time the library's own choice and every supported tile
   (x split-K in {auto, 1}) with HIP events (20 back-to-back launches after a GPU spin);
3. write storygen_amd/tuning/mi355x_tiles.json with the winners that beat the heuristic by > 3 %.
A tile hint only changes the schedule (and fp32 summation order); tests/test_kernels_gpu.py covers every tile."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from storygen_amd import ops  # noqa: E402

ops.load_tile_table("none")  # tune from the heuristic tiles, not from a previous table
ops.apply_env_options()      # SG_* development variables -> sg_debug_set_option
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.sampler import StoryGenSampler  # noqa: E402
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
TILES = [(256, 128), (128, 128), (256, 64), (128, 64), (64, 128), (64, 64)]
F16, F32 = torch.float16, torch.float32


G = int(sys.argv[sys.argv.index("--ref-ahead") + 1]) if "--ref-ahead" in sys.argv else 5      # bench.py's default group size


def census():
    """Every GEMM / convolution signature of one unit of the schedule bench.py runs: a group of G steps = the batched reference pass
    of G steps + G main passes (G = 1: one step)."""
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, 0)
    inputs = synthetic_inputs(1, 3, 64, 64, 0, 768)
    smp = StoryGenSampler(arch, sd, dev, 1, 64, 64, 3, use_graph=False, ref_ahead=G)
    smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
    sink = []
    ops.TUNE_SINK = sink
    for _ in range(G):
        smp.step()
    torch.cuda.synchronize()
    ops.TUNE_SINK = None
    seen = {}
    for sig, rec in sink:
        e = seen.setdefault(sig, dict(rec=rec, count=0))
        e["count"] += 1
    del smp
    torch.cuda.empty_cache()
    return seen


def timeit(fn, n=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(8_000_000)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


def rnd(*shape, dtype=F16, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def make_call(rec, ws):
    dt = {None: None, "torch.float16": F16, "torch.float32": F32}
    if rec["kind"] == "gemm":
        M, N, K = rec["M"], rec["N"], rec["K"]
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        n_out = N // 2 if rec["epilogue"] else N
        out = torch.empty(M, n_out, dtype=F32 if rec["out_f32"] else F16, device=dev)
        kw = dict(epilogue=rec["epilogue"], workspace=ws)
        if rec["bias"]:
            kw["bias"] = rnd(N)
        if rec["rowbias"]:
            kw["rowbias"], kw["rows_per_batch"] = rnd(max(1, M // rec["rows_per_batch"]), N, dtype=F32), rec["rows_per_batch"]
        if rec["res1"]:
            kw["res1"] = rnd(M, n_out, dtype=dt[rec["res1"]])
        if rec["res2"]:
            kw["res2"] = rnd(M, n_out, dtype=dt[rec["res2"]])
        if rec["out2"]:
            kw["out2"] = torch.empty(M, n_out, dtype=F16, device=dev)
        if rec.get("ln_out"):
            kw["ln_out"] = torch.empty(M, (N // 64 + 1) & ~1, 2, dtype=F32, device=dev)
        if rec.get("ln"):
            tokens, nvec = (M, N) if rec["ln"] == 1 else (N, M)
            st = torch.zeros(tokens, (K // 64 + 1) & ~1, 2, dtype=F32, device=dev)
            st[:, : K // 64, 1] = 64.0                       # unit variance per block
            kw["ln"] = (rec["ln"], st, rnd(nvec, dtype=F32), rnd(nvec, dtype=F32), 1e-5)
            kw.pop("workspace")
            return lambda tile, split: ops.gemm(a, w, out, tile=tile, split_k=1, **kw)
        return lambda tile, split: ops.gemm(a, w, out, tile=tile, split_k=split, **kw)
    B, H, W, Ci, Co = rec["B"], rec["H"], rec["W"], rec["Cin"], rec["Cout"]
    xp = torch.zeros(B, H + 2, W + 2, Ci, dtype=F16, device=dev)
    xp[:, 1:-1, 1:-1] = rnd(B, H, W, Ci)
    w = rnd(Co, 3, 3, Ci, scale=(9 * Ci) ** -0.5)
    hin = H * 2 if rec["ups"] else H
    Ho = (hin - 1) // rec["stride"] + 1
    out = torch.empty(B, Ho, Ho, Co, dtype=F32 if rec["out_f32"] else F16, device=dev)
    kw = dict(stride=rec["stride"], upsample2x=rec["ups"], workspace=ws, x_padded=True)
    if rec["bias"]:
        kw["bias"] = rnd(Co)
    if rec["rowbias"]:
        kw["rowbias"] = rnd(B, Co, dtype=F32)
    if rec["res1"]:
        kw["res1"] = rnd(B, Ho, Ho, Co, dtype=dt[rec["res1"]])
    assert rec["padded"]
    return lambda tile, split: ops.conv3x3(xp, w, out, tile=tile, split_k=split, **kw)


def main():
    t0 = time.time()
    shapes = census()
    print(f"{len(shapes)} distinct gemm/conv signatures ({sum(e['count'] for e in shapes.values())} launches per group of {G} steps), census {time.time() - t0:.0f}s", flush=True)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    table, log = {}, []
    saved = 0.0
    for sig, e in sorted(shapes.items(), key=lambda kv: -kv[1]["count"]):
        call = make_call(e["rec"], ws)
        base = timeit(lambda: call(None, 0))
        best, best_cfg = base, None
        # split-K: the library's choice (0), none (1), and explicit counts where the K axis is long enough to matter
        kt = (e["rec"]["K"] if e["rec"]["kind"] == "gemm" else 9 * e["rec"]["Cin"]) // 64
        splits = (0, 1) if (kt < 8 or e["rec"].get("ln")) else tuple(s for s in (0, 1, 2, 3, 4, 6, 8) if s <= kt // 2)
        for tile in TILES:
            for split in splits:
                try:
                    us = timeit(lambda: call(tile, split), n=12)
                except RuntimeError:
                    continue
                if us < best:
                    best, best_cfg = us, (tile[0], tile[1], split)
        gain = (base - best) * e["count"]
        line = f"{sig:44s} x{e['count']:3d}  auto {base:7.1f} us  best {best:7.1f} us  {best_cfg}"
        print(line, flush=True)
        log.append(line)
        if best_cfg is not None and best < 0.97 * base:
            table[sig] = list(best_cfg)
            saved += gain
    path = os.path.join(ROOT, "storygen_amd", "tuning", "mi355x_tiles.json")
    kept = 0
    if os.path.exists(path) and "--fresh" not in sys.argv:      # shapes this census did not meet (other group sizes) keep their entries
        with open(path) as f:
            for sig, cfg in json.load(f).get("tiles", {}).items():
                if sig not in shapes:
                    table[sig] = cfg
                    kept += 1
    out = dict(device="MI355X gfx950", made_by="tools/tune_tiles.py", workload=f"BASELINE config 2 (512x512, R=3, N=1), ref_ahead {G}",
               est_saving_us_per_step=round(saved / G, 1), entries_kept_from_previous_table=kept, tiles=table)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"wrote {path}: {len(table)} overrides ({kept} kept from the previous table), ~{saved / G:.0f} us/step (sequential) in {time.time() - t0:.0f}s")
    gdir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(gdir):
        with open(os.path.join(gdir, "mi355x_tiles.json"), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        with open(os.path.join(gdir, "tune_tiles.log"), "w") as f:
            f.write("\n".join(log) + "\n")


if __name__ == "__main__":
    main()
