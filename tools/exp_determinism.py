#!/usr/bin/env python
"""Bisect a schedule-dependent difference (development tool): the same 5 steps under the one-graph schedule, separately launched
graphs, and separately launched graphs with a high-priority main stream — every output should be bit-identical.  Repeats each variant
to tell a race (run-to-run differences) from a plan difference (stable difference between variants).

    python tools/exp_determinism.py [nolat] [nomerge] [nowide] [reps=N] [only=one-graph]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from storygen_amd import _lib  # noqa: E402

_other = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("lib=")), None)     # lib=storygen_amd/lib/libstorygen_hip_<variant>.so
if _other:
    _lib.LIB_PATH = os.path.abspath(_other)
from storygen_amd import engine as E, ops  # noqa: E402
from storygen_amd.arch import SD15_CONFIG, build_arch  # noqa: E402
from storygen_amd.engine import EngineWeights  # noqa: E402
from storygen_amd.sampler import StoryGenSampler  # noqa: E402
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict  # noqa: E402


def main():
    try:
        if "nolat" in sys.argv:
            ops.debug_set_option("lat_tiles", 0)
        if "nowide" in sys.argv:
            ops.debug_set_option("lat_wide", 0)
    except RuntimeError as e:        # an older library without these options
        print("option not available:", e)
    print("development options:", ops.apply_env_options())
    if "nomerge" in sys.argv:
        E.FF_PROJ_MERGE = False
    vt = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("vt=")), None)
    if vt is not None:              # vt=ref:down_blocks.1 -> only the V^T projections of the reference engine's transformers whose prefix contains that string
        which, sub = vt.split(":", 1)
        E.PAIR_GEMMS = False
        E.VT_LAT_FILTER = lambda prefix, consume: (consume == (which == "main") or which == "both") and sub in prefix
        if "wide" in sys.argv:
            E.VT_LAT_TILE = (64, 128, 8)
        if "hash" in sys.argv:
            def rec():
                d = dict(buf=torch.zeros(4096, 5, dtype=torch.int64, device="cuda:0"), ctr=torch.zeros(1, dtype=torch.int64, device="cuda:0"))
                if "images" in sys.argv:
                    d["vts"] = torch.zeros(24, 640 * 1024, dtype=torch.float16, device="cuda:0")
                    d["xs"] = torch.zeros(24, 640 * 1024, dtype=torch.float16, device="cuda:0")
                    d["sts"] = torch.zeros(24, 1024 * 12 * 2, dtype=torch.float32, device="cuda:0")
                return d
            E.VT_HASH = {True: rec(), False: rec()}
            E.VT_PREFIX = sub
        if "check" in sys.argv:
            E.VT_CHECK = torch.zeros((), dtype=torch.int64, device="cuda:0")
            E.VT_MASK = torch.zeros(640, 768, dtype=torch.int32, device="cuda:0")
            E.VT_DIFF = torch.zeros(640, 768, dtype=torch.float32, device="cuda:0")
    if "nopairs" in sys.argv:       # every paired projection as two plain launches (engine.PAIR_GEMMS), free to take the latency kernel
        E.PAIR_GEMMS = False
        E.PAIR_FALLBACK_LAT = True
    reps = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("reps=")), 3)
    hw = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("hw=")), 32)          # hw=64 G=5 R=3: the contract configuration
    G = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("G=")), 1)
    R = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("R=")), 2)
    nsteps = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("steps=")), 5)
    only = next((a.split("=")[1] for a in sys.argv if a.startswith("only=")), None)
    dev = torch.device("cuda:0")
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, 0)
    inputs = synthetic_inputs(1, R, hw, hw, 21, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, dev)
    outs = {}
    snap0 = None
    hash0 = vts0 = None
    for name, kw in (("one graph", dict()), ("split", dict(split_graphs=True)), ("split+priority", dict(split_graphs=True, stream_priority=True)),
                     ("eager", dict(use_graph=False)), ("graph-no-overlap", dict(overlap=False))):
        if only and name.replace(" ", "-") != only:
            continue
        for rep in range(reps):
            smp = StoryGenSampler(arch, None, dev, 1, hw, hw, R, weights=wts, **({"use_graph": True, "ref_ahead": G} | kw))
            smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
            outs[name, rep] = smp.run(max_steps=nsteps).clone()
            torch.cuda.synchronize()
            if E.VT_HASH is not None:
                for consume in (True, False):
                    HS, who = E.VT_HASH[consume], "main" if consume else "reference"
                    n = int(HS["ctr"].item())
                    rows = HS["buf"][:n].cpu()
                    HS["ctr"].zero_()
                    vts = HS["vts"][:n].clone() if "vts" in HS else None
                    if hash0 is None:
                        hash0, vts0 = {}, {}
                    if consume not in hash0 or hash0[consume].shape != rows.shape:
                        hash0[consume], vts0[consume] = rows, vts
                        print(f"{name} rep {rep}: {who} engine: {n} hinted V^T launches recorded (reference for the following repeats)", flush=True)
                        continue
                    d = rows != hash0[consume]          # [n, 5]: raw copy, LayerNorm partials, q|k, V^T, (tokens)
                    if not bool(d.any()):
                        continue
                    kinds = {"same inputs, V^T differs": int((~d[:, 0] & ~d[:, 1] & d[:, 3]).sum()), "same inputs, q|k differs": int((~d[:, 0] & ~d[:, 1] & d[:, 2]).sum()),
                             "raw copy differs": int(d[:, 0].sum()), "LayerNorm partials differ": int(d[:, 1].sum())}
                    i = int(d.any(dim=1).nonzero()[0])
                    print(f"{name} rep {rep}: {who} engine: {kinds}; first differing launch index {i} of {n}: {d[i].tolist()}", flush=True)
                    if vts is None or bool(d[i, 0]) or bool(d[i, 1]):
                        continue
                    M, Cc = int(rows[i, 4]), 640
                    a, b = vts[i, : M * Cc].float().view(Cc, M).cpu(), vts0[consume][i, : M * Cc].float().view(Cc, M).cpu()
                    nz = (a != b).nonzero()
                    rr, cc = sorted(set(nz[:, 0].tolist())), sorted(set(nz[:, 1].tolist()))
                    print(f"   V^T [{Cc}, {M}]: {nz.shape[0]} elements differ: rows {rr[:20]} cols {cc[:70]}")
                    # which run is right, and what does the difference follow?  Recompute the launch on the host in fp64 from ITS inputs.
                    eng = smp.main if consume else smp.ref
                    xf = next(v for k, v in eng.xfs.items() if E.VT_PREFIX in k)
                    x = HS["xs"][i, : M * Cc].double().view(M, Cc).cpu()
                    W, cvec, dvec = xf.w_v1f.double().cpu(), xf.c_v1.double().cpu(), xf.d_v1.double().cpu()
                    mean, var = x.mean(1), x.var(1, unbiased=False)
                    rstd = (var + 1e-5).rsqrt()
                    want = rstd[None, :] * (W @ x.t()) - (mean * rstd)[None, :] * cvec[:, None] + dvec[:, None]
                    r0, c0 = rr[0], cc
                    print(f"      host fp64 {[round(float(want[r0, c]), 5) for c in c0[:6]]}\n      first run {[round(float(b[r0, c]), 5) for c in c0[:6]]}\n      this run  {[round(float(a[r0, c]), 5) for c in c0[:6]]}")
                    dl = [float(a[r0, c] - b[r0, c]) for c in c0]
                    print(f"      diff {[round(v, 5) for v in dl[:8]]}\n      diff / rstd_col {[round(v / float(rstd[c]), 5) for v, c in zip(dl[:8], c0)]}\n      "
                          f"diff / (mean rstd)_col {[round(v / float(mean[c] * rstd[c]), 4) for v, c in zip(dl[:8], c0)]}\n      c_row {float(cvec[r0]):.5f} d_row {float(dvec[r0]):.5f}; "
                          f"d of rows r-4..r+4 {[round(float(dvec[r]), 4) for r in range(max(0, r0 - 4), min(Cc, r0 + 5))]}; c of rows r-4..r+4 {[round(float(cvec[r]), 4) for r in range(max(0, r0 - 4), min(Cc, r0 + 5))]}", flush=True)
            if E.VT_CHECK is not None:
                print(f"{name} rep {rep}: elements of the hinted V^T launches that differ from the 64x64-per-wave kernel beyond rounding: {int(E.VT_CHECK.item())}", flush=True)
                E.VT_CHECK.zero_()
                if E.VT_MASK is not None and int(E.VT_MASK.sum()) > 0:
                    nz = E.VT_MASK.nonzero()
                    print("   mismatching (row, col, count, diff):", [(int(r), int(c), int(E.VT_MASK[r, c]), round(float(E.VT_DIFF[r, c]), 4)) for r, c in nz[:64].tolist()], flush=True)
                    E.VT_MASK.zero_()
            if "buffers" in sys.argv:      # which intermediate buffers differ from the first repeat's?  (reference-pass outputs do not depend on the latents)
                snap = {}
                for i, (c, kv) in enumerate(zip(smp.ctx_sets, smp.kv_sets)):
                    for k in c:
                        snap[f"ctx{i}.{k}"] = c[k].clone()
                        snap[f"k{i}.{k}"], snap[f"vt{i}.{k}"] = kv[k][0].clone(), kv[k][1].clone()
                for l, L in enumerate(smp.main.lv):
                    for k, v in L.items():
                        if v is not None:
                            snap[f"main.lv{l}.{k}"] = v.clone()
                for l, L in enumerate(smp.ref.lv):
                    for k, v in L.items():
                        if v is not None:
                            snap[f"ref.lv{l}.{k}"] = v.clone()
                if rep == 0:
                    snap0 = snap
                else:
                    bad = [k for k in snap if not torch.equal(snap[k], snap0[k])]
                    for k in bad:
                        if k.endswith(".vt") or k.endswith(".qk") or k.endswith(".q2") or k.endswith(".q"):
                            a, b = snap[k].float(), snap0[k].float()
                            d = (a - b).abs()
                            nz = d.nonzero()
                            rows, cols = nz[:, 0].unique(), nz[:, 1].unique()
                            print(f"   {k} {tuple(a.shape)}: {nz.shape[0]} elements differ, max {float(d.max()):.3e}; rows {rows[:12].tolist()}..{rows[-4:].tolist()} ({rows.numel()}), "
                                  f"cols {cols[:12].tolist()}..{cols[-4:].tolist()} ({cols.numel()}); nan in either: {bool(torch.isnan(a).any() or torch.isnan(b).any())}")
                    if bad:
                        print(f"rep {rep}: {len(bad)} buffers differ: ref-pass outputs {[k for k in bad if k[0] in 'ckv']}; ref engine {[k for k in bad if k.startswith('ref.')]}; "
                              f"main engine {[k for k in bad if k.startswith('main.')][:40]}", flush=True)
            del smp
    base = outs[next(iter(outs))]
    for (name, rep), t in outs.items():
        d = float((t - base).abs().max())
        print(f"{name:16s} rep {rep}: max |diff| vs one-graph rep 0 = {d:.3e}{'' if d else '  (bit-identical)'}")


if __name__ == "__main__":
    main()
