#!/usr/bin/env python
"""bench.py — StoryGen denoising hot loop on MI355X: denoising steps/s at 512x512, 3 prior-frame context, bs=1.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one iteration of /root/reference/model/pipeline.py:412-461 with classifier-free guidance: the reference
passes over the R=3 prior frames + 1 main pass (batch 3, attn3 over 12 288 context tokens) + CFG + DDIM.  As written
that is 11.05 TFLOP of conv/GEMM/attention contractions; the sampler computes each distinct reference
sample once (4 instead of 9 sample-forwards in this mode) with identical results, so it EXECUTES less —
`tflop_per_step_executed` is measured (sum of the launches' algorithmic FLOPs) and is the only figure any
utilisation number below is derived from.
Workload = BASELINE.json configs[1] ("inference.py: 50-step DDIM, 512x512, 3 prior-frame context, fp16,
1xMI355X"), SD-1.5-architecture UNet (909 M params) with synthetic fp16 weights and synthetic inputs
(storygen_amd/synth.py) — there is no network for checkpoints.  N GPUs = N independent samples, one per GPU
(weak scaling, no per-step communication; one RCCL all-gather of the final latents after the loop).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — the kernel family with the largest share of the step, timed in situ with HIP events on the launch
                 stream during one extra eager step after the timed region: achieved = algorithmic FLOPs of its
                 launches / their summed duration; peak = 2500 TFLOP/s dense fp16 MFMA
                 (/opt/skills/guides/MI355X_MICROARCH.md).  `families` lists every MFMA-class family the same way.
  cpu_baseline — the oracle (oracle/storygen_oracle.py, kind "port") timed on the host cores on a bounded sample:
                 one reference pass + one main pass (batch 3, R=3); step time = 3*t_ref + t_main.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0          # dense MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
HW, R, N_PER_GPU, T = 64, 3, 1, 50
DEFAULT_REF_AHEAD = 5               # reference passes of 5 consecutive steps as one batched UNet call inside one hipGraph per group
REF_GF, MAIN_GF = 803.3, 1273.1    # per-sample algorithmic GFLOP of one ref / main pass at 64x64, R=3
STEP_TFLOP = 3 * (R * REF_GF + MAIN_GF) / 1000.0


def _one_socket_cores():
    """(physical cores of one socket, logical CPUs visible to this process) from /proc/cpuinfo; falls back to the affinity mask."""
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        cores = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = int(line.split(":")[1])
                elif line.startswith("core id"):
                    core = int(line.split(":")[1])
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        sockets = len({p for p, _ in cores}) or 1
        per_socket = len(cores) // sockets
        if per_socket >= 1:
            return min(per_socket, nproc), nproc
    except OSError:
        pass
    return max(1, nproc // 2), nproc


def cpu_baseline(arch, sd, inputs):
    """The oracle (kind "port") on the host cores, bounded sample: one reference pass + one main pass (batch 3, R = 3), best of 2
    each, on the physical cores of ONE socket (oversubscribing every logical CPU made the same pass slower: 83.9 s per step on 128
    threads in round 3 against the reference's own loop at 53.9 s on 6)."""
    from oracle import storygen_oracle as O
    cfg = arch.config
    sd = {k: v.float() for k, v in sd.items()}        # the cached state dict is stored as fp16 (exact); the oracle computes in fp32
    threads, nproc = _one_socket_cores()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    sched = O.DDIM()
    t_main = sched.timesteps(T)[0]
    ref_t = t_main // 10
    an = sched.add_noise
    x = torch.cat([an(inputs["zero_prompt"], inputs["noise"], ref_t), an(inputs["image_prompts"][0], inputs["noise"], ref_t),
                   an(inputs["image_prompts"][0], inputs["noise"], ref_t)])
    e = torch.cat([inputs["prev_uncond"][0], inputs["prev_text"][0], inputs["prev_text"][0]])
    try:
        with torch.no_grad():
            t_refs, t_mains = [], []
            for _ in range(2):
                t0 = time.perf_counter()
                _, feats = O.unet_forward(sd, cfg, x, ref_t, e, None)
                t_refs.append(time.perf_counter() - t0)
            ctx = {k: torch.cat([v] * R, dim=1) for k, v in feats.items()}
            xm = torch.cat([inputs["latents"]] * 3)
            em = torch.cat([inputs["uncond"], inputs["uncond"], inputs["text"]])
            for _ in range(2):
                t0 = time.perf_counter()
                O.unet_forward(sd, cfg, xm, t_main, em, ctx)
                t_mains.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(prev_threads)
    t_ref, t_main_s = min(t_refs), min(t_mains)
    step_s = R * t_ref + t_main_s
    return {"value": 1.0 / step_s, "unit": "denoising steps/s", "cores": threads, "nproc": nproc, "kind": "port",
            "sample": f"oracle fp32 on {threads} threads (physical cores of one socket; {nproc} logical CPUs visible), best of 2: 1 ref pass "
                      f"({t_ref:.1f}s; runs {', '.join(f'{t:.1f}' for t in t_refs)}) + 1 main pass ({t_main_s:.1f}s; runs "
                      f"{', '.join(f'{t:.1f}' for t in t_mains)}), batch 3, R=3; step = 3*t_ref + t_main = {step_s:.1f}s", "seconds_per_step": step_s,
            # the reference's OWN pipeline loop (model/pipeline.py on the diffusers shim, CPU fp32) over all 50 steps of this
            # workload when the full-depth golden was made in the build container (oracle/make_golden.py sd15_64_r3_full,
            # gpurun_out/golden_full.log: 2 693 s / 50 steps on 6 threads) — quoted, not re-timed here (it cannot travel)
            "reference_pipeline_seconds_per_step": 53.9, "reference_pipeline_threads": 6}


def measured_traffic(kernel: str):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this same command (FETCH_SIZE x2 per the gfx950
    correction of MI355X_MICROARCH.md §HBM, + WRITE_SIZE), committed with their provenance in profiles/traffic.json;
    None when no such measurement is on file."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    e = t.get("kernels", {}).get(kernel)
    return None if e is None else e["hbm_bytes_per_launch"]


def traffic_provenance():
    """'current' when profiles/traffic.json was measured on a build of exactly these kernel sources, else 'stale'/'none'."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return "none"
    from storygen_amd.build import source_hash
    with open(path) as f:
        return "current" if json.load(f).get("kernel_source_hash") == source_hash() else "stale"


def in_situ_roofline(sampler, dump_algorithmic=None):
    """One extra eager step with every MFMA-class launch bracketed by HIP events on its launch stream.  Every GEMM / convolution
    launch also reports the profiler class (kernel instantiation, grid) it falls into and its ALGORITHMIC bytes (ops.PLAN_SINK);
    dump_algorithmic = path: the per-class table goes there for tools/traffic_from_pmc.py to set beside the measured HBM bytes."""
    from storygen_amd import ops
    sink, aux, plans = [], [], []
    ops.PROFILE_SINK, ops.AUX_SINK, ops.PLAN_SINK = sink, aux, plans
    sides = (sampler.side_main, sampler.side_ref)
    sampler.side_main = sampler.side_ref = None     # one stream: per-kernel durations without co-running neighbours
    try:
        sampler.params.copy_(sampler.table[min(sampler.k, sampler.table.shape[0] - 1)], non_blocking=True)
        # a GPU-side spin first, so the slower eager host stays ahead of the device and every kernel starts right
        # after its start event (otherwise host launch latency would be billed to short kernels)
        torch.cuda._sleep(200_000_000)
        sampler._step_body()
        torch.cuda.synchronize()
    finally:
        ops.PROFILE_SINK = ops.AUX_SINK = ops.PLAN_SINK = None
        sampler.side_main, sampler.side_ref = sides
    classes = {}
    for kname, grid, nbytes, family, shape in plans:
        c = classes.setdefault(f"{kname}|{grid}", {"kernel": kname, "grid": grid, "launches": 0, "algorithmic_bytes": 0.0, "shapes": {}})
        c["launches"] += 1
        c["algorithmic_bytes"] += nbytes
        c["shapes"][f"{family} {shape}"] = c["shapes"].get(f"{family} {shape}", 0) + 1
    if dump_algorithmic:
        with open(dump_algorithmic, "w") as f:
            json.dump({"note": "algorithmic bytes (every operand once: A, W, C, residuals, second output) of the GEMM / convolution launches "
                               "of one instrumented group of steps, per profiler class (kernel instantiation | grid size in threads)",
                       "steps_in_sample": getattr(sampler, "G", 1), "classes": classes}, f, indent=1)
    fam = {}
    for name, flops, a, b, _shape in sink:
        f = fam.setdefault(name, {"launches": 0, "ms": 0.0, "gflop": 0.0})
        f["launches"] += 1
        f["ms"] += a.elapsed_time(b)
        f["gflop"] += flops / 1e9
    for f in fam.values():
        f["tflops"] = f["gflop"] / f["ms"] if f["ms"] > 0 else 0.0
        f["avg_us"] = 1e3 * f["ms"] / f["launches"]
        f["frac_of_peak"] = f["tflops"] / PEAK_FP16_TFLOPS
    # kernels, not call sites: sg_gemm_f16 and sg_conv3x3_nhwc_f16 are the same device mainloop (mma_pipe_body: mma_pipe_kernel with 64x64 per
    # wave, mma_lat_kernel — round 6 — with 32x32 per wave on a deeper ring), the three
    # attention head dims are instantiations of attn_fwd_kernel
    kernels = {}
    for name, f in fam.items():
        kname = ("mma_pipe_body (gemm + conv3x3: mma_pipe_kernel / mma_lat_kernel)" if name in ("gemm", "conv3x3") else
                 "ff_fused_kernel" if name == "ff_fused" else "attn_fwd_kernel")
        k = kernels.setdefault(kname, {"launches": 0, "ms": 0.0, "gflop": 0.0})
        for key in k:
            k[key] += f[key]
    for k in kernels.values():
        k["tflops"] = k["gflop"] / k["ms"] if k["ms"] > 0 else 0.0
        k["avg_us"] = 1e3 * k["ms"] / k["launches"]
        k["frac_of_peak"] = k["tflops"] / PEAK_FP16_TFLOPS
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    d = kernels[dom]
    g = getattr(sampler, "G", 1)      # ref_ahead: the instrumented body is one group = g steps (one batched reference pass)
    executed_tflop = sum(f["gflop"] for f in fam.values()) / 1e3 / g
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(d["tflops"], 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(d["frac_of_peak"], 4), "traffic": measured_traffic(dom), "traffic_provenance": traffic_provenance(),
            "algorithmic_bytes_per_launch": (round(sum(p[2] for p in plans) / len(plans)) if plans else None),
            "launches_per_step": round(d["launches"] / g, 2),
            "avg_launch_us": round(d["avg_us"], 1), "gflop_per_step": round(d["gflop"] / g, 1), "steps_in_sample": g,
            "hbm_families": hbm_families(aux, g),
            "kernels": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in sorted(kernels.items())},
            "families": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in sorted(fam.items())}}
    return roof, executed_tflop


def hbm_families(aux, g):
    """The bandwidth-class kernels (GroupNorm, LayerNorm, feature copies) of the instrumented step: algorithmic bytes / event time
    per family (each event pair includes ~2 us of launch gap, so these are lower bounds on the kernels' own GB/s)."""
    fam = {}
    for name, nbytes, a, b, _shape in aux:
        f = fam.setdefault(name, {"launches": 0, "ms": 0.0, "mbytes": 0.0})
        f["launches"] += 1
        f["ms"] += a.elapsed_time(b)
        f["mbytes"] += nbytes / 1e6
    for f in fam.values():
        f["hbm_gbps"] = round(f["mbytes"] / f["ms"], 1) if f["ms"] > 0 else 0.0
        f["frac_of_8tbps"] = round(f["hbm_gbps"] / 8000.0, 4)
        f["launches"] = round(f["launches"] / g, 2)
        f["ms"], f["mbytes"] = round(f["ms"] / g, 3), round(f["mbytes"] / g, 1)
    return fam


def train_step_bench(args):
    """BASELINE configs[3] (non-contract line): storygen_amd.train.UNetTrainer.train_step, eager (no hipGraph yet)."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X")
    from storygen_amd.arch import SD15_CONFIG, build_arch
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd import train as _train
    from storygen_amd.train import UNetTrainer
    _train.SPLITK_WORKSPACE = not args.train_no_splitk_workspace
    dev = torch.device("cuda", 0)
    arch = build_arch(SD15_CONFIG)
    sd = synthetic_state_dict(arch, 0)
    bs = 4
    batch = synthetic_train_batch(bs, HW, arch.config["cross_attention_dim"], 0)
    if args.optimizer != "none":
        # the whole loop body of train_StorySalon_stage2.py:304-333: gradients, global-norm clip, AdamW / 8-bit AdamW, refreshed weights
        from storygen_amd.model import UNet2DConditionModel
        from storygen_amd.training import Stage2Trainer
        unet = UNet2DConditionModel.from_config(SD15_CONFIG)
        unet.load_state_dict(sd)
        st2 = Stage2Trainer(unet.to(dev, torch.float32), bs, HW, HW, n_ref=R, use_8bit_adam=args.optimizer == "adamw8bit",
                            use_graph=not args.no_graph)
        tr = st2.trainer

        def step(b):
            out = st2.step(b, use_refs=(0, 1, 2))
            return out["loss"], st2.named
    else:
        tr = UNetTrainer(arch, sd, dev, bs, HW, HW, n_ref=R)
        step = tr.train_step if args.no_graph else tr.train_step_graph      # default: the whole step replayed as one hipGraph
    for _ in range(args.warmup + (0 if args.no_graph else 1)):          # (+1: the capturing call)
        loss, grads = step(batch)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, grads = step(batch)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    # forward FLOPs: 3 reference passes + main pass per sample; backward of the main pass counted as 2x its forward for the
    # dgrads + the attention recompute (SURVEY §8d config 4 estimate: ~23 TFLOP per step at bs=4)
    fwd = bs * (R * REF_GF + MAIN_GF) / 1000.0
    # executed FLOPs of one step, counted: one extra EAGER step with every MFMA-class launch reporting its algorithmic FLOPs
    # (GEMM / conv forward, dgrads and weight gradients, attention forward with 4 and backward with 6 + 8 N_q N_k D per head)
    from storygen_amd import ops as _ops
    sink = []
    _ops.PROFILE_SINK = sink
    try:
        tr.train_step(batch)
        torch.cuda.synchronize(dev)
    finally:
        _ops.PROFILE_SINK = None
    fam = {}
    for name, flops, a, b, _shape in sink:
        f = fam.setdefault(name, {"launches": 0, "ms": 0.0, "gflop": 0.0})
        f["launches"] += 1
        f["ms"] += a.elapsed_time(b)
        f["gflop"] += flops / 1e9
    executed = sum(f["gflop"] for f in fam.values()) / 1e3
    step_s = dt / args.steps
    roof = {"bound": "mfma", "kernel": "whole training step (all MFMA-class launches)", "achieved": round(executed / step_s, 1),
            "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s", "frac": round(executed / step_s / PEAK_FP16_TFLOPS, 4), "traffic": None,
            "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "gflop": round(v["gflop"], 1),
                             "tflops": round(v["gflop"] / v["ms"], 1) if v["ms"] > 0 else 0.0} for k, v in sorted(fam.items())}}
    print(json.dumps({"metric": "stage-2 training steps/sec @512x512, bs=4, 3 reference frames (non-contract)",
                      "value": round(args.steps / dt, 4), "unit": "it/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f16", "data": "synthetic",
                      "config": {"workload": "NON-CONTRACT RUN, BASELINE configs[3]: train_StorySalon_stage2.py step, bs=4, fp16 operands / "
                                             "fp32 residual stream and gradients, attn3 gradients only; "
                                             + ("eager launches" if args.no_graph else "whole step = one hipGraph replay, loss-scaled fp16 gradient operands"),
                                 "hipgraph": not args.no_graph, "grad_scale": tr.last_grad_scale, "gradients": len(grads), "loss": float(loss),
                                 "optimizer": args.optimizer + ("" if args.optimizer == "none" else " + clip_grad_norm_(1.0), in the timed step")},
                      "tflop_forward_per_step": round(fwd, 3), "tflop_per_step_executed": round(executed, 3), "roofline": roof}), flush=True)


def default_ref_ahead(args) -> int:
    """The group size of the look-ahead schedule when --ref-ahead is not given: the largest G <= DEFAULT_REF_AHEAD dividing --steps."""
    if args.ref_ahead is not None:
        return max(1, args.ref_ahead)
    if args.no_graph or args.no_overlap:
        return 1
    return max(g for g in range(1, DEFAULT_REF_AHEAD + 1) if args.steps % g == 0)


def cached_state_dict(arch, seed: int, rank: int, use_dist: bool):
    """synth.synthetic_state_dict(arch, seed), synthesised ONCE per node: the first local rank draws the 909 M parameters (a minute of
    host time) and stores them as fp16 — lossless, the values are fp16-exact — under the temp directory; every other rank (and every
    later bench.py run on the box) memory-maps the file.  Before round 6 each of N ranks synthesised all of it, sharing the host's
    cores.  The file name carries a hash of the architecture, the seed and synth.py itself."""
    import hashlib
    import tempfile
    from storygen_amd import synth
    with open(synth.__file__, "rb") as f:
        src = f.read()
    key = hashlib.sha256(repr((sorted(arch.config.items(), key=str), seed)).encode() + src).hexdigest()[:16]
    path = os.path.join(os.environ.get("SG_CACHE_DIR", tempfile.gettempdir()), f"storygen_amd_synth_{key}.pt")
    first = int(os.environ.get("LOCAL_RANK", rank)) == 0
    sd = None
    if first and not os.path.exists(path):
        sd = synth.synthetic_state_dict(arch, seed)
        try:
            tmp = f"{path}.{os.getpid()}.tmp"
            torch.save({k: v.to(torch.float16) for k, v in sd.items()}, tmp)
            os.replace(tmp, path)
        except OSError:
            pass                                     # no cache then: the other ranks fall back to synthesising
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
    if sd is None:
        try:
            sd = torch.load(path, mmap=True, weights_only=True)
        except Exception:
            sd = synth.synthetic_state_dict(arch, seed)
    return sd


def timed_steps(sampler, steps: int, warmup_run: int, use_dist: bool, dev):
    """The contract's timed region: warmup_run untimed steps, then EXACTLY `steps` steps bracketed by a barrier + device
    synchronisation on both sides; returns the seconds of the slowest rank.  dev = None: no device to synchronise (--dry-run)."""
    import torch.distributed as dist

    def barrier():
        if use_dist:
            dist.barrier()
        if dev is not None:
            torch.cuda.synchronize(dev)

    for _ in range(warmup_run):
        sampler.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        sampler.step()
    barrier()
    dt = time.perf_counter() - t0
    timed_steps.by_rank = [dt]
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dev is not None else "cpu")
        every = [torch.zeros_like(tt) for _ in range(dist.get_world_size())]
        dist.all_gather(every, tt)                  # per-rank times: a straggler shows in the line (ms_per_step_by_rank)
        timed_steps.by_rank = [float(t.item()) for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def timed_loop(sampler, inputs, n_steps: int, stage: str, use_dist: bool, dev, repeats: int = 2):
    """One whole image: prepare() (text K / V, time-embedding tables, the primer reference pass — nothing overlaps with it) + run() of
    all n_steps steps, the last group without its look-ahead pass — /root/reference/model/pipeline.py:411-469 from the first scheduler
    call to the last step.  Graphs are already captured (the timed region ran before).  Returns the milliseconds of each repeat
    (slowest rank) and the host milliseconds prepare() took to return."""
    import torch.distributed as dist

    def barrier():
        if use_dist:
            dist.barrier()
        if dev is not None:
            torch.cuda.synchronize(dev)

    out, prep = [], []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        sampler.prepare(inputs, n_steps, stage, 7.5, 3.5)
        prep.append(1e3 * (time.perf_counter() - t0))
        sampler.run()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if dev is not None else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        out.append(1e3 * dt)
    return out, prep


def gather_and_check(sampler, world: int, dev):
    """The one collective of the data-parallel path (all-gather of the final latents) and the checks on its result: one sample per
    rank, all finite, pairwise different (every rank denoised its own, rank-seeded sample).  Returns (final, gather ms, finite)."""
    from storygen_amd.sampler import gather_latents
    t0 = time.perf_counter()
    final = gather_latents(sampler.latents)
    if dev is not None:
        torch.cuda.synchronize(dev)
    gather_ms = 1e3 * (time.perf_counter() - t0)
    finite = bool(torch.isfinite(final).all())
    assert final.shape[0] == world * N_PER_GPU, f"all-gather returned {final.shape[0]} samples for {world} ranks"
    distinct = all(not torch.equal(final[i], final[j]) for i in range(final.shape[0]) for j in range(i))
    assert distinct, "two ranks produced identical latents: the batch shard is not per-rank"
    return final, gather_ms, finite


def dry_run(args, world: int, rank: int):
    """--dry-run: everything of main() that is not the GPU — ranks from the launcher's environment, process group (gloo), group
    schedule, barrier + max-over-ranks timing, the one all-gather, the per-rank-distinct assertion, one JSON line from rank 0 — on
    a stand-in engine (tests/stub_engine.py).  Measures nothing and says so."""
    import types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import stub_engine
    import storygen_amd.sampler as S
    from storygen_amd.arch import build_arch, load_config
    S.UNetEngine = stub_engine.StubEngine
    S.ops = types.SimpleNamespace(add_noise=stub_engine.add_noise, cfg_ddim_step=stub_engine.cfg_ddim_step)
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    arch = build_arch(load_config(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                                       up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=8, attention_head_dim=4,
                                       norm_num_groups=8)))
    g = torch.Generator().manual_seed(1000 + rank)                      # inputs seeded by rank, like synthetic_inputs(seed=rank)
    r = lambda *sh: torch.randn(*sh, generator=g)                      # noqa: E731
    hw, S_ = 4, 5
    inputs = dict(latents=r(1, 4, hw, hw), noise=r(1, 4, hw, hw), image_prompts=r(R, 1, 4, hw, hw), zero_prompt=r(1, 4, hw, hw),
                  text=r(1, S_, 8), uncond=r(1, S_, 8), prev_text=r(R, 1, S_, 8), prev_uncond=r(1, 1, S_, 8).expand(R, 1, S_, 8).clone())
    G = default_ref_ahead(args)
    if G > 1 and args.steps % G:
        raise SystemExit(f"--ref-ahead {G} must divide --steps {args.steps}")
    warmup_run = -(-args.warmup // G) * G
    sampler = S.StoryGenSampler(arch, None, "cpu", 1, hw, hw, R, S_, use_graph=False, weights=object(), ref_ahead=G, time_tables=False)
    sampler.prepare(inputs, -(-max(T, args.steps + warmup_run) // G) * G, args.stage, 7.5, 3.5)
    dt = timed_steps(sampler, args.steps, warmup_run, use_dist, None)
    final, _, _ = gather_and_check(sampler, world, None)
    distinct = True
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN of bench.py's launcher path on the CPU (gloo, stand-in engine): not a measurement", "dry_run": True,
                          "value": None, "unit": "denoising steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": None, "host_ms_per_stub_step_max_over_ranks": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "stand-in engine, 4x4 latent", "ref_ahead": G, "warmup_run": warmup_run, "stage": args.stage,
                                     "parallelism": f"dp{world} (one sample per rank, final all-gather)"},
                          "latents_gathered": int(final.shape[0]), "latents_finite": bool(torch.isfinite(final).all()),
                          "latents_distinct_per_rank": distinct}), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="TEST ONLY, measures nothing: the launcher path (torch.distributed.run ranks, barrier, max-over-ranks timing, the "
                         "final all-gather, the per-rank-distinct check, the JSON line) on the CPU with the gloo backend and a stand-in "
                         "engine (tests/stub_engine.py) on a tiny UNet description; the JSON says dry_run and carries value null")
    ap.add_argument("--dump-algorithmic", default=None, metavar="PATH",
                    help="write the algorithmic bytes per profiler class (kernel instantiation | grid) of the instrumented step to PATH "
                         "(tools/traffic_from_pmc.py merges it into profiles/traffic.json)")
    ap.add_argument("--stage", choices=("multi-image-condition", "auto-regressive"), default="multi-image-condition",
                    help="multi-image-condition = the contract line (SURVEY 8d); auto-regressive = the mode /root/reference/inference.py:133 "
                         "defaults to (every prior frame at its own noise level: 2R distinct reference samples per step instead of R + 1) — "
                         "a NON-CONTRACT line, named as such in the JSON")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-dedup", action="store_true", help="run all 3R reference samples as written")
    ap.add_argument("--no-overlap", action="store_true", help="one stream: reference pass, then main pass")
    ap.add_argument("--ref-ahead", type=int, default=None,
                    help="batch the reference passes of G consecutive steps into one UNet call (sampler ref_ahead): one hipGraph per "
                         "group of G steps, the batched reference pass of the next group forked beside the group's G main passes. The "
                         "timed window starts on a group boundary (extra untimed warm-up steps, reported as warmup_run) and G must "
                         "divide --steps, so that it contains exactly steps/G batched reference passes and steps main passes.  Default: "
                         f"the largest G <= {DEFAULT_REF_AHEAD} that divides --steps (same-box A/B of G = 1 / 2 / 5 / 10: "
                         "profiles/r05a_ab_ref_ahead_one_graph.txt)")
    ap.add_argument("--config5-shape", action="store_true",
                    help="NOT the contract workload: BASELINE configs[4]'s shape (768x768 = 96x96 latent, 5 prior frames); fp16 "
                         "attention unless --fp8-attention; the JSON names it in config.workload")
    ap.add_argument("--split-graphs", action="store_true",
                    help="A/B: reference and main pass as separately launched hipGraphs on two streams (measured: they do not overlap)")
    ap.add_argument("--ref-cus", type=int, default=0,
                    help="EXPERIMENT: launch the batched reference pass eagerly (no graph) on a stream restricted to this many CUs "
                         "(hipExtStreamCreateWithCUMask), one group ahead of the main-pass graphs; 0 = off")
    ap.add_argument("--ref-cu-layout", choices=("first", "spread"), default="first")
    ap.add_argument("--ref-eager", action="store_true", help="EXPERIMENT: the reference pass kernel by kernel from the host instead of a graph replay")
    ap.add_argument("--stream-priority", action="store_true",
                    help="with --split-graphs / --ref-ahead: main-pass graphs on a high-priority stream")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="BASELINE configs[4]'s attention path: head-dim-40 image / self attention on the e4m3 MFMA kernel (with "
                         "--config5-shape; results differ from the fp16 path by fp8 rounding, so never the contract line)")
    ap.add_argument("--no-gn-epilogue", action="store_true", help="A/B: every GroupNorm makes its own statistics pass")
    ap.add_argument("--fp16-block-stream", action="store_true",
                    help="A/B (changes results, never the contract line): fp16 residual stream inside the transformer blocks")
    ap.add_argument("--no-gemm-pairs", action="store_true", help="A/B: q|k + V^T, q2 + q3, k3 + v3^T as separate launches")
    ap.add_argument("--attn-pair", action="store_true", help="A/B: text and image cross-attention of a block in one launch")
    ap.add_argument("--no-loop", action="store_true", help="skip the whole-image line (loop_50_steps_ms: prepare() + all 50 steps, two repeats)")
    ap.add_argument("--no-time-tables", action="store_true",
                    help="A/B switch: every UNet call recomputes the time-embedding chain (as written) instead of reading the rows tabulated at prepare()")
    ap.add_argument("--no-short-rows", action="store_true",
                    help="A/B: the zero-image context rows keep R copies of their feature map, as written (default: one copy — softmax over R "
                         "copies of the same keys equals softmax over one)")
    ap.add_argument("--no-splitk-in-gn", action="store_true",
                    help="A/B: split-K convolutions at the 16x16 / 8x8 levels run their own second pass instead of leaving it to the GroupNorm")
    ap.add_argument("--no-ff-fused", action="store_true", help="A/B: GEGLU feed-forward of the 64x64 level as two GEMM launches")
    ap.add_argument("--ref-fp32-stream", action="store_true",
                    help="A/B switch: the reference engine's residual stream in fp32 like the main engine's (default since round 6: fp16 — its outputs are fp16 features)")
    ap.add_argument("--no-ff-proj-merge", action="store_true",
                    help="A/B switch: ff.net.2 and proj_out as two GEMMs (as written) instead of one K = 5C GEMM at C = 640 / 1280 (engine.FF_PROJ_MERGE)")
    ap.add_argument("--no-ff-split", action="store_true",
                    help="A/B: the fused feed-forward of a small launch (<= 16 k tokens) as one workgroup per 128 tokens (default: the hidden "
                         "units split over two workgroups, partial sums added by proj_out's contraction)")
    ap.add_argument("--no-shared-head", action="store_true",
                    help="A/B: the main pass runs its three CFG samples at batch 3 from conv_in on, as written (default: everything up to "
                         "the first cross-attention once — the samples share latent and timestep)")
    ap.add_argument("--optimizer", choices=("none", "adamw", "adamw8bit"), default="none",
                    help="with --train-step: include the reference's clip_grad_norm_ + optimizer step (storygen_amd.training.Stage2Trainer)")
    ap.add_argument("--train-no-splitk-workspace", action="store_true",
                    help="with --train-step (development A/B): the training classes without split-K scratch, as in rounds 2 - 5")
    ap.add_argument("--train-step", action="store_true",
                    help="NOT the contract workload: BASELINE configs[3] — stage-2 training step, bs=4, 512x512, 3 reference frames "
                         "(forward of 3 reference passes + main pass, backward of the main pass, 80 attn3 gradients); reports it/s")
    args = ap.parse_args()
    if args.train_step:
        return train_step_bench(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (RCCL)
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if world != args.gpus:
        args.gpus = world
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no GPU visible (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # one process per GPU under torch.distributed.run (RCCL); a torchrun launch with a single rank still goes through the
    # same collectives, so the N > 1 code path can be exercised on a 1-GPU box
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if world > 1:
        # N ranks synthesise the same 909 M parameters on the host at once: give each its share of the logical CPUs instead of
        # letting every rank start a thread per CPU
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, ncpu // world))

    from storygen_amd.arch import SD15_CONFIG, build_arch
    from storygen_amd.sampler import StoryGenSampler, gather_latents
    from storygen_amd.synth import synthetic_inputs, synthetic_state_dict
    if os.environ.get("SG_DEV_OPTIONS") == "1":          # A/B runs of development options (tools/next_round/*.sh): never the default
        from storygen_amd import ops
        print("development options:", ops.apply_env_options(), file=sys.stderr)
    if args.no_gemm_pairs:
        from storygen_amd import engine as _engine
        _engine.PAIR_GEMMS = False
    if args.no_gn_epilogue:
        from storygen_amd import engine as _engine
        _engine.GN_EPILOGUE_STATS = False
    if args.attn_pair:
        from storygen_amd import engine as _engine
        _engine.ATTN_PAIR = True
    if args.fp16_block_stream:
        from storygen_amd import engine as _engine
        _engine.FP16_BLOCK_STREAM = True
    if args.no_splitk_in_gn:
        from storygen_amd import engine as _engine
        _engine.SPLITK_IN_GN = False
    if args.no_ff_fused:
        from storygen_amd import engine as _engine
        _engine.FF_FUSED = False
    if args.no_ff_split:
        from storygen_amd import engine as _engine
        _engine.FF_SPLIT_MAX_TOKENS = 0
    if args.no_ff_proj_merge:
        from storygen_amd import engine as _engine
        _engine.FF_PROJ_MERGE = False

    hw, n_ref = (96, 5) if args.config5_shape else (HW, R)
    # per-sample GFLOP of one ref / main pass: 64x64 R=3, or 96x96 R=5
    ref_gf, main_gf = (2148.1, 5594.2) if args.config5_shape else (REF_GF, MAIN_GF)
    step_tflop = 3 * (n_ref * ref_gf + main_gf) / 1000.0
    arch = build_arch(SD15_CONFIG)
    sd = cached_state_dict(arch, 0, rank, use_dist)
    inputs = synthetic_inputs(N_PER_GPU, n_ref, hw, hw, seed=rank, cross_attention_dim=arch.config["cross_attention_dim"])
    G = default_ref_ahead(args)                     # default: the largest group size <= DEFAULT_REF_AHEAD that divides the timed window
    if G > 1 and args.steps % G:
        raise SystemExit(f"--ref-ahead {G} must divide --steps {args.steps}")
    warmup_run = -(-args.warmup // G) * G          # the timed window starts on a group boundary
    sampler = StoryGenSampler(arch, sd, dev, N_PER_GPU, hw, hw, n_ref, use_graph=not args.no_graph, dedup=not args.no_dedup,
                              overlap=not args.no_overlap, ref_ahead=G, split_graphs=args.split_graphs,
                              stream_priority=args.stream_priority, fp8_attention=args.fp8_attention,
                              short_rows=not args.no_short_rows, time_tables=not args.no_time_tables,
                              shared_head=not args.no_shared_head, ref_cus=args.ref_cus, ref_cu_layout=args.ref_cu_layout,
                              ref_eager=args.ref_eager, ref_fp16_stream=None if args.ref_fp32_stream else (True, True))
    # the schedule holds whole groups: 50 steps (the reference's DDIM table) for G in {1, 2, 5}; any other G (a --steps it must divide)
    # rounds the table up to the next multiple
    n_sched = -(-max(T, args.steps + warmup_run) // G) * G
    sampler.prepare(inputs, n_sched, args.stage, 7.5, 3.5)

    dt = timed_steps(sampler, args.steps, warmup_run, use_dist, dev)
    final, gather_ms, finite = gather_and_check(sampler, world, dev)
    distinct = True                                             # asserted by gather_and_check
    sampler.check_guards()                                      # the LayerNorm fold stayed inside its range (raises otherwise)
    # one whole image, prepare() to last step (not the contract's `value`: the steady-state window above is)
    loop_ms = loop_prep = None
    n_loop = -(-T // G) * G
    if not args.no_loop and not args.no_graph:
        loop_ms, loop_prep = timed_loop(sampler, inputs, n_loop, args.stage, use_dist, dev)
        sampler.check_guards()

    if rank == 0:
        value = world * N_PER_GPU * args.steps / dt
        out = {
            "metric": ("UNet denoising steps/sec @768x768, 5 prior-frame ctx, bs=1 (non-contract)" if (args.config5_shape or args.fp8_attention) else
                       "UNet denoising steps/sec @512x512, 3 prior-frame ctx, bs=1"
                       + (" (non-contract: stage auto-regressive, the reference's inference.py default)" if args.stage != "multi-image-condition" else "")),
            "value": round(value, 4),
            "unit": "denoising steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "ms_per_step_by_rank": [round(1e3 * t / args.steps, 3) for t in timed_steps.by_rank],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 (attention operands e4m3)" if args.fp8_attention else "f16", "data": "synthetic",
            "config": {"workload": ("NON-CONTRACT RUN, BASELINE configs[4]: 768x768 (96x96x4 latent), R=5 prior frames, "
                                    + ("fp8 (e4m3) MFMA attention for the head-dim-40 image / self attention" if args.fp8_attention
                                       else "fp16 attention") if args.config5_shape else
                                    "BASELINE configs[1]: StoryGen denoising loop, 512x512 (64x64x4 latent), R=3 prior "
                                    "frames, CFG batch 3, DDIM, SD-1.5 UNet + attn3 (909M params, synthetic fp16 weights)"),
                       "stage": args.stage, "samples_per_gpu": N_PER_GPU, "parallelism": f"dp{world} (one sample per GPU, final all-gather)",
                       "hipgraph": not args.no_graph, "dedup_identical_reference_samples": not args.no_dedup,
                       "overlap_ref_pass_of_next_step": sampler.overlap, "ref_ahead": G, "warmup_run": warmup_run,
                       "split_graphs": sampler.split, "stream_priority": sampler.stream_priority, "ref_pass_on_cus": args.ref_cus, "ref_pass_eager": args.ref_eager,
                       "paired_gemm_launches": not args.no_gemm_pairs, "paired_text_image_attention": args.attn_pair,
                       "groupnorm_stats_from_epilogues": not args.no_gn_epilogue, "fp16_block_stream": args.fp16_block_stream,
                       "short_zero_image_rows": not args.no_short_rows, "time_embedding_tables": not args.no_time_tables, "splitk_reduce_in_groupnorm": not args.no_splitk_in_gn,
                       "fused_feed_forward_64x64": not args.no_ff_fused, "fused_feed_forward_hidden_split": not (args.no_ff_fused or args.no_ff_split), "ff2_proj_out_one_gemm": not args.no_ff_proj_merge, "reference_engine_fp16_stream": not args.ref_fp32_stream, "shared_cfg_head_of_main_pass": bool(sampler.main.cfg_shared_head)},
            "tflop_per_step_as_written": round(step_tflop, 3),
            "final_allgather_ms": round(gather_ms, 3), "latents_gathered": int(final.shape[0]), "latents_finite": finite,
            "latents_distinct_per_rank": distinct,
        }
        if loop_ms is not None:
            best = min(loop_ms)
            out["loop_50_steps_ms"] = round(best * T / n_loop, 2)
            out["loop"] = {"steps": n_loop, "ms_each_repeat": [round(v, 2) for v in loop_ms], "prepare_host_ms": [round(v, 2) for v in loop_prep],
                           "includes": "prepare() (text K/V, time tables, un-overlapped primer reference pass) + run(); last group without look-ahead pass",
                           "ms_per_step_over_the_loop": round(best / n_loop, 3),
                           "overhead_ms_per_step_vs_steady_state": round(best / n_loop - 1e3 * dt / args.steps, 3)}
        out["roofline"], executed = in_situ_roofline(sampler, args.dump_algorithmic)
        if args.config5_shape:
            out["roofline"]["traffic"] = None          # profiles/traffic.json was measured on the contract workload
        out["tflop_per_step_executed"] = round(executed, 3)
        out["mfma_frac_whole_step"] = round(value * executed / (world * PEAK_FP16_TFLOPS), 4)
        out["sample_forwards_per_step"] = {"reference": sampler.executed_sample_forwards()[0],
                                           "main": sampler.executed_sample_forwards()[1], "as_written": 3 * n_ref + 3}
        if args.fp8_attention:
            # the e4m3 path is the throughput OPTION BASELINE configs[4] names, not a parity path: state what it costs in accuracy next to
            # what it buys (since round 5: nothing — the fp16 D = 40 kernel got the softmax-in-MFMA work, the e4m3 kernel did not)
            n_cmp = 2 * G if G > 1 else 10
            sampler.prepare(inputs, n_sched, args.stage, 7.5, 3.5)
            a = sampler.run(max_steps=n_cmp).clone()
            ref16 = StoryGenSampler(arch, None, dev, N_PER_GPU, hw, hw, n_ref, weights=sampler.weights, ref_ahead=G)
            ref16.prepare(inputs, n_sched, args.stage, 7.5, 3.5)
            b = ref16.run(max_steps=n_cmp).clone()
            torch.cuda.synchronize()
            out["fp8_attention"] = {"latents_rel_l2_vs_fp16_path": round(float((a.double() - b.double()).norm() / b.double().norm()), 6),
                                    "after_steps": n_cmp,
                                    "status": "measured negative: slower than the fp16 attention on the same box since round 5 (compare the "
                                              "--config5-shape line) and outside the 1e-3 parity bar (3.9e-3 .. 7.6e-3 vs the oracle golden, "
                                              "tests/test_unet_gpu.py); kept as an opt-in, the product default for config 5 is fp16"}
            del ref16
        if world == 1 and not args.no_cpu_baseline and not args.config5_shape:
            out["cpu_baseline"] = cpu_baseline(arch, sd, inputs)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
