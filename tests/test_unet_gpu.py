"""End-to-end parity of the HIP UNet / denoising loop against the oracle (CPU fp32 restatement of the reference)
and against the golden vectors produced by the reference itself (tests/golden/*.pt, oracle/make_golden.py).

Tolerances (fp16 storage, fp32 accumulate vs an fp32 oracle): rel-L2 on epsilon / features / latents.  The
north-star bar is 1e-3 relative on the latents; per-tensor bars for intermediate features are looser because a
single fp16 rounding of an activation is already 2^-11 = 4.9e-4 relative."""
import os

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_EPS = 6e-3        # one UNet pass, epsilon (fp16 operands: ~100 roundings of 2^-11 down the whole pass)
TOL_FEAT = 2e-3       # ... each of the 16 harvested features against the reference's own (round 6: a 10x regression of ONE block shows here)
TOL_LATENT = 1e-3     # latents after a denoising step (north-star)


@pytest.fixture(scope="module")
def sd15(gpu):
    from storygen_amd.arch import SD15_CONFIG, build_arch
    from storygen_amd.synth import synthetic_state_dict
    arch = build_arch(SD15_CONFIG)
    return arch, synthetic_state_dict(arch, 0)


def _ref_and_main_inputs(inputs, sched, t_main):
    ref_t = t_main // 10
    an = sched.add_noise
    x = torch.cat([an(inputs["zero_prompt"], inputs["noise"], ref_t), an(inputs["image_prompts"][0], inputs["noise"], ref_t),
                   an(inputs["image_prompts"][0], inputs["noise"], ref_t)])
    e = torch.cat([inputs["prev_uncond"][0], inputs["prev_text"][0], inputs["prev_text"][0]])
    xm = torch.cat([inputs["latents"]] * 3)
    em = torch.cat([inputs["uncond"], inputs["uncond"], inputs["text"]])
    return ref_t, x, e, xm, em


def test_unet_passes_vs_oracle_32x32(gpu, sd15):
    """SD-1.5 architecture at a 32x32 latent (2 prior frames): harvest pass features + eps, then the main pass that
    consumes them, HIP vs the oracle run live on the host (the reference itself cannot run this size)."""
    from oracle import storygen_oracle as O
    from storygen_amd.engine import UNetEngine
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    cfg, hw, R = arch.config, 32, 2
    inputs = synthetic_inputs(1, R, hw, hw, 1, cfg["cross_attention_dim"])
    sched = O.DDIM()
    t_main = sched.timesteps(5)[0]
    ref_t, x, e, xm, em = _ref_and_main_inputs(inputs, sched, t_main)
    with torch.no_grad():
        o_eps, o_feats = O.unet_forward(sd, cfg, x, ref_t, e, None)
        ctx = {k: torch.cat([v, 0.5 * v], dim=1) for k, v in o_feats.items()}      # two distinct "frames"
        o_main, _ = O.unet_forward(sd, cfg, xm, t_main, em, ctx)
    eng = UNetEngine(arch, sd, gpu, 3, hw, hw, R)
    eng.set_inputs(x, ref_t, e)
    eps = eng.forward(harvest_slot=0).clone()
    torch.cuda.synchronize()
    errs = {"eps(ref)": rel_l2(eps.cpu(), o_eps)}
    for k, v in eng.features(0).items():
        errs[k] = rel_l2(v.float().cpu(), o_feats[k])
    # main pass on the ORACLE's features so that the two passes are judged independently
    for k, v in ctx.items():
        eng.ctx[k].copy_(v.to(gpu, torch.float16))
    eng.set_inputs(xm, t_main, em)
    eps_m = eng.forward(consume=True)
    torch.cuda.synchronize()
    errs["eps(main)"] = rel_l2(eps_m.cpu(), o_main)
    print({k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= TOL_EPS, errs


def test_ff2_and_proj_out_as_one_gemm_is_the_same_pass_32x32(gpu, sd15):
    """engine.FF_PROJ_MERGE (round 6): ff.net.2 and proj_out of a transformer block (model/attention.py:300,121-123) contracted in ONE
    K = 5C GEMM over [GEGLU output | raw copy of h3] with the host-made product [W_out W_2 | W_out] — against the two GEMMs as
    written on the same inputs (both within the one-pass bar of the oracle, and within it of each other), for a reference-style pass
    and a main pass consuming context.  The C = 320 level runs the fused feed-forward kernel either way."""
    from oracle import storygen_oracle as O
    from storygen_amd import engine as E
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    cfg, hw, R = arch.config, 32, 2
    inputs = synthetic_inputs(1, R, hw, hw, 3, cfg["cross_attention_dim"])
    sched = O.DDIM()
    t_main = sched.timesteps(5)[0]
    ref_t, x, e, xm, em = _ref_and_main_inputs(inputs, sched, t_main)
    with torch.no_grad():
        o_eps, o_feats = O.unet_forward(sd, cfg, x, ref_t, e, None)
        ctx = {k: torch.cat([v, 0.5 * v], dim=1) for k, v in o_feats.items()}
        o_main, _ = O.unet_forward(sd, cfg, xm, t_main, em, ctx)
    got = {}
    try:
        for merge in (True, False):
            E.FF_PROJ_MERGE = merge
            eng = E.UNetEngine(arch, sd, gpu, 3, hw, hw, R)
            assert any(xf.w_ffo is not None for xf in eng.xfs.values())
            eng.set_inputs(x, ref_t, e)
            eps_r = eng.forward(harvest_slot=0).clone().cpu()
            for k, v in ctx.items():
                eng.ctx[k].copy_(v.to(gpu, torch.float16))
            eng.set_inputs(xm, t_main, em)
            eps_m = eng.forward(consume=True).clone().cpu()
            assert eng.check_ln_guard() == 0
            got[merge] = (eps_r, eps_m)
            del eng
    finally:
        E.FF_PROJ_MERGE = True
    errs = {f"{'merged' if m else 'as written'} {n}": rel_l2(t, o) for m in (True, False) for n, t, o in (("ref", got[m][0], o_eps), ("main", got[m][1], o_main))}
    errs["merged vs as written (ref)"] = rel_l2(got[True][0], got[False][0])
    errs["merged vs as written (main)"] = rel_l2(got[True][1], got[False][1])
    print({k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= TOL_EPS, errs


def test_layernorm_fold_guard_through_a_transformer_block(gpu):
    """The LayerNorm fold's guard in situ: a transformer block whose stream sits 30 sigma off zero (proj_in bias)
    is outside the fold's range — UNetEngine.check_ln_guard raises instead of returning an epsilon whose LayerNorm inputs were rounded
    at 30 x 2^-11 — and the same pass with engine.LN_FOLD = False (LayerNorm launches on the fp32 stream, as the reference computes
    them, model/attention.py:250,268,283,298) raises nothing and stays within the one-pass bar of the oracle.  |x| >= 65504 in the
    stream: the fp16 copy saturates and the guard names the range (the pass itself is lost either way: every fp16 operand derived
    from such a stream overflows)."""
    import __graft_entry__ as ge
    from oracle import storygen_oracle as O
    from storygen_amd import engine as E
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_inputs, synthetic_state_dict
    cfg = load_config(ge.SMOKE_CONFIG)
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 5)
    inp = synthetic_inputs(1, 1, 16, 16, 5, cfg["cross_attention_dim"])
    x, t, e = inp["latents"], 500, inp["text"]
    eng = E.UNetEngine(arch, sd, gpu, 1, 16, 16, 0)
    eng.set_inputs(x, t, e)
    eng.forward()
    assert eng.check_ln_guard() == 0                                  # synthetic weights: inside the range
    key = "up_blocks.1.attentions.2.proj_in.bias"                     # the last transformer of level 0: its h0 is still in the buffer
    sigma = float(eng.lv[0]["h0"].std(dim=-1).mean())
    for off, flag in ((30.0 * sigma, "sigma"), (1.0e5, "65504")):
        sd2 = dict(sd)
        sd2[key] = sd[key].float() + off
        errs = {}
        for fold in ((True, False) if flag == "sigma" else (True,)):
            E.LN_FOLD = fold
            try:
                eng2 = E.UNetEngine(arch, sd2, gpu, 1, 16, 16, 0)
                eng2.set_inputs(x, t, e)
                eps = eng2.forward().float().cpu()
                if fold:
                    with pytest.raises(FloatingPointError, match=flag):
                        eng2.check_ln_guard()
                    assert eng2.check_ln_guard() == 0                 # cleared by the raising call
                else:
                    assert eng2.check_ln_guard() == 0
            finally:
                E.LN_FOLD = True
            if flag == "sigma":
                if not errs:
                    with torch.no_grad():
                        want, _ = O.unet_forward(sd2, cfg, x, t, e, None)
                assert torch.isfinite(eps).all()
                errs[fold] = rel_l2(eps, want)
        if flag == "sigma":
            print(f"stream offset {off:.3g} (sigma {sigma:.3g}): eps rel-L2 vs oracle, folded {errs[True]:.2e} / LayerNorm launches {errs[False]:.2e}")
            assert errs[False] <= TOL_EPS, errs
        # (a stream at 1e5 overflows every fp16 tensor derived from it — h4, q, k — with or without the fold: the engine's fp16 operands
        # assume |stream| < 65504 throughout, and the guard is where that assumption is checked)


def _probe(t, summary):
    return t.float().cpu().flatten()[summary["idx"]]


@pytest.mark.parametrize("case", ["sd15_64_r1", "sd15_64_r3"])
def test_denoise_steps_vs_reference_golden_64x64(gpu, sd15, case):
    """BASELINE configs 1 and 2 (first steps): 512x512, R = 1 / 3 prior frames, against latents produced by the
    reference's own pipeline loop (tests/golden, oracle/make_golden.py)."""
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    path = os.path.join(GOLDEN, f"{case}.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    gold = torch.load(path, weights_only=False)
    arch, sd = sd15
    R, hw = gold["n_ref"], gold["hw"]
    inputs = synthetic_inputs(1, R, hw, hw, gold["seed"], arch.config["cross_attention_dim"])
    smp = StoryGenSampler(arch, sd, gpu, 1, hw, hw, R, use_graph=True)
    stage = "multi-image-condition"
    smp.prepare(inputs, gold["n_steps"], stage, *gold["guidance"])
    want = gold["stages"][stage]["latents"]
    trace = []
    smp.run(max_steps=len(want), trace=trace)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(trace, want)]
    # features of the last reference pass / eps of the main pass of step 0 are also pinned by the golden file
    print(case, "latent rel-L2 per step:", [f"{e:.2e}" for e in errs])
    assert max(errs) <= TOL_LATENT, errs


@pytest.mark.parametrize("G", [1, 5])
@pytest.mark.parametrize("case,stage", [("sd15_64_r3_full", "multi-image-condition"), ("sd15_64_r3_ar_full", "auto-regressive")])
def test_full_depth_50_steps_vs_reference_golden_64x64(gpu, sd15, case, stage, G):
    """BASELINE config 2 at FULL depth — the north-star's bar is on the final latents: all 50 DDIM steps of the DEFAULT
    schedule (one hipGraph per step, dedup of identical reference samples, reference pass of step k+1 overlapped with the
    main pass of step k) against the latents the reference's own pipeline loop produced after every step
    (oracle/make_golden.py `sd15_64_r3_full` / `sd15_64_r3_ar_full`: 512x512, R=3, guidance 7.5 / 3.5).
    Bar: rel-L2 <= 1e-3 after EVERY step, in particular at steps 9 / 24 / 49.
    G = 5: the same with the group schedule bench.py runs (ref_ahead = 5: ten hipGraph replays, each the batched reference pass of the
    next five steps forked beside five main passes) — same bar, every step (sampler.lat_trace)."""
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    path = os.path.join(GOLDEN, f"{case}.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    gold = torch.load(path, weights_only=False)
    arch, sd = sd15
    R, hw = gold["n_ref"], gold["hw"]
    inputs = synthetic_inputs(1, R, hw, hw, gold["seed"], arch.config["cross_attention_dim"])
    smp = StoryGenSampler(arch, sd, gpu, 1, hw, hw, R, ref_ahead=G)           # every other default: graph + dedup + overlap
    assert smp.use_graph and smp.dedup and smp.overlap and not smp.split and smp.group == (G > 1)
    smp.prepare(inputs, gold["n_steps"], stage, *gold["guidance"])
    assert len(smp.graphs) == 2 and (G == 1 or smp.group_direct)
    want = gold["stages"][stage]["latents"]
    assert len(want) == gold["n_steps"] == 50
    trace = []
    smp.run(trace=trace)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(trace, want)]
    print(case, f"ref_ahead={G}", "latent rel-L2 at steps 0/9/24/49:", [f"{errs[i]:.2e}" for i in (0, 9, 24, 49)], "max", f"{max(errs):.2e}")
    assert len(errs) == 50 and max(errs) <= TOL_LATENT, errs


def test_stage_no_vs_oracle_32x32(gpu, sd15):
    """stage 'no' (pipeline.py:436-438,444-445; the oracle's restatement of it is pinned to the reference's own run by
    tests/golden/tiny_no.pt): no reference pass, main pass without image context, plain text CFG."""
    from oracle import storygen_oracle as O
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 5, arch.config["cross_attention_dim"])
    want = []
    O.sample_loop(sd, arch.config, inputs, 50, "no", 7.5, 3.5, max_steps=3, trace=want)
    smp = StoryGenSampler(arch, sd, gpu, 1, 32, 32, 2)
    smp.prepare(inputs, 50, "no", 7.5, 3.5)
    got = []
    smp.run(max_steps=3, trace=got)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(got, want)]
    assert max(errs) <= TOL_LATENT, errs
    # and back to a context stage on the same sampler object (the layout is rebuilt)
    smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
    smp.run(max_steps=1)
    torch.cuda.synchronize()
    assert torch.isfinite(smp.latents).all()


@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
def test_pndm_loop_vs_oracle_32x32(gpu, sd15, stage):
    """PNDM / PLMS (skip_prk_steps) in the loop — the scheduler class ckpt/stable-diffusion-v1-5/scheduler/scheduler_config.json
    names and model/pipeline.py:7-16 accepts: 5 inference steps = 6 UNet evaluations (the second timestep is visited twice),
    which exercises every branch of the multistep rule (1, 2, 3 and 4 history terms); against the oracle's stateful
    restatement of diffusers' PNDMScheduler (parity unpinned: diffusers is absent, see oracle.storygen_oracle.PNDM)."""
    from oracle import storygen_oracle as O
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.scheduler import PNDMSchedule
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 9, arch.config["cross_attention_dim"])
    # all 6 evaluations (every branch of the multistep rule) in the contract stage; the other stage differs only in the reference
    # passes' noise levels, so its first 3 evaluations (incl. the repeated timestep) are enough — the oracle runs live on the CPU
    n_eval = 6 if stage == "multi-image-condition" else 3
    want = []
    O.sample_loop(sd, arch.config, inputs, 5, stage, 7.5, 3.5, max_steps=n_eval, trace=want, scheduler="pndm")
    smp = StoryGenSampler(arch, sd, gpu, 1, 32, 32, 2, schedule=PNDMSchedule(skip_prk_steps=True))
    smp.prepare(inputs, 5, stage, 7.5, 3.5)
    assert smp.timesteps == [801, 601, 601, 401, 201, 1] and smp.num_steps == 6
    got = []
    smp.run(max_steps=n_eval, trace=got)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(got, want)]
    print(stage, [f"{e:.2e}" for e in errs])
    # a 5-step schedule multiplies the per-pass epsilon error by far larger coefficients than the 50-step one the 1e-3 bar
    # is stated for: 3e-3 here
    assert len(errs) == n_eval and max(errs) <= 3e-3, errs


def test_pndm_on_the_50_step_schedule_meets_the_north_star_bar_32x32(gpu, sd15):
    """The same PNDM / PLMS loop on the schedule the 1e-3 bar is stated for (50 inference steps): the first four UNet evaluations
    (1, 2, 3 history terms and the repeated second timestep) against the oracle's restatement, bar 1e-3 — the 3e-3 of the
    5-step test above is the schedule's coefficients, not the update rule."""
    from oracle import storygen_oracle as O
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.scheduler import PNDMSchedule
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 9, arch.config["cross_attention_dim"])
    want = []
    O.sample_loop(sd, arch.config, inputs, 50, "multi-image-condition", 7.5, 3.5, max_steps=4, trace=want, scheduler="pndm")
    smp = StoryGenSampler(arch, sd, gpu, 1, 32, 32, 2, schedule=PNDMSchedule(skip_prk_steps=True))
    smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
    got = []
    smp.run(max_steps=4, trace=got)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(got, want)]
    print("PNDM, 50-step schedule, evaluations 1..4:", [f"{e:.2e}" for e in errs])
    assert len(errs) == 4 and max(errs) <= TOL_LATENT, errs      # (the 4-history-term branch: the 5-step test above)


def test_unet_single_pass_vs_reference_golden_64x64(gpu, sd15):
    """One harvest pass + one main pass at 64x64 against probes of the reference UNet's own outputs."""
    from oracle import storygen_oracle as O
    from storygen_amd.engine import UNetEngine
    from storygen_amd.synth import synthetic_inputs
    path = os.path.join(GOLDEN, "sd15_64_r1.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    gold = torch.load(path, weights_only=False)
    arch, sd = sd15
    R, hw = gold["n_ref"], gold["hw"]
    inputs = synthetic_inputs(1, R, hw, hw, gold["seed"], arch.config["cross_attention_dim"])
    sched = O.DDIM()
    u = gold["unet"]
    ref_t, x, e, xm, em = _ref_and_main_inputs(inputs, sched, u["t_main"])
    assert ref_t == u["t_ref"]
    eng = UNetEngine(arch, sd, gpu, 3, hw, hw, R)
    eng.set_inputs(x, ref_t, e)
    eps = eng.forward(harvest_slot=0).clone()
    errs = {"eps(ref)": rel_l2(eps.cpu(), u["ref_sample"]["full"])}
    for k, v in eng.features(0).items():
        errs[k] = rel_l2(_probe(v, u["feats"][k]), u["feats"][k]["values"])
    eng.set_inputs(xm, u["t_main"], em)
    eps_m = eng.forward(consume=True)          # R = 1: the context is exactly the harvested slot
    errs["eps(main)"] = rel_l2(eps_m.cpu(), u["main_sample"]["full"])
    print({k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) <= TOL_EPS, errs
    feats = {k: v for k, v in errs.items() if not k.startswith("eps")}
    assert len(feats) == 16 and max(feats.values()) <= TOL_FEAT, feats      # per-block bar: every harvested feature on its own


def test_graph_replay_matches_eager(gpu, sd15):
    """The captured hipGraph step and the eager step are the same kernels: bit-identical latents."""
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 5, arch.config["cross_attention_dim"])
    outs = []
    for use_graph in (False, True):
        smp = StoryGenSampler(arch, sd, gpu, 1, 32, 32, 2, use_graph=use_graph)
        smp.prepare(inputs, 4, "auto-regressive", 7.5, 3.5)
        outs.append(smp.run().clone())
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


def test_paired_text_image_attention_is_the_same_trajectory(gpu, sd15, monkeypatch):
    """engine.ATTN_PAIR (text + image cross-attention of every block in one sg_attn_fwd_pair_f16 launch, off by default — DESIGN
    §5.2b) runs the same arithmetic as the two launches: bit-identical latents after 4 steps."""
    from storygen_amd import engine
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 3, 32, 32, 5, arch.config["cross_attention_dim"])
    outs = []
    for paired in (False, True):
        monkeypatch.setattr(engine, "ATTN_PAIR", paired)
        smp = StoryGenSampler(arch, sd, gpu, 1, 32, 32, 3)
        smp.prepare(inputs, 4, "multi-image-condition", 7.5, 3.5)
        outs.append(smp.run().clone())
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])


def test_loop_vs_oracle_both_stages_32x32(gpu, sd15):
    """The whole loop (R=2, first 2 steps of the 50-step schedule BASELINE config 2 uses — the 1e-3 latent bar is
    stated for that schedule: a coarser one multiplies the same epsilon error by a larger DDIM coefficient) in both
    stages against the oracle loop at 32x32."""
    from oracle import storygen_oracle as O
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 7, arch.config["cross_attention_dim"])
    smp = StoryGenSampler(arch, sd, gpu, 1, 32, 32, 2, use_graph=True)
    for stage in ("multi-image-condition", "auto-regressive"):
        want = []
        O.sample_loop(sd, arch.config, inputs, 50, stage, 7.5, 3.5, max_steps=2, trace=want)      # (3 until round 3: suite time)
        smp.prepare(inputs, 50, stage, 7.5, 3.5)
        got = []
        smp.run(max_steps=2, trace=got)
        torch.cuda.synchronize()
        errs = [rel_l2(a.cpu(), b) for a, b in zip(got, want)]
        print(stage, [f"{e:.2e}" for e in errs])
        assert max(errs) <= TOL_LATENT, (stage, errs)


@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
def test_loop_two_samples_three_steps_vs_oracle_golden_32x32(gpu, sd15, stage):
    """N = 2 story frames, R = 2, THREE steps against the oracle loop's latents (tests/golden/sd15_32_n2_r2.pt, oracle/make_golden_n2.py;
    until round 6 the oracle ran live, ~4 CPU-minutes, behind SG_SLOW_TESTS — i.e. never under the driver), on the default schedule and
    on the group schedule (ref_ahead = 3): pins the row-major unit order of the reference batch, the per-unit noise expansion
    (noise[n(u)]) and the later steps of the trajectory for N > 1."""
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    path = os.path.join(GOLDEN, "sd15_32_n2_r2.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (oracle/make_golden_n2.py)")
    gold = torch.load(path, weights_only=False)
    arch, sd = sd15
    inputs = synthetic_inputs(gold["n_samples"], gold["n_ref"], gold["hw"], gold["hw"], gold["seed_inputs"], arch.config["cross_attention_dim"])
    want = gold["latents"][stage]
    for G in (1, 3):
        smp = StoryGenSampler(arch, sd, gpu, 2, 32, 32, 2, use_graph=True, ref_ahead=G)
        smp.prepare(inputs, gold["table_steps"], stage, *gold["guidance"])
        got = []
        smp.run(max_steps=3, trace=got)
        torch.cuda.synchronize()
        errs = [rel_l2(a.cpu(), b) for a, b in zip(got, want)]
        print(stage, f"N=2 ref_ahead={G}", [f"{e:.2e}" for e in errs])
        assert len(errs) == 3 and max(errs) <= TOL_LATENT, (stage, G, errs)


@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
def test_dedup_of_identical_reference_samples_is_equivalent(gpu, sd15, stage):
    """The sampler's deduplicated reference pass (each distinct sample once) against the as-written batch
    (all 3R samples, pipeline.py:429-430): same arithmetic on the same values — only tile / split-K plans differ with
    the batch size, which reorders fp32 sums and so decorrelates the fp16 roundings downstream — so both must sit
    within the latent bar of the oracle and of each other."""
    from oracle import storygen_oracle as O
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 9, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    outs = []
    for dedup in (True, False):
        smp = StoryGenSampler(arch, None, gpu, 1, 32, 32, 2, use_graph=False, dedup=dedup, weights=wts)
        smp.prepare(inputs, 50, stage, 7.5, 3.5)
        assert smp.U == ((3 if stage == "multi-image-condition" else 4) if dedup else 6)
        outs.append(smp.run(max_steps=2).clone().cpu())
    want = O.sample_loop(sd, arch.config, inputs, 50, stage, 7.5, 3.5, max_steps=2)
    errs = [rel_l2(o, want) for o in outs]
    print(stage, "dedup/as-written vs oracle:", [f"{e:.2e}" for e in errs], "dedup vs as-written:", f"{rel_l2(outs[0], outs[1]):.2e}")
    assert max(errs) <= TOL_LATENT and rel_l2(outs[0], outs[1]) <= TOL_LATENT


def test_tabulated_time_embedding_is_the_same_trajectory(gpu, sd15):
    """StoryGenSampler(time_tables=True, the default): the Timesteps -> TimestepEmbedding -> 22 x time_emb_proj chain
    (unet_2d_condition.py:392-398) is evaluated once per distinct timestep at prepare() and every UNet call looks its rows up — the same
    kernels on the same values, so the latents are bit-identical to recomputing the chain in every call.  A second prepare() with another
    schedule refills the table in place under the captured graph; a timestep that is not in the table gives NaN rows, not stale ones."""
    from storygen_amd import ops
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 23, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    outs = {}
    for tab in (False, True):
        smp = StoryGenSampler(arch, None, gpu, 1, 32, 32, 2, weights=wts, time_tables=tab)
        graphs = None
        # same stage twice with different step counts: the layout is unchanged, so the SAME captured graphs run on a table refilled in
        # place (build_time_table(fresh=False) under live graphs); then another stage (new layout: new engines, new capture)
        for steps, stage in ((50, "auto-regressive"), (20, "auto-regressive"), (25, "multi-image-condition")):
            smp.prepare(inputs, steps, stage, 7.5, 3.5)
            if steps == 50:
                graphs = list(smp.graphs)
            elif steps == 20:
                assert len(graphs) == 2 and all(a is b for a, b in zip(graphs, smp.graphs)), "the second prepare() must not re-capture"
            outs[tab, steps] = smp.run(max_steps=4).clone()
            torch.cuda.synchronize()
        assert (smp.main.time_table is not None) == tab
    for steps in (50, 20, 25):
        assert torch.isfinite(outs[True, steps]).all()
        assert torch.equal(outs[False, steps], outs[True, steps]), steps
    assert not torch.equal(outs[True, 50], outs[True, 20])                                      # different schedules: different latents
    # the lookup itself: hits copy the row, a miss is NaN
    keys, tab_rows = smp.main.time_table
    t = torch.tensor([float(keys[1]), 12345.0, float(keys[0])], device=gpu)
    out = torch.zeros(3, tab_rows.shape[1], device=gpu)
    ops.lookup_rows(t, keys, tab_rows, out)
    assert torch.equal(out[0], tab_rows[1]) and torch.equal(out[2], tab_rows[0]) and torch.isnan(out[1]).all()


def test_two_branch_graph_is_bit_reproducible_run_to_run(gpu, sd15):
    """Round 6: the same five steps through FRESH samplers (new buffers, new captured graphs) four times, default schedule — the main
    pass and the forked reference pass run concurrently inside one graph, so this is where a timing-dependent kernel shows: every
    repeat must reproduce the first bit for bit.  (With the columns-are-tokens LayerNorm fold on the 32x32-per-wave kernel this
    failed in 4 - 29 of 30 repeats at this size, profiles/r06h_*; the shipped plan keeps it on the 64x64-per-wave kernel.)  The
    32x32 latent matters: its 4x4 level has 48 tokens, i.e. ragged tiles on every GEMM."""
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 21, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    outs = []
    for _ in range(4):
        smp = StoryGenSampler(arch, None, gpu, 1, 32, 32, 2, use_graph=True, weights=wts)
        smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
        outs.append(smp.run(max_steps=5).clone())
        torch.cuda.synchronize()
        del smp
    assert all(torch.isfinite(o).all() for o in outs)
    assert all(torch.equal(outs[0], o) for o in outs[1:]), [float((o - outs[0]).abs().max()) for o in outs]


def test_split_graphs_with_stream_priority_is_the_same_trajectory(gpu, sd15):
    """split_graphs (+ stream_priority): the same kernels as the single-graph overlap schedule, launched as two graphs
    on two (prioritised) streams — bit-identical latents."""
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 21, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    outs = []
    for kw in (dict(), dict(split_graphs=True), dict(split_graphs=True, stream_priority=True)):
        smp = StoryGenSampler(arch, None, gpu, 1, 32, 32, 2, use_graph=True, weights=wts, **kw)
        smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
        outs.append(smp.run(max_steps=5).clone())
        torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("stage,G,split", [("multi-image-condition", 5, False), ("auto-regressive", 2, False), ("multi-image-condition", 4, True)])
def test_ref_ahead_batches_reference_passes_of_G_steps(gpu, sd15, stage, G, split):
    """ref_ahead = G: the reference samples of G consecutive steps run as one batched UNet call, one group ahead of the main passes
    that consume them — as ONE hipGraph per group (default: the pass forked beside the group's G main passes, features and their
    K / V^T written in place into the group's contiguous context sets) or as separately launched graphs (split_graphs).  Same per-sample
    arithmetic as the step-by-step schedule (only the batch-dependent tile / split-K plans differ), so every latent of a trajectory
    of two full groups and more must stay within twice the latent bar of the default schedule's."""
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 13, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    n = 2 * G + 1 if split else 3 * G
    traces = []
    for g in (1, G):
        smp = StoryGenSampler(arch, None, gpu, 1, 32, 32, 2, use_graph=True, weights=wts, ref_ahead=g, split_graphs=split and g > 1)
        smp.prepare(inputs, 50, stage, 7.5, 3.5)
        assert smp.U == g * smp.U0 and len(smp.ctx_sets) == 2 * g
        assert smp.group == (g > 1 and not split) and (not smp.group or (smp.group_direct and len(smp.graphs) == 2))
        tr = []
        smp.run(max_steps=n, trace=tr)
        torch.cuda.synchronize()
        assert len(tr) == n
        traces.append([t.cpu() for t in tr])
    errs = [rel_l2(a, b) for a, b in zip(traces[1], traces[0])]
    print(stage, f"ref_ahead={G} split={split} vs step-by-step, per step:", [f"{e:.1e}" for e in errs])   # measured: 3.7e-4 ... 7.8e-4 at step 9
    # two fp16 realisations of the same trajectory (different tile / split-K plans reorder the fp32 sums and so decorrelate
    # the fp16 roundings), each within TOL_LATENT of the fp32 truth — the default schedule's distance to the oracle is
    # asserted by the tests above; this one bounds the distance between the two schedules
    assert max(errs) <= 2 * TOL_LATENT, errs


@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
def test_shared_cfg_head_is_the_same_main_pass(gpu, sd15, stage):
    """cfg_shared_head (default for one story frame per GPU): the three CFG samples of the main pass are the same latent at the same
    timestep (pipeline.py:448-453), so conv_in, the first ResnetBlock2D and the first transformer up to its query projections run once,
    its text attention for (uncond, text) and its image attention for (zero-image, frames) only.  Same arithmetic on the same values
    as the batch-3 pass (only the batch-dependent tile / split-K plans of those few layers differ), so the two trajectories must stay
    within the distance two fp16 realisations of one trajectory have — and the shared pass must execute fewer FLOPs."""
    from storygen_amd import ops
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 29, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    traces, flops = [], []
    for head in (False, True):
        smp = StoryGenSampler(arch, None, gpu, 1, 32, 32, 2, use_graph=True, weights=wts, shared_head=head)
        smp.prepare(inputs, 50, stage, 7.5, 3.5)
        assert smp.main.cfg_shared_head == head
        tr = []
        smp.run(max_steps=3, trace=tr)
        torch.cuda.synchronize()
        traces.append([t.cpu() for t in tr])
        sink = []
        ops.PROFILE_SINK = sink
        try:
            smp._main_pass(0)
            torch.cuda.synchronize()
        finally:
            ops.PROFILE_SINK = None
        flops.append(sum(f for _, f, _, _, _ in sink))
        smp.check_guards()
    errs = [rel_l2(a, b) for a, b in zip(traces[1], traces[0])]
    print(stage, "shared head vs batch 3, per step:", [f"{e:.1e}" for e in errs], f"main-pass GFLOP {flops[0] / 1e9:.1f} -> {flops[1] / 1e9:.1f}")
    # two fp16 realisations of one trajectory (the tile / split-K plans of the shared layers differ with the batch), each within
    # TOL_LATENT of the fp32 truth (asserted against the goldens above, on the default = shared-head path): as for ref_ahead, this
    # bounds their distance from each other — measured 3.7e-4 ... 6.2e-4 over three steps on different boxes
    assert max(errs) <= 2 * TOL_LATENT, errs
    # the guarantee the engine relies on is checked on its first eager pass: three different latents are refused, not averaged away
    eng = smp.main
    eng._head_checked = False
    eng.x_in[1].add_(1.0)
    with pytest.raises(ValueError, match="same latent"):
        eng.forward(consume=True, text_cache=True)
    assert flops[1] < 0.98 * flops[0]          # (32x32, R = 2: 2.7 % of the main pass; 512x512, R = 3: the step executes 6.435 instead of 6.585 TFLOP)


def test_group_schedule_eager_equals_its_graph(gpu, sd15):
    """The group schedule without a graph (reference pass of the next group, then the G main passes, in order on one stream) runs the
    same kernels on the same buffers as its captured form: bit-identical latents after every step."""
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 17, arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    traces = []
    for graph in (True, False):
        smp = StoryGenSampler(arch, None, gpu, 1, 32, 32, 2, use_graph=graph, weights=wts, ref_ahead=2)
        smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
        tr = []
        smp.run(max_steps=4, trace=tr)
        torch.cuda.synchronize()
        traces.append(tr)
    assert all(torch.equal(a, b) for a, b in zip(*traces))


def test_distinct_prev_uncond_disables_zero_sharing(gpu, sd15):
    """The zero-image sample is shared across frames only when its inputs really are identical."""
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(1, 2, 32, 32, 9, arch.config["cross_attention_dim"])
    inputs["prev_uncond"] = inputs["prev_uncond"].clone()
    inputs["prev_uncond"][1] += 0.25
    smp = StoryGenSampler(arch, sd, gpu, 1, 32, 32, 2, use_graph=False)
    smp.prepare(inputs, 50, "multi-image-condition", 7.5, 3.5)
    assert smp.U == 4


@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
def test_two_samples_per_gpu_vs_oracle_32x32(gpu, sd15, stage):
    """N = 2 independent story frames in one sampler (num_images_per_prompt / batch of prompts): exercises the general-N
    layout of the deduplicated reference batch, the harvest scatter and the shared-context attention groups."""
    from oracle import storygen_oracle as O
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    arch, sd = sd15
    inputs = synthetic_inputs(2, 2, 32, 32, 13, arch.config["cross_attention_dim"])
    want = []
    O.sample_loop(sd, arch.config, inputs, 50, stage, 7.5, 3.5, max_steps=2, trace=want)
    smp = StoryGenSampler(arch, sd, gpu, 2, 32, 32, 2, use_graph=True)
    smp.prepare(inputs, 50, stage, 7.5, 3.5)
    assert smp.U == (2 * (1 + 2) if stage == "multi-image-condition" else 2 * 2 * 2)
    got = []
    smp.run(max_steps=2, trace=got)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(got, want)]
    per_sample = [rel_l2(got[-1][n].cpu(), want[-1][n]) for n in range(2)]
    print(stage, [f"{e:.2e}" for e in errs], "per sample", [f"{e:.2e}" for e in per_sample])
    assert max(errs + per_sample) <= TOL_LATENT


def _config5_golden():
    path = os.path.join(GOLDEN, "sd15_96_r5.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (oracle/make_golden_config5.py)")
    gold = torch.load(path, weights_only=False)
    assert gold["hw"] == 96 and gold["n_ref"] == 5 and len(gold["latents"]) >= 2
    return gold


@pytest.mark.parametrize("G", [1, 5])
def test_config5_fp16_path_vs_oracle_golden_96x96_r5(gpu, sd15, G):
    """BASELINE config 5's shape (768x768 = 96x96 latent, 5 prior frames; 46 080 context tokens at the first level) through the fp16
    path, against the latents of the oracle's loop at that shape (tests/golden/sd15_96_r5.pt: oracle.storygen_oracle with
    block-index feature keys — the reference's own height heuristic cannot run 96x96; the restatement is pinned to the
    reference at 64x64).  Default schedule (graph + dedup + overlap); bar: the north-star's 1e-3 after every stored step, and the
    as-written schedule (no dedup) agrees."""
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    gold = _config5_golden()
    arch, sd = sd15
    assert gold["seed_weights"] == 0
    inputs = synthetic_inputs(1, 5, 96, 96, gold["seed_inputs"], arch.config["cross_attention_dim"])
    wts = EngineWeights(arch, sd, gpu)
    n = len(gold["latents"]) // G * G
    smp = StoryGenSampler(arch, None, gpu, 1, 96, 96, 5, weights=wts, ref_ahead=G)
    smp.prepare(inputs, gold["n_steps"], gold["stage"], *gold["guidance"])
    trace = []
    smp.run(max_steps=n, trace=trace)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(trace, gold["latents"])]
    print(f"config 5 shape, fp16 path vs oracle, ref_ahead {G}, steps 1..{n}:", [f"{e:.2e}" for e in errs])
    assert n >= 5 and len(errs) == n and max(errs) <= TOL_LATENT, errs
    if G > 1:
        return
    first = trace[0].clone().cpu()
    del smp
    torch.cuda.empty_cache()
    smp = StoryGenSampler(arch, None, gpu, 1, 96, 96, 5, use_graph=False, dedup=False, weights=wts)
    smp.prepare(inputs, gold["n_steps"], gold["stage"], *gold["guidance"])
    aw = smp.run(max_steps=1).clone().cpu()
    assert rel_l2(aw, gold["latents"][0]) <= TOL_LATENT and rel_l2(aw, first) <= TOL_LATENT


@pytest.mark.parametrize("G", [1, 5])
def test_config5_fp8_attention_vs_oracle_golden_96x96_r5(gpu, sd15, G):
    """BASELINE config 5 as named: 768x768 (96x96 latent), 5 prior frames, the head-dim-40 image / self attention on the fp8
    (e4m3) MFMA path — deviation stated AGAINST THE ORACLE at steps 1, 2 and the last stored step (10 when the golden holds ten).
    e4m3 keeps 3 mantissa bits of Q, K, V and P: the attention outputs move by 3-6e-2 (test_attention_fp8_d40), the predicted noise
    of a pass by ~1.4e-2, and through the DDIM coefficients the latents by a few 1e-3 per step early in the schedule.  This is
    OUTSIDE the north-star's 1e-3 (which the fp16 path meets, test above): the fp8 path is the throughput option BASELINE config 5
    names, with the bound asserted here — 1.5e-2 over the first ten steps — and written in DESIGN.md 2."""
    from storygen_amd.engine import EngineWeights
    from storygen_amd.sampler import StoryGenSampler
    from storygen_amd.synth import synthetic_inputs
    gold = _config5_golden()
    arch, sd = sd15
    inputs = synthetic_inputs(1, 5, 96, 96, gold["seed_inputs"], arch.config["cross_attention_dim"])
    n = len(gold["latents"]) // G * G
    smp = StoryGenSampler(arch, sd, gpu, 1, 96, 96, 5, fp8_attention=True, ref_ahead=G)
    smp.prepare(inputs, gold["n_steps"], gold["stage"], *gold["guidance"])
    trace = []
    smp.run(max_steps=n, trace=trace)
    torch.cuda.synchronize()
    errs = [rel_l2(a.cpu(), b) for a, b in zip(trace, gold["latents"])]
    print(f"config 5, fp8 attention vs oracle, ref_ahead {G}, steps 1..{n}:", [f"{e:.2e}" for e in errs])
    assert all(torch.isfinite(t).all() for t in trace)
    assert max(errs) <= 1.5e-2, errs


def test_bench_multi_gpu_path_over_rccl_with_one_rank(gpu):
    """The N > 1 code path of bench.py (torch.distributed.run launch, RCCL init, barrier, max-over-ranks, the single all-gather
    of the final latents) exercised with the one GPU this box has: world_size 1 through the same launcher the driver uses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["latents_gathered"] == 1 and res["latents_finite"] and res["latents_distinct_per_rank"]
    assert res["config"]["parallelism"].startswith("dp1") and res["value"] > 0
