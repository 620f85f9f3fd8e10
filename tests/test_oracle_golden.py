"""Pins the oracle (oracle/storygen_oracle.py, the CPU fp32 restatement) against the golden vectors that were
produced by the REFERENCE ITSELF — /root/reference/model/{unet_2d_condition,unet_2d_blocks,attention,pipeline}.py
executed verbatim on the clean-room diffusers shim by oracle/make_golden.py (tests/golden/*.pt).  The reference
ships no tests or fixtures of its own for this path, so these files ARE the pinning.

CPU-only; sized for a few minutes on 8 cores: the full loop check runs on the `tiny` config (a 4-level StoryGen UNet
with 32..128 channels at the 64x64 latent the reference's block heuristic requires); the SD-1.5-sized
files are checked through their single-pass probes in the slower, opt-in test below.
"""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-5   # fp32 vs fp32, different summation orders (chunked attention, functional vs module graph)


def _load(case):
    path = os.path.join(GOLDEN, f"{case}.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} missing")
    return torch.load(path, weights_only=False)


def _setup(gold):
    from storygen_amd.arch import build_arch
    from storygen_amd.synth import synthetic_inputs, synthetic_state_dict
    arch = build_arch(gold["config"])
    sd = synthetic_state_dict(arch, gold["seed"])
    inputs = synthetic_inputs(1, gold["n_ref"], gold["hw"], gold["hw"], gold["seed"], arch.config["cross_attention_dim"])
    return arch, sd, inputs


def _unet_pass_inputs(O, gold, inputs):
    sched = O.DDIM()
    u = gold["unet"]
    an = sched.add_noise
    x = torch.cat([an(inputs["zero_prompt"], inputs["noise"], u["t_ref"]), an(inputs["image_prompts"][0], inputs["noise"], u["t_ref"]),
                   an(inputs["image_prompts"][0], inputs["noise"], u["t_ref"])])
    e = torch.cat([inputs["prev_uncond"][0], inputs["prev_text"][0], inputs["prev_text"][0]])
    xm = torch.cat([inputs["latents"]] * 3)
    em = torch.cat([inputs["uncond"], inputs["uncond"], inputs["text"]])
    return x, e, xm, em


def _summary_err(t, s):
    if "full" in s:
        return rel_l2(t, s["full"])
    return rel_l2(t.float().flatten()[s["idx"]], s["values"])


def test_golden_files_were_made_by_the_reference():
    for case in ("tiny", "sd15_64_r1", "sd15_64_r3"):
        g = _load(case)
        assert g["made_by"] == "oracle/make_golden.py"
        for st in g["stages"].values():
            assert st["restatement_rel_l2"] < TOL   # the self-check make_golden.py recorded in the build container
            assert len(st["latents"]) == g["exec_steps"]


def test_oracle_unet_passes_vs_reference_golden_tiny():
    """One harvest pass (eps + all 16 features) and the main pass that consumes them."""
    from oracle import storygen_oracle as O
    gold = _load("tiny")
    arch, sd, inputs = _setup(gold)
    u = gold["unet"]
    x, e, xm, em = _unet_pass_inputs(O, gold, inputs)
    with torch.no_grad():
        eps, feats = O.unet_forward(sd, arch.config, x, u["t_ref"], e, None)
        errs = {"eps(ref)": _summary_err(eps, u["ref_sample"])}
        assert list(feats) == list(u["feats"]) == arch.feature_keys
        for k, v in feats.items():
            errs[k] = _summary_err(v, u["feats"][k])
        ctx = {k: torch.cat([v] * gold["n_ref"], dim=1) for k, v in feats.items()}
        eps_m, empty = O.unet_forward(sd, arch.config, xm, u["t_main"], em, ctx)
    assert empty == {}
    errs["eps(main)"] = _summary_err(eps_m, u["main_sample"])   # make_golden.py: context = R copies of pass 0
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
def test_oracle_loop_vs_reference_golden_tiny(stage):
    """pipeline.py:411-469 as restated by the oracle vs the latents the reference's own pipeline produced."""
    from oracle import storygen_oracle as O
    gold = _load("tiny")
    arch, sd, inputs = _setup(gold)
    want = gold["stages"][stage]["latents"][:1]
    got = []
    O.sample_loop(sd, arch.config, inputs, gold["n_steps"], stage, *gold["guidance"], max_steps=len(want), trace=got)
    errs = [rel_l2(a, b) for a, b in zip(got, want)]
    assert max(errs) < TOL, errs


def test_oracle_loop_stage_no_vs_reference_golden_tiny():
    """stage 'no' (pipeline.py:436-438,444-445: no reference pass, main pass without image context) against the latents the
    reference's pipeline produced for it (tests/golden/tiny_no.pt)."""
    from oracle import storygen_oracle as O
    gold = _load("tiny_no")
    arch, sd, inputs = _setup(gold)
    want = gold["stages"]["no"]["latents"]
    got = []
    O.sample_loop(sd, arch.config, inputs, gold["n_steps"], "no", *gold["guidance"], max_steps=len(want), trace=got)
    assert max(rel_l2(a, b) for a, b in zip(got, want)) < TOL


@pytest.mark.skipif(os.environ.get("SG_SLOW_TESTS") != "1", reason="~1 min of CPU; set SG_SLOW_TESTS=1")
def test_oracle_unet_pass_vs_reference_golden_sd15():
    from oracle import storygen_oracle as O
    gold = _load("sd15_64_r1")
    arch, sd, inputs = _setup(gold)
    u = gold["unet"]
    x, e, _, _ = _unet_pass_inputs(O, gold, inputs)
    with torch.no_grad():
        eps, feats = O.unet_forward(sd, arch.config, x, u["t_ref"], e, None)
    errs = {"eps(ref)": _summary_err(eps, u["ref_sample"])}
    for k, v in feats.items():
        errs[k] = _summary_err(v, u["feats"][k])
    assert max(errs.values()) < TOL, errs


def test_oracle_leaf_semantics():
    """The leaf restatements against torch primitives / closed forms (the diffusers 0.13.1 semantics of SURVEY §8c)."""
    from oracle import storygen_oracle as O
    t = torch.tensor([0.0, 1.0, 981.0])
    emb = O.timestep_embedding(t, 320, True, 0)
    assert emb.shape == (3, 320)
    assert torch.allclose(emb[0, :160], torch.ones(160)) and torch.allclose(emb[0, 160:], torch.zeros(160))   # [cos|sin]
    f1 = torch.exp(-torch.log(torch.tensor(10000.0)) * 1 / 160)
    assert torch.allclose(emb[2, 1], torch.cos(981.0 * f1), atol=1e-5)
    assert torch.allclose(emb[2, 161], torch.sin(981.0 * f1), atol=1e-5)
    s = O.DDIM()
    assert s.timesteps(50)[:3] == [981, 961, 941] and s.timesteps(50)[-1] == 1     # steps_offset = 1
    assert s.timesteps(1) == [1]
    x, n = torch.randn(2, 4), torch.randn(2, 4)
    a = s.alphas_cumprod[500]
    assert torch.allclose(s.add_noise(x, n, 500), a.sqrt() * x + (1 - a).sqrt() * n)
    # eta=0 DDIM: stepping with the true noise from x_t = sqrt(a)x0 + sqrt(1-a)n lands on sqrt(a')x0 + sqrt(1-a')n
    xt = s.add_noise(x, n, 981)
    ap = s.alphas_cumprod[961]
    assert torch.allclose(s.step(n, 981, xt, 50), ap.sqrt() * x + (1 - ap).sqrt() * n, atol=1e-5)


@pytest.mark.parametrize("use_refs", [(2,), pytest.param((0, 1, 2), marks=pytest.mark.skipif(os.environ.get("SG_SLOW_TESTS") != "1",
                                                                                               reason="~40 s of CPU; set SG_SLOW_TESTS=1"))])
def test_oracle_train_step_vs_reference_golden(use_refs):
    """BASELINE config 4 (stage-2 training step): loss and the 80 attn3 gradient tensors of the oracle's restatement
    (oracle.storygen_oracle.train_step, train_StorySalon_stage2.py:291-327) against what the reference's own UNet + torch autograd
    produced (oracle/make_golden_train.py).  This pins the oracle for the backward path before any HIP backward kernel exists."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    gold = _load("tiny_train")
    arch = build_arch(gold["config"])
    sd = synthetic_state_dict(arch, gold["seed"])
    batch = synthetic_train_batch(gold["batch"], gold["hw"], arch.config["cross_attention_dim"], gold["seed"])
    g = gold["cases"]["refs_" + "".join(map(str, use_refs))]
    loss, grads = O.train_step(sd, arch.config, batch, use_refs)
    assert abs(float(loss) - g["loss"]) <= 1e-5 * abs(g["loss"])
    assert set(grads) == set(g["grads"]) and len(grads) == 5 * len(arch.feature_keys)
    errs = {}
    for k, e in g["grads"].items():
        assert tuple(grads[k].shape) == tuple(e["shape"])
        errs[k] = max(rel_l2(grads[k].flatten()[e["idx"]], e["values"]), abs(float(grads[k].double().norm()) - e["l2"]) / e["l2"])
    assert max(errs.values()) < 1e-4, max(errs.items(), key=lambda kv: kv[1])


def test_oracle_stage1_train_step_vs_reference_golden():
    """Stage 1 (train_StorySalon_stage1.py:171-179,262-291): modules named `*attn1` trainable, no reference pass, main pass with
    image_hidden_states=None — the oracle's loss and 80 attn1 gradients against the reference's own UNet + autograd
    (oracle/make_golden_train.py stage1)."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    gold = _load("tiny_train_stage1")
    assert gold["trainable"] == "attn1"
    arch = build_arch(gold["config"])
    sd = synthetic_state_dict(arch, gold["seed"])
    batch = synthetic_train_batch(gold["batch"], gold["hw"], arch.config["cross_attention_dim"], gold["seed"])
    g = gold["cases"]["refs_"]
    loss, grads = O.train_step(sd, arch.config, batch, (), trainable="attn1")
    assert abs(float(loss) - g["loss"]) <= 1e-5 * abs(g["loss"])
    assert set(grads) == set(g["grads"]) and all(".attn1." in k for k in grads)
    for k, e in g["grads"].items():
        assert rel_l2(grads[k].flatten()[e["idx"]], e["values"]) < 1e-4, k
        assert abs(float(grads[k].double().norm()) - e["l2"]) <= 1e-4 * e["l2"], k


def test_oracle_coco_train_step_vs_reference_golden():
    """train_COCO.py:286-316 (three frames at one noise level, unmasked loss): oracle vs the reference's own UNet + autograd."""
    if os.environ.get("SG_SLOW_TESTS") != "1":
        pytest.skip("~40 s of CPU; set SG_SLOW_TESTS=1 (oracle/make_golden_train.py coco ran the same comparison when it wrote the fixture)")
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    gold = _load("tiny_train_coco")
    arch = build_arch(gold["config"])
    sd = synthetic_state_dict(arch, gold["seed"])
    batch = synthetic_train_batch(gold["batch"], gold["hw"], arch.config["cross_attention_dim"], gold["seed"])
    batch["mask"] = torch.zeros_like(batch["mask"])
    g = gold["cases"]["refs_012"]
    loss, grads = O.train_step(sd, arch.config, batch, (0, 1, 2), ref_levels="coco")
    assert abs(float(loss) - g["loss"]) <= 1e-5 * abs(g["loss"])
    for k, e in g["grads"].items():
        assert rel_l2(grads[k].flatten()[e["idx"]], e["values"]) < 1e-4, k


def test_clip_text_oracle_matches_transformers_golden():
    """oracle/encoders_oracle.py::clip_text_forward against the outputs transformers' CLIPTextModel produced on the same (fp16-rounded)
    weights — the fixture oracle/make_golden_encoders.py wrote; that script also checks a 768-wide, 12-head configuration."""
    from oracle import encoders_oracle as eo
    gold = torch.load(os.path.join(GOLDEN, "clip_text_tiny.pt"), weights_only=True)
    sd = {k: v.float() for k, v in gold["state_dict"].items()}
    hidden, pooled = eo.clip_text_forward(sd, gold["input_ids"], heads=gold["heads"])
    assert rel_l2(hidden, gold["last_hidden_state"]) < 1e-5
    assert rel_l2(pooled, gold["pooled"]) < 1e-5
    # the restatement accepts transformers 4.x's `text_model.` prefix as well
    h2, _ = eo.clip_text_forward({"text_model." + k: v for k, v in sd.items()}, gold["input_ids"], heads=gold["heads"])
    assert torch.equal(h2, hidden)


def test_vae_oracle_structure():
    """The AutoencoderKL restatement is unpinned (diffusers is absent): check what can be checked — the reference's VAE config gives
    SD's 83 653 863 parameters under diffusers' names, shapes chain, the asymmetric stride-2 padding, mode() of the posterior."""
    import json
    from oracle import encoders_oracle as eo
    sd = eo.vae_random_state()
    assert sum(v.numel() for v in sd.values()) == 83_653_863
    cfg_path = "/root/reference/ckpt/stable-diffusion-v1-5/vae/config.json"
    if os.path.exists(cfg_path):
        cfg = json.load(open(cfg_path))
        assert tuple(cfg["block_out_channels"]) == (128, 256, 512, 512) and cfg["layers_per_block"] == 2 and cfg["latent_channels"] == 4
    small = eo.vae_random_state(block_out=(32, 64), layers_per_block=1, seed=1)
    x = torch.rand(2, 3, 16, 24)
    m = eo.vae_encode_moments(small, x)
    assert tuple(m.shape) == (2, 8, 8, 12)
    assert torch.equal(eo.gaussian_sample(m, None), m[:, :4])
    assert tuple(eo.vae_decode(small, m[:, :4]).shape) == (2, 3, 16, 24)
    # Downsample2D(padding=0) pads right/bottom only: shifting the image content by one pixel towards the top-left changes the result
    # differently from a symmetric padding — pin the asymmetric form on a delta image
    w = {"c.weight": torch.zeros(1, 1, 3, 3), "c.bias": torch.zeros(1)}
    w["c.weight"][0, 0, 0, 0] = 1.0                                  # picks input pixel (2*oy, 2*ox) under (0,1,0,1) padding
    d = torch.zeros(1, 1, 4, 4)
    d[0, 0, 2, 2] = 1.0
    y = eo._conv(torch.nn.functional.pad(d, (0, 1, 0, 1)), w, "c", stride=2, padding=0)
    assert y[0, 0, 1, 1] == 1.0 and y.sum() == 1.0


def test_vae_oracle_leaves_vs_torch_modules():
    """the AutoencoderKL restatement cannot be pinned end to end (diffusers is not installable here), so every
    LEAF is pinned instead — against the module classes the reference's own UNet runs on when the goldens are made
    (oracle/diffusers_shim ResnetBlock2D / Downsample2D / Upsample2D with the VAE's constructor arguments: temb_channels=None,
    eps 1e-6, padding 0) and against torch's own primitives (scaled_dot_product_attention for the one-head AttentionBlock, whose
    q / k pre-scaling by C^-1/4 each is the default 1/sqrt(C) scale).  Residual risk, stated: how diffusers 0.13.1 WIRES these
    leaves (block order, the mid block, quant / post_quant convs) is restated from its published source, not executed."""
    import sys
    from oracle import encoders_oracle as eo
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "diffusers_shim")
    sys.path.insert(0, shim)
    try:
        from diffusers.models.resnet import Downsample2D, ResnetBlock2D, Upsample2D
    finally:
        sys.path.remove(shim)
    torch.manual_seed(3)
    # ResnetBlock2D, with and without the 1x1 shortcut
    for cin, cout in ((32, 32), (32, 64)):
        m = ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=None, groups=8, eps=1e-6).eval()
        for prm in m.parameters():
            prm.data.normal_(0, 0.2)
        sd = {"r." + k: v.detach() for k, v in m.state_dict().items()}
        x = torch.randn(2, cin, 9, 7)
        assert torch.allclose(eo._resnet(x, sd, "r.", 8), m(x, None), atol=1e-5, rtol=1e-5)
    # Downsample2D(padding=0): right/bottom zero pad + stride-2 conv; Upsample2D: nearest x2 + conv
    d = Downsample2D(16, use_conv=True, out_channels=16, padding=0, name="op").eval()
    x = torch.randn(2, 16, 10, 6)
    sdd = {"c.weight": d.conv.weight.detach(), "c.bias": d.conv.bias.detach()}
    assert torch.allclose(eo._conv(F.pad(x, (0, 1, 0, 1)), sdd, "c", stride=2, padding=0), d(x), atol=1e-6)
    u = Upsample2D(16, use_conv=True, out_channels=16).eval()
    sdu = {"c.weight": u.conv.weight.detach(), "c.bias": u.conv.bias.detach()}
    assert torch.allclose(eo._conv(F.interpolate(x, scale_factor=2.0, mode="nearest"), sdu, "c"), u(x), atol=1e-6)
    # AttentionBlock: GroupNorm -> one-head attention over the H*W tokens -> proj + residual, on torch's own SDPA
    C, groups = 32, 8
    sda = {f"a.{n}.{w}": torch.randn(*sh) * 0.3 for n in ("query", "key", "value", "proj_attn")
           for w, sh in (("weight", (C, C)), ("bias", (C,)))}
    sda["a.group_norm.weight"], sda["a.group_norm.bias"] = torch.randn(C) * 0.2 + 1.0, torch.randn(C) * 0.2
    x = torch.randn(2, C, 6, 5)
    h = F.group_norm(x, groups, sda["a.group_norm.weight"], sda["a.group_norm.bias"], 1e-6).flatten(2).transpose(1, 2)
    q, k, v = (F.linear(h, sda[f"a.{n}.weight"], sda[f"a.{n}.bias"]) for n in ("query", "key", "value"))
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    want = F.linear(o, sda["a.proj_attn.weight"], sda["a.proj_attn.bias"]).transpose(1, 2).reshape(2, C, 6, 5) + x
    assert torch.allclose(eo._attention_block(x, sda, "a.", groups), want, atol=2e-5, rtol=1e-5)
    # posterior sample: mean + exp(0.5 clamp(logvar, -30, 20)) * eps  (DiagonalGaussianDistribution)
    mom = torch.randn(2, 8, 4, 4) * 20
    n = torch.randn(2, 4, 4, 4)
    assert torch.allclose(eo.gaussian_sample(mom, n), mom[:, :4] + torch.exp(0.5 * mom[:, 4:].clamp(-30, 20)) * n)
