"""Host-side composition of the training step (storygen_amd/train_blocks.py, storygen_amd/train.py) on the CPU: every
storygen_amd.ops entry point is replaced by a plain-torch stand-in written from the C-ABI contracts (tests/fake_ops.py),
so what is checked here is the op ORDER and WIRING — weight transposes for the dgrads, residual bookkeeping, the attn3
weight-gradient contractions, GEGLU interleaving, padded conv inputs, the skip-connection tape — against the oracle
(oracle/storygen_backward.py for the blocks, oracle.storygen_oracle.train_step for the whole step).  The kernels behind
those entry points are tested separately on hardware (tests/test_backward_gpu.py)."""
import pytest
import torch
import torch.nn.functional as F

import fake_ops
from conftest import rel_l2


@pytest.fixture
def cpu_ops(monkeypatch):
    fake_ops.install(monkeypatch)
    # the real wrappers refuse CPU tensors; the training code only calls through the patched module attributes
    return fake_ops


def _r(gen, *s, sc=1.0):
    return (torch.randn(*s, generator=gen) * sc).half().float()


def test_transformer_block_composition(cpu_ops):
    from oracle import storygen_backward as B
    from oracle import storygen_oracle as O
    from storygen_amd.train_blocks import TransformerBlockTrain
    from test_backward_gpu import _block_sd
    C, heads, Bn, N, S, Nc = 64, 2, 2, 24, 13, 40
    sd = _block_sd(C, 96, 3)
    g = torch.Generator().manual_seed(1)
    h, text, ctx, dout = _r(g, Bn, N, C), _r(g, Bn, S, 96), _r(g, Bn, Nc, C), _r(g, Bn, N, C)
    with torch.no_grad():
        want_out, _ = O.transformer_block(sd, "b", h, text, ctx, heads)
        want_dh, want_g = B.transformer_block_bwd(sd, "b", h, text, ctx, heads, dout)
    blk = TransformerBlockTrain(sd, "b", heads, "cpu")
    out = blk.forward(h.reshape(Bn * N, C).contiguous(), text.half().reshape(Bn * S, 96).contiguous(),
                      ctx.half().reshape(Bn * Nc, C).contiguous(), Bn)
    assert rel_l2(out.view(Bn, N, C), want_out) < 3e-3                     # fp16 operand storage between the ops
    dh, grads = blk.backward(dout.reshape(Bn * N, C).contiguous())
    assert rel_l2(dh.view(Bn, N, C), want_dh) < 1e-2
    assert set(grads) == {"to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight", "to_out.0.bias"}
    for k, v in grads.items():
        assert rel_l2(v, want_g[f"b.attn3.{k}"]) < 1e-2, k


@pytest.mark.parametrize("cin,cout", [(32, 32), (64, 32)])
def test_resnet_block_composition(cpu_ops, cin, cout):
    from oracle import storygen_backward as B
    from oracle import storygen_oracle as O
    from storygen_amd.train_blocks import ResnetBlockTrain
    Bn, H, W, temb, groups = 2, 6, 5, 48, 4
    g = torch.Generator().manual_seed(5)
    sd = {"r.norm1.weight": 1.0 + _r(g, cin, sc=0.1), "r.norm1.bias": _r(g, cin, sc=0.1), "r.norm2.weight": 1.0 + _r(g, cout, sc=0.1),
          "r.norm2.bias": _r(g, cout, sc=0.1), "r.conv1.weight": _r(g, cout, cin, 3, 3, sc=(9 * cin) ** -0.5),
          "r.conv1.bias": _r(g, cout, sc=0.1), "r.conv2.weight": _r(g, cout, cout, 3, 3, sc=(9 * cout) ** -0.5),
          "r.conv2.bias": _r(g, cout, sc=0.1), "r.time_emb_proj.weight": _r(g, cout, temb, sc=temb ** -0.5),
          "r.time_emb_proj.bias": _r(g, cout, sc=0.1)}
    if cin != cout:
        sd["r.conv_shortcut.weight"], sd["r.conv_shortcut.bias"] = _r(g, cout, cin, 1, 1, sc=cin ** -0.5), _r(g, cout, sc=0.1)
    x, emb, dout = _r(g, Bn, cin, H, W) + 0.3, _r(g, Bn, temb), _r(g, Bn, cout, H, W)
    with torch.no_grad():
        want = O.resnet_block(sd, "r", x, emb, groups, 1e-5)
        want_dx = B.resnet_block_bwd(sd, "r", x, emb, groups, 1e-5, dout)
        tproj = F.linear(F.silu(emb), sd["r.time_emb_proj.weight"], sd["r.time_emb_proj.bias"])
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(Bn * H * W, -1).contiguous()       # noqa: E731
    blk = ResnetBlockTrain(sd, "r", groups, 1e-5, "cpu")
    out = blk.forward(nhwc(x), tproj, Bn, H, W)
    assert rel_l2(out, nhwc(want)) < 3e-3
    assert rel_l2(blk.backward(nhwc(dout)), nhwc(want_dx)) < 1e-2


class _OracleRefEngine:
    """Stand-in for the inference UNetEngine of the reference passes: features from the oracle, written into the same
    [B, R*hw, C] fp16 context buffers."""

    def __init__(self, sd, cfg, arch, B, R, H, W):
        from storygen_amd.arch import feature_shapes
        self.sd, self.cfg = sd, cfg
        self.shapes = feature_shapes(arch, H, W)
        self.ctx = {k: torch.zeros(B, R * n, c, dtype=torch.float16) for k, (n, c) in self.shapes.items()}

    def set_inputs(self, x, t, text):
        self.inputs = (x.float(), t, text.float())

    def forward(self, harvest_slot):
        from oracle import storygen_oracle as O
        x, t, text = self.inputs
        with torch.no_grad():
            feats = O.unet_forward(self.sd, self.cfg, x, t, text, None)[1]
        for k, (n, _) in self.shapes.items():
            self.ctx[k][:, harvest_slot * n:(harvest_slot + 1) * n] = feats[k].half()


@pytest.mark.parametrize("use_refs", [(0, 1, 2), (1, 2)])
def test_training_step_composition_vs_oracle(cpu_ops, use_refs):
    """UNetTrainer.train_step (tape, skip bookkeeping, down / up sampler dgrads, conv_out dgrad as a conv_in, loss) on a
    2-level StoryGen UNet against oracle.storygen_oracle.train_step: loss and all attn3 gradients."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    cfg = load_config(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, attention_head_dim=2,
                           norm_num_groups=8, sample_size=64))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 7)
    Bn, hw = 2, 8
    batch = synthetic_train_batch(Bn, hw, 48, 7)
    want_loss, want = O.train_step(sd, cfg, batch, use_refs)
    tr = UNetTrainer(arch, sd, "cpu", Bn, hw, hw, n_ref=3, ref_engine=_OracleRefEngine(sd, cfg, arch, Bn, 3, hw, hw))
    loss, grads = tr.train_step(batch, use_refs)
    assert abs(float(loss) - float(want_loss)) <= 5e-3 * abs(float(want_loss))
    assert set(grads) == set(want) and len(grads) == 5 * len(arch.feature_keys)
    errs = {k: rel_l2(grads[k], want[k]) for k in want}
    assert max(errs.values()) < 2e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:3]


def test_stage1_block_composition_attn1_gradients_without_image_context(cpu_ops):
    """Stage 1 (train_StorySalon_stage1.py:175-179,288): the block runs without image context (attn3 skipped) and returns the five
    attn1 weight gradients — against torch autograd through the oracle's block."""
    from oracle import storygen_oracle as O
    from storygen_amd.train_blocks import TransformerBlockTrain
    from test_backward_gpu import _block_sd
    C, heads, Bn, N, S = 64, 2, 2, 24, 13
    sd = _block_sd(C, 96, 3)
    g = torch.Generator().manual_seed(2)
    h, text, dout = _r(g, Bn, N, C), _r(g, Bn, S, 96), _r(g, Bn, N, C)
    names = [f"b.attn1.{k}" for k in ("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight", "to_out.0.bias")]
    params = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    hin = h.clone().requires_grad_(True)
    want_out, _ = O.transformer_block(params, "b", hin, text, None, heads)
    want = torch.autograd.grad(want_out, [hin] + [params[k] for k in names], dout)
    blk = TransformerBlockTrain(sd, "b", heads, "cpu", trainable="attn1")
    out = blk.forward(h.reshape(Bn * N, C).contiguous(), text.half().reshape(Bn * S, 96).contiguous(), None, Bn)
    assert rel_l2(out.view(Bn, N, C), want_out.detach()) < 3e-3
    dh, grads = blk.backward(dout.reshape(Bn * N, C).contiguous())
    assert rel_l2(dh.view(Bn, N, C), want[0]) < 1e-2
    assert set(grads) == {k[len("b.attn1."):] for k in names}
    for k, w in zip(names, want[1:]):
        assert rel_l2(grads[k[len("b.attn1."):]], w) < 1e-2, k
    with pytest.raises(NotImplementedError):
        TransformerBlockTrain(sd, "b", heads, "cpu", trainable="attn2")


def test_stage1_training_step_composition_vs_oracle(cpu_ops):
    """UNetTrainer(trainable="attn1").train_step(batch, use_refs=()) — no reference pass, no image context — against
    oracle.storygen_oracle.train_step(..., (), trainable="attn1"), itself pinned to the reference's stage-1 step
    (tests/golden/tiny_train_stage1.pt)."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    cfg = load_config(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, attention_head_dim=2,
                           norm_num_groups=8, sample_size=64))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 7)
    Bn, hw = 2, 8
    batch = {k: v for k, v in synthetic_train_batch(Bn, hw, 48, 7).items() if k not in ("ref_latents", "ref_noise", "prev_text")}
    want_loss, want = O.train_step(sd, cfg, batch, (), trainable="attn1")
    tr = UNetTrainer(arch, sd, "cpu", Bn, hw, hw, n_ref=0, trainable="attn1")
    loss, grads = tr.train_step(batch, ())
    assert abs(float(loss) - float(want_loss)) <= 5e-3 * abs(float(want_loss))
    assert set(grads) == set(want) and all(".attn1." in k for k in grads) and len(grads) == 5 * len(arch.feature_keys)
    errs = {k: rel_l2(grads[k], want[k]) for k in want}
    assert max(errs.values()) < 2e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    # the refresh after an optimizer step goes to the attn1 operand copies, in place
    name = next(iter(grads))
    blk = tr.xfs[name.split(".transformer_blocks.")[0]].blk
    before = blk.w["attn1.to_q"].data_ptr()
    tr.set_trainable_parameters({k: torch.zeros_like(sd[k]) for k in grads})
    assert blk.w["attn1.to_q"].data_ptr() == before and float(blk.w["attn1.to_q"].abs().max()) == 0.0


def test_coco_training_step_composition_vs_oracle(cpu_ops):
    """train_COCO.py:286-316 — every frame at noise level ref_t, unmasked loss — through UNetTrainer.ref_levels = "coco"."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    cfg = load_config(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, attention_head_dim=2,
                           norm_num_groups=8, sample_size=64))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 7)
    Bn, hw = 2, 8
    batch = synthetic_train_batch(Bn, hw, 48, 7)
    batch["mask"] = torch.zeros_like(batch["mask"])
    want_loss, want = O.train_step(sd, cfg, batch, (0, 1, 2), ref_levels="coco")
    other_loss, _ = O.train_step(sd, cfg, batch, (0, 1, 2))
    assert abs(float(want_loss) - float(other_loss)) > 1e-4 * abs(float(want_loss))        # the two rules really differ
    tr = UNetTrainer(arch, sd, "cpu", Bn, hw, hw, n_ref=3, ref_engine=_OracleRefEngine(sd, cfg, arch, Bn, 3, hw, hw))
    tr.ref_levels = "coco"
    loss, grads = tr.train_step(batch, (0, 1, 2))
    assert abs(float(loss) - float(want_loss)) <= 5e-3 * abs(float(want_loss))
    assert max(rel_l2(grads[k], want[k]) for k in want) < 2e-2


def test_main_pass_autograd_function_routes_gradients_to_the_attn3_parameters(cpu_ops):
    """storygen_amd.train.MainPassFunction — what the drop-in model's forward uses under autograd: a plain torch loss on
    its output, loss.backward(), and the attn3 leaves must receive the oracle's gradients; nothing else gets one."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import MainPassFunction, UNetTrainer
    cfg = load_config(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, attention_head_dim=2,
                           norm_num_groups=8, sample_size=64))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 9)
    Bn, hw, use_refs = 2, 8, (0, 1, 2)
    batch = synthetic_train_batch(Bn, hw, 48, 9)
    want_loss, want = O.train_step(sd, cfg, batch, use_refs)
    ref = _OracleRefEngine(sd, cfg, arch, Bn, 3, hw, hw)
    tr = UNetTrainer(arch, sd, "cpu", Bn, hw, hw, n_ref=3, ref_engine=ref)
    # the reference passes and the input noising, as the training script does them around the model call
    sched = O.DDIM()
    t = batch["timesteps"].long()
    ref_t = (batch["timesteps"] / 10).long()
    for slot, i in enumerate(use_refs):
        ti = ref_t * (3 - i)
        ref.set_inputs(O.ddpm_add_noise(sched, batch["ref_latents"][i], batch["ref_noise"], ti), ti, batch["prev_text"][i])
        ref.forward(slot)
    noisy = O.ddpm_add_noise(sched, batch["latents"], batch["noise"], t)
    names = sorted(k for k in sd if k.endswith(O.TRAINABLE_SUFFIXES))
    params = [sd[k].clone().requires_grad_(True) for k in names]
    keys = list(ref.ctx)
    feats = [ref.ctx[k].reshape(-1, ref.ctx[k].shape[2]).contiguous() for k in keys]
    text16 = batch["text"].half().reshape(-1, 48).contiguous()
    pred = MainPassFunction.apply(tr, names, keys, noisy.contiguous(), t.float(), text16, *feats, *params)
    keep = 1.0 - batch["mask"]
    loss = F.mse_loss(pred * keep, batch["noise"] * keep)
    loss.backward()
    assert abs(float(loss.detach()) - float(want_loss)) <= 5e-3 * abs(float(want_loss))
    errs = {n: rel_l2(p.grad, want[n]) for n, p in zip(names, params)}
    assert max(errs.values()) < 2e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:3]


def test_main_pass_autograd_function_stage1_routes_gradients_to_attn1(cpu_ops):
    """Stage 1 through the same autograd node: no features, attn1 leaves (train_StorySalon_stage1.py:175-179,288-291)."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import MainPassFunction, UNetTrainer
    cfg = load_config(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, attention_head_dim=2,
                           norm_num_groups=8, sample_size=64))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 9)
    Bn, hw = 2, 8
    batch = synthetic_train_batch(Bn, hw, 48, 9)
    want_loss, want = O.train_step(sd, cfg, batch, (), trainable="attn1")
    tr = UNetTrainer(arch, sd, "cpu", Bn, hw, hw, n_ref=0, trainable="attn1")
    t = batch["timesteps"].long()
    noisy = O.ddpm_add_noise(O.DDIM(), batch["latents"], batch["noise"], t)
    names = sorted(k for k in sd if k.endswith(O.trainable_suffixes("attn1")))
    params = [sd[k].clone().requires_grad_(True) for k in names]
    pred = MainPassFunction.apply(tr, names, [], noisy.contiguous(), t.float(), batch["text"].half().reshape(-1, 48).contiguous(), *params)
    keep = 1.0 - batch["mask"]
    loss = F.mse_loss(pred * keep, batch["noise"] * keep)
    loss.backward()
    assert abs(float(loss.detach()) - float(want_loss)) <= 5e-3 * abs(float(want_loss))
    errs = {n: rel_l2(p.grad, want[n]) for n, p in zip(names, params)}
    assert max(errs.values()) < 2e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:3]
