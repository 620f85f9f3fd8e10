"""HIP backward kernels (BASELINE config 4) against the hand-written CPU backward (oracle/storygen_backward.py, itself
checked against torch.autograd and the reference's gradients in tests/test_oracle_backward.py).

First hardware run: round 2, call 1 (gpurun_out/r2c1): every kernel-level and block-level test green as written."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(shape, dev, scale=1.0, seed=0, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("M,C,dual", [(1024, 320, True), (300, 640, False), (77, 1280, True)])
def test_layernorm_bwd(gpu, M, C, dual):
    from oracle import storygen_backward as B
    from storygen_amd import ops
    x = rnd((M, C), gpu, 2.0, 1, torch.float32) + 0.5
    dy1, dy2 = rnd((M, C), gpu, 1.0, 2), rnd((M, C), gpu, 1.0, 3)
    g1, g2 = rnd((C,), gpu, 1.0, 4), rnd((C,), gpu, 1.0, 5)
    res = rnd((M, C), gpu, 1.0, 6, torch.float32)
    out = torch.empty(M, C, dtype=torch.float32, device=gpu)
    ops.layernorm_bwd(x, dy1, g1, out, 1e-5, dy2 if dual else None, g2 if dual else None, res, 2.0)
    want = 2.0 * res + B.layer_norm_bwd(x, g1.float(), dy1.float())
    if dual:
        want = want + B.layer_norm_bwd(x, g2.float(), dy2.float())
    assert rel(out, want) < 1e-5


def test_geglu_bwd(gpu):
    from oracle import storygen_backward as B
    from storygen_amd import ops
    M, N4 = 200, 256
    val, gate, du = rnd((M, N4), gpu, 1.5, 1), rnd((M, N4), gpu, 1.5, 2), rnd((M, N4), gpu, 1.0, 3)
    il = lambda a, b: torch.stack([a.view(M, -1, 32), b.view(M, -1, 32)], dim=2).reshape(M, 2 * N4)   # noqa: E731
    proj = il(val, gate).contiguous()
    dproj = torch.empty_like(proj)
    ops.geglu_bwd(proj, du, dproj)
    dval = du.float() * F.gelu(gate.float())
    dgate = B.gelu_bwd(gate.float(), du.float() * val.float())
    assert rel(dproj, il(dval, dgate)) < 1e-3


@pytest.mark.parametrize("B_,H,W,C,silu,f32out", [(2, 32, 32, 320, True, True), (3, 16, 16, 1280, False, True), (1, 64, 64, 640, True, False),
                                                  (2, 8, 12, 960, True, False)])
def test_groupnorm_bwd(gpu, B_, H, W, C, silu, f32out):
    from oracle import storygen_backward as B
    from storygen_amd import ops
    HW = H * W
    x = rnd((B_, HW, C), gpu, 2.0, 1, torch.float32) + 1.0
    dy = rnd((B_, HW, C), gpu, 1.0, 2)
    g, b = rnd((C,), gpu, 1.0, 3) + 1.0, rnd((C,), gpu, 0.5, 4)
    ws = torch.empty(ops.groupnorm_bwd_workspace_bytes(B_, 32), dtype=torch.uint8, device=gpu)
    xi = x.transpose(1, 2).reshape(B_, C, H, W)
    dyi = dy.float().transpose(1, 2).reshape(B_, C, H, W)
    if silu:
        n = F.group_norm(xi, 32, g.float(), b.float(), 1e-5)
        dyi = B.silu_bwd(n, dyi)
    want = B.group_norm_bwd(xi, g.float(), dyi, 32, 1e-5).reshape(B_, C, HW).transpose(1, 2)
    if f32out:
        res = rnd((B_, HW, C), gpu, 1.0, 5, torch.float32)
        out = torch.empty(B_, HW, C, dtype=torch.float32, device=gpu)
        ops.groupnorm_bwd(x, dy, g, b, out, 32, 1e-5, silu, ws, res=res)
        assert rel(out, want + res) < 1e-4
    else:
        out = torch.zeros(B_, H + 2, W + 2, C, dtype=torch.float16, device=gpu)
        ops.groupnorm_bwd(x, dy, g, b, out, 32, 1e-5, silu, ws)
        assert rel(out[:, 1:-1, 1:-1].reshape(B_, HW, C), want) < 1e-3
        assert float(out[:, 0].abs().max()) == 0 and float(out[:, :, 0].abs().max()) == 0


@pytest.mark.parametrize("M,C,f32", [(4096, 320, True), (200, 72, False), (64, 1280, False)])
def test_transpose(gpu, M, C, f32):
    from storygen_amd import ops
    src = rnd((M, C), gpu, 1.0, 1, torch.float32 if f32 else torch.float16)
    dst = torch.empty(C, M, dtype=torch.float16, device=gpu)
    ops.transpose(src, dst)
    assert torch.equal(dst, src.half().t())


@pytest.mark.parametrize("f32", [False, True])
def test_transpose_batched_is_the_per_row_transpose_in_one_launch(gpu, f32):
    """sg_transpose_batched_f16 (round 6: the attention operands K^T, Q^T, dO^T, V^T of a training step, 4 launches -> 1) on
    row- and batch-strided views, ragged tile edges."""
    from storygen_amd import ops
    B, M, C = 3, 200, 72
    big = rnd((B, M + 8, C + 8), gpu, 1.0, 5, torch.float32 if f32 else torch.float16)
    src = big[:, :M, :C]                                              # row stride C + 8, batch stride (M + 8)(C + 8)
    dst_big = torch.zeros(B, C, M + 16, dtype=torch.float16, device=gpu)
    dst = dst_big[:, :, :M]
    ops.transpose_batched(src, dst)
    assert torch.equal(dst, src.half().transpose(1, 2))
    assert float(dst_big[:, :, M:].abs().max()) == 0.0                # nothing written beyond the M columns


def test_weight_gradient_as_a_gemm_on_transposes(gpu):
    """dW[n,k] = sum_m dy[m,n] x[m,k] through the forward GEMM kernel."""
    from storygen_amd import ops
    M, N, K = 4096, 320, 640
    dy, x = rnd((M, N), gpu, 1.0, 1), rnd((M, K), gpu, 1.0, 2)
    dyt, xt = torch.empty(N, M, dtype=torch.float16, device=gpu), torch.empty(K, M, dtype=torch.float16, device=gpu)
    ops.transpose(dy, dyt), ops.transpose(x, xt)
    dw = torch.empty(N, K, dtype=torch.float32, device=gpu)
    ops.gemm(dyt, xt, dw)
    assert rel(dw, dy.float().t() @ x.float()) < 1e-3


def test_conv_dgrad_through_the_forward_conv_kernel(gpu):
    """Stride-1, stride-2 (zero-stuffed) and upsample (sum2x2) dgrads: forward kernel + rotated weights vs autograd."""
    from storygen_amd import ops
    from storygen_amd.repack import conv3x3_krsc
    B_, H, W, Ci, Co = 2, 16, 16, 64, 128
    w = rnd((Co, Ci, 3, 3), gpu, 0.05, 1)
    wt = conv3x3_krsc(w.flip(2, 3).transpose(0, 1).contiguous())                 # dgrad weight: [Ci, 3, 3, Co]
    for stride in (1, 2):
        Ho, Wo = H // stride, W // stride
        dy = rnd((B_, Ho, Wo, Co), gpu, 1.0, 2)
        x = torch.zeros(B_, Ci, H, W, device=gpu, requires_grad=True)
        y = F.conv2d(x, w.float(), stride=stride, padding=1)
        want = torch.autograd.grad(y, x, dy.float().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
        pad = torch.zeros(B_, H + 2, W + 2, Co, dtype=torch.float16, device=gpu)
        if stride == 1:
            ops.pad_cast(dy, pad)
        else:
            ops.zero_stuff(dy, pad)
        dx = torch.empty(B_, H, W, Ci, dtype=torch.float32, device=gpu)
        ops.conv3x3(pad, wt, dx, x_padded=True)
        assert rel(dx, want) < 1e-3, stride
    # nearest-2x upsampling followed by the convolution
    dy = rnd((B_, 2 * H, 2 * W, Co), gpu, 1.0, 3)
    x = torch.zeros(B_, Ci, H, W, device=gpu, requires_grad=True)
    y = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w.float(), padding=1)
    want = torch.autograd.grad(y, x, dy.float().permute(0, 3, 1, 2))[0].permute(0, 2, 3, 1)
    pad = torch.zeros(B_, 2 * H + 2, 2 * W + 2, Co, dtype=torch.float16, device=gpu)
    ops.pad_cast(dy, pad)
    du = torch.empty(B_, 2 * H, 2 * W, Ci, dtype=torch.float32, device=gpu)
    ops.conv3x3(pad, wt, du, x_padded=True)
    dx = torch.empty(B_, H, W, Ci, dtype=torch.float32, device=gpu)
    ops.sum2x2(du, dx)
    assert rel(dx, want) < 1e-3


def test_mse_grad(gpu):
    from storygen_amd import ops
    pred, noise = rnd((4, 4, 64, 64), gpu, 1.0, 1, torch.float32), rnd((4, 4, 64, 64), gpu, 1.0, 2, torch.float32)
    mask = (rnd((4, 4, 64, 64), gpu, 1.0, 3, torch.float32) > 0.5).float()
    d, loss = torch.empty_like(pred), torch.empty(1, device=gpu)
    ops.mse_grad(pred, noise, mask, d, loss)
    p = pred.clone().requires_grad_(True)
    keep = 1.0 - mask
    want = F.mse_loss(p * keep, noise * keep)
    assert abs(float(loss) - float(want)) < 1e-5 * float(want)
    assert rel(d, torch.autograd.grad(want, p)[0]) < 1e-5


@pytest.mark.parametrize("B_,heads,D,Nq,Nk", [(2, 8, 40, 256, 320), (1, 8, 40, 1024, 776), (2, 8, 80, 256, 256), (1, 8, 160, 64, 192),
                                             (4, 8, 40, 4096, 4096)])
def test_attention_backward(gpu, B_, heads, D, Nq, Nk):
    """lse2 from the training forward, (lse2, delta) pairs, dQ, and transposed dK / dV against the CPU formulas
    (oracle/storygen_backward.py::attention_core_bwd, itself checked against autograd)."""
    from oracle import storygen_backward as B
    from storygen_amd import ops
    C = heads * D
    big = Nq * Nk > 4_000_000
    q, k, v = rnd((B_, Nq, C), gpu, 1.0, 1), rnd((B_, Nk, C), gpu, 1.0, 2), rnd((B_, Nk, C), gpu, 1.0, 3)
    do = rnd((B_, Nq, C), gpu, 1.0, 4)
    vt = v.transpose(1, 2).contiguous()
    o = torch.empty_like(q)
    lse2 = torch.empty(B_, heads, Nq, dtype=torch.float32, device=gpu)
    scale = D ** -0.5
    ops.attention_lse(q, k, vt, o, lse2, heads, scale)
    with torch.no_grad():
        dev = gpu if big else "cpu"              # the large case checks against the same formulas evaluated on the device
        qf, kf, vf, dof = (t.to(dev).float() for t in (q, k, v, do))
        o_ref, lse_ref = B.attention_core(qf, kf, vf, heads)
        dq_ref, dk_ref, dv_ref = B.attention_core_bwd(qf, kf, vf, o_ref, lse_ref, dof, heads)
    assert rel(o.to(dev), o_ref) < 2e-3
    assert float((lse2.to(dev) - lse_ref * 1.4426950408889634).abs().max()) < 2e-3
    ld2 = torch.empty(B_, heads, Nq, 2, dtype=torch.float32, device=gpu)
    ops.attention_bwd_prep(o, do, lse2, ld2, heads)
    delta_ref = (dof * o_ref).reshape(B_, Nq, heads, D).sum(-1).transpose(1, 2)
    assert torch.equal(ld2[..., 0], lse2) and rel(ld2[..., 1].to(dev), delta_ref) < 5e-3
    dq = torch.empty_like(q)
    ops.attention_bwd_dq(q, k, k.transpose(1, 2).contiguous(), v, do, ld2, dq, heads, scale)
    assert rel(dq.to(dev), dq_ref) < 5e-3
    dkt, dvt = (torch.empty(B_, C, Nk, dtype=torch.float16, device=gpu) for _ in range(2))
    ops.attention_bwd_dkv(q, q.transpose(1, 2).contiguous(), k, v, do, do.transpose(1, 2).contiguous(), ld2, dkt, dvt, heads, scale)
    assert rel(dkt.transpose(1, 2).to(dev), dk_ref) < 5e-3
    assert rel(dvt.transpose(1, 2).to(dev), dv_ref) < 5e-3


def _block_sd(C, ctx_dim, seed, dev="cpu"):
    """Random BasicTransformerBlock parameters (fp16-exact values so that device and oracle see the same numbers)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).half().float()      # noqa: E731
    sd = {}
    for n in ("norm1", "norm2", "norm3", "norm4"):
        sd[f"b.{n}.weight"], sd[f"b.{n}.bias"] = 1.0 + r(C, sc=0.1), r(C, sc=0.1)
    for a, kd in (("attn1", C), ("attn2", ctx_dim), ("attn3", C)):
        sd[f"b.{a}.to_q.weight"] = r(C, C, sc=C ** -0.5)
        sd[f"b.{a}.to_k.weight"], sd[f"b.{a}.to_v.weight"] = r(C, kd, sc=kd ** -0.5), r(C, kd, sc=kd ** -0.5)
        sd[f"b.{a}.to_out.0.weight"], sd[f"b.{a}.to_out.0.bias"] = r(C, C, sc=C ** -0.5), r(C, sc=0.1)
    sd["b.ff.net.0.proj.weight"], sd["b.ff.net.0.proj.bias"] = r(8 * C, C, sc=C ** -0.5), r(8 * C, sc=0.1)
    sd["b.ff.net.2.weight"], sd["b.ff.net.2.bias"] = r(C, 4 * C, sc=(4 * C) ** -0.5), r(C, sc=0.1)
    return sd


def test_transformer_block_training_forward_and_backward(gpu):
    """storygen_amd.train_blocks.TransformerBlockTrain (the composition of the backward kernels) against the hand-written
    CPU backward of the same block: output, input gradient and the five attn3 parameter gradients."""
    from oracle import storygen_backward as B
    from oracle import storygen_oracle as O
    from storygen_amd.train_blocks import TransformerBlockTrain
    C, heads, Bn, N, S, Nc = 320, 8, 2, 256, 77, 512
    sd = _block_sd(C, 768, 3)
    h = rnd((Bn, N, C), "cpu", 1.0, 1, torch.float32)
    text, ctx = rnd((Bn, S, 768), "cpu", 1.0, 2).float(), rnd((Bn, Nc, C), "cpu", 1.0, 3).float()
    dout = rnd((Bn, N, C), "cpu", 1.0, 4, torch.float32)
    with torch.no_grad():
        want_out, _ = O.transformer_block(sd, "b", h, text, ctx, heads)
        want_dh, want_g = B.transformer_block_bwd(sd, "b", h, text, ctx, heads, dout)
    blk = TransformerBlockTrain(sd, "b", heads, gpu)
    out = blk.forward(h.to(gpu).reshape(Bn * N, C).contiguous(), text.half().to(gpu).reshape(Bn * S, 768).contiguous(),
                      ctx.half().to(gpu).reshape(Bn * Nc, C).contiguous(), Bn)
    assert rel(out.cpu().view(Bn, N, C), want_out) < 2e-3
    dh, grads = blk.backward(dout.to(gpu).reshape(Bn * N, C).contiguous())
    torch.cuda.synchronize()
    assert rel(dh.cpu().view(Bn, N, C), want_dh) < 1e-2
    for k, g in grads.items():
        assert rel(g.cpu(), want_g[f"b.attn3.{k}"]) < 1e-2, k


@pytest.mark.parametrize("cin,cout", [(320, 320), (640, 320)])
def test_resnet_block_training_forward_and_backward(gpu, cin, cout):
    from oracle import storygen_backward as B
    from oracle import storygen_oracle as O
    from storygen_amd.train_blocks import ResnetBlockTrain
    Bn, H, W, temb = 2, 16, 16, 1280
    g = torch.Generator().manual_seed(5)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).half().float()      # noqa: E731
    sd = {"r.norm1.weight": 1.0 + r(cin, sc=0.1), "r.norm1.bias": r(cin, sc=0.1), "r.norm2.weight": 1.0 + r(cout, sc=0.1),
          "r.norm2.bias": r(cout, sc=0.1), "r.conv1.weight": r(cout, cin, 3, 3, sc=(9 * cin) ** -0.5), "r.conv1.bias": r(cout, sc=0.1),
          "r.conv2.weight": r(cout, cout, 3, 3, sc=(9 * cout) ** -0.5), "r.conv2.bias": r(cout, sc=0.1),
          "r.time_emb_proj.weight": r(cout, temb, sc=temb ** -0.5), "r.time_emb_proj.bias": r(cout, sc=0.1)}
    if cin != cout:
        sd["r.conv_shortcut.weight"], sd["r.conv_shortcut.bias"] = r(cout, cin, 1, 1, sc=cin ** -0.5), r(cout, sc=0.1)
    x = rnd((Bn, cin, H, W), "cpu", 1.0, 1, torch.float32) + 0.3
    emb, dout = rnd((Bn, temb), "cpu", 1.0, 2, torch.float32), rnd((Bn, cout, H, W), "cpu", 1.0, 3, torch.float32)
    with torch.no_grad():
        want = O.resnet_block(sd, "r", x, emb, 32, 1e-5)
        want_dx = B.resnet_block_bwd(sd, "r", x, emb, 32, 1e-5, dout)
        tproj = F.linear(F.silu(emb), sd["r.time_emb_proj.weight"], sd["r.time_emb_proj.bias"])
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(Bn * H * W, -1).contiguous().to(gpu)       # noqa: E731
    blk = ResnetBlockTrain(sd, "r", 32, 1e-5, gpu)
    out = blk.forward(nhwc(x), tproj.to(gpu), Bn, H, W)
    assert rel(out.cpu(), nhwc(want).cpu()) < 2e-3
    dx = blk.backward(nhwc(dout))
    torch.cuda.synchronize()
    assert rel(dx.cpu(), nhwc(want_dx).cpu()) < 1e-2


@pytest.mark.parametrize("use_refs", [(0, 1, 2), (2,)])
def test_training_step_vs_oracle(gpu, use_refs):
    """BASELINE config 4 end to end on a 2-level StoryGen UNet (320 / 640 channels, 16x16 latent, batch 2): loss and the
    attn3 gradients of storygen_amd.train.UNetTrainer against oracle.storygen_oracle.train_step (autograd; pinned to the
    reference's gradients by tests/test_oracle_golden.py).  Bar: 1e-2 rel-L2 per gradient tensor."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    cfg = load_config(dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=768, attention_head_dim=8,
                           sample_size=128))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 7)
    batch = synthetic_train_batch(2, 16, 768, 7)
    want_loss, want = O.train_step(sd, cfg, batch, use_refs)
    tr = UNetTrainer(arch, sd, gpu, 2, 16, 16, n_ref=3)
    loss, grads = tr.train_step(batch, use_refs)
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want_loss)) <= 2e-3 * abs(float(want_loss))
    assert set(grads) == set(want)
    errs = {k: rel(grads[k].cpu(), want[k]) for k in want}
    print("worst gradients:", sorted(errs.items(), key=lambda kv: -kv[1])[:3])
    assert max(errs.values()) < 1e-2


def test_stage1_training_step_vs_oracle(gpu):
    """Stage 1 (train_StorySalon_stage1.py:175-179,262-291) on the same 2-level UNet: no reference pass, main pass without image
    context, the 30 attn1 gradients against oracle.storygen_oracle.train_step(..., (), trainable="attn1") — pinned to the reference's
    own stage-1 step by tests/golden/tiny_train_stage1.pt."""
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    cfg = load_config(dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=768, attention_head_dim=8,
                           sample_size=128))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 7)
    batch = {k: v for k, v in synthetic_train_batch(2, 16, 768, 7).items() if k not in ("ref_latents", "ref_noise", "prev_text")}
    want_loss, want = O.train_step(sd, cfg, batch, (), trainable="attn1")
    tr = UNetTrainer(arch, sd, gpu, 2, 16, 16, n_ref=0, trainable="attn1")
    loss, grads = tr.train_step(batch, ())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(want_loss)) <= 2e-3 * abs(float(want_loss))
    assert set(grads) == set(want) and all(".attn1." in k for k in grads)
    errs = {k: rel(grads[k].cpu(), want[k]) for k in want}
    print("stage 1, worst gradients:", sorted(errs.items(), key=lambda kv: -kv[1])[:3])
    assert max(errs.values()) < 1e-2
    l2, g2 = tr.train_step_graph(batch, ())                       # the captured step returns the same numbers
    torch.cuda.synchronize()
    assert abs(float(l2) - float(loss)) <= 1e-5 * abs(float(loss))
    assert max(rel(g2[k], grads[k]) for k in grads) < 1e-3


def test_training_step_graph_replay_matches_eager(gpu):
    """UNetTrainer.train_step_graph — the whole stage-2 step (reference passes, main pass, loss, backward) captured once and
    replayed as one hipGraph — must return the eager step's loss and 80 gradients, also for a NEW batch fed to the captured graph."""
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    cfg = load_config(dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=768, attention_head_dim=8,
                           sample_size=128))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 7)
    b1, b2 = synthetic_train_batch(2, 16, 768, 7), synthetic_train_batch(2, 16, 768, 8)
    tr = UNetTrainer(arch, sd, gpu, 2, 16, 16, n_ref=3)
    tr.train_step(b1)                                   # settles the automatic gradient scale
    tr.grad_scale = float(tr.last_grad_scale)           # same fixed scale for both paths
    want = [(float(l), {k: v.clone() for k, v in g.items()}) for l, g in (tr.train_step(b1), tr.train_step(b2))]
    for b, (wl, wg) in zip((b1, b2), want):             # first call captures, second replays with new inputs
        loss, grads = tr.train_step_graph(b)
        torch.cuda.synchronize()
        assert abs(float(loss) - wl) <= 1e-6 * abs(wl)
        assert max(rel(grads[k], wg[k]) for k in wg) < 1e-6
    assert len(tr._graphs) == 1


def test_training_step_graph_skips_a_step_with_non_finite_gradients_and_regrows_the_scale(gpu):
    """the graph path must behave like the reference's fp16 GradScaler (train_StorySalon_stage2.py:138-141,328) — a batch
    whose gradients are non-finite at every loss scale is SKIPPED (flag, no exception), all captured graphs share one scale, and the
    scale grows back after `scale_growth_interval` good steps."""
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    cfg = load_config(dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=768, attention_head_dim=8,
                           sample_size=128))
    arch = build_arch(cfg)
    tr = UNetTrainer(arch, synthetic_state_dict(arch, 7), gpu, 2, 16, 16, n_ref=3)
    good = synthetic_train_batch(2, 16, 768, 7)
    loss, grads = tr.train_step_graph(good)
    torch.cuda.synchronize()
    assert not tr.last_step_skipped and all(bool(torch.isfinite(g).all()) for g in grads.values())
    scale0 = float(tr.grad_scale)
    tr.train_step_graph(good, use_refs=(2,))                          # a second captured variant at the same scale
    assert len(tr._graphs) == 2
    bad = dict(good)
    bad["noise"] = good["noise"].clone()
    bad["noise"].view(-1)[0] = float("nan")                           # the loss gradient is NaN whatever the scale
    loss, grads = tr.train_step_graph(bad)
    torch.cuda.synchronize()
    assert tr.last_step_skipped, "a batch with NaN gradients must be reported as skipped, not raise"
    assert float(tr.grad_scale) == scale0 and len(tr._graphs) == 2, "a non-finite LOSS is the batch's fault: the scale stays"
    # an overflow of the scaled backward (finite loss): the scale drops 16x and every captured variant is dropped with it
    tr.grad_scale = scale0 * 2.0 ** 24                                # (the two variants captured at scale0 are still on file)
    tr._graphs.pop((0, 1, 2))
    loss, grads = tr.train_step_graph(good)
    torch.cuda.synchronize()
    assert not tr.last_step_skipped and all(bool(torch.isfinite(g).all()) for g in grads.values())
    assert float(tr.grad_scale) < scale0 * 2.0 ** 24 and len(tr._graphs) == 1, "the lowered scale invalidates every captured variant"
    # good batches again: not skipped, and after `scale_growth_interval` of them the scale doubles
    tr.scale_growth_interval = 3
    tr._good_steps = 0
    lowered = float(tr.grad_scale)
    for _ in range(3):
        loss, grads = tr.train_step_graph(good)
        torch.cuda.synchronize()
        assert not tr.last_step_skipped and all(bool(torch.isfinite(g).all()) for g in grads.values())
    assert float(tr.grad_scale) == 2.0 * lowered and not tr._graphs


def test_training_step_at_baseline_config4_size_vs_reference_golden(gpu):
    """BASELINE config 4 AT ITS OWN SIZE: SD-1.5 UNet, 64x64 latent, batch 4, reference frames (0, 1, 2) — loss and all 80 attn3
    gradients of UNetTrainer.train_step_graph (the whole step as one hipGraph, loss scaling included) against the gradients the
    REFERENCE's own UNet + torch.autograd produced on CPU fp32 (tests/golden/sd15_train_bs4.pt, oracle/make_golden_train_sd15.py).
    Here the backward attention sees Nq 4096 x Nk 12 288 at D = 40 — the regime round 2 never compared with anything.
    Bars: loss 2e-3; every gradient 1e-2 (L2 norm, the 512-entry index sample, and the full tensor for the 9 stored ones)."""
    import os
    from storygen_amd.arch import build_arch
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from storygen_amd.train import UNetTrainer
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sd15_train_bs4.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated")
    gold = torch.load(path, weights_only=False)
    arch = build_arch(gold["config"])
    sd = synthetic_state_dict(arch, gold["seed"])
    B, hw = gold["batch"], gold["hw"]
    batch = synthetic_train_batch(B, hw, arch.config["cross_attention_dim"], gold["seed"])
    tr = UNetTrainer(arch, sd, gpu, B, hw, hw, n_ref=3)
    loss, grads = tr.train_step_graph(batch, tuple(gold["use_refs"]))
    torch.cuda.synchronize()
    print(f"config 4 size: loss {float(loss):.6f} vs reference {gold['loss']:.6f}; gradient scale {tr.last_grad_scale:g}")
    assert abs(float(loss) - gold["loss"]) <= 2e-3 * abs(gold["loss"])
    assert set(grads) == set(gold["grads"])
    errs, nfull = {}, 0
    for k, e in gold["grads"].items():
        g = grads[k].float().cpu()
        assert tuple(g.shape) == tuple(e["shape"]) and torch.isfinite(g).all(), k
        errs[k] = max(abs(float(g.double().norm()) - e["l2"]) / e["l2"], rel(g.flatten()[e["idx"]], e["values"]))
        if "full" in e:
            errs[k] = max(errs[k], rel(g, e["full"]))
            nfull += 1
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print("worst gradients:", [(k.replace("transformer_blocks.0.", ""), f"{v:.2e}") for k, v in worst], f"({nfull} compared in full)")
    assert nfull >= 8 and max(errs.values()) < 1e-2
