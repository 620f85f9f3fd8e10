"""The hand-written backward pass (oracle/storygen_backward.py: the formulas the HIP backward kernels of BASELINE config 4
will implement, in the order the device engine will run them) against torch.autograd: every leaf formula on random
inputs in fp64, then the whole training step against storygen_oracle.train_step — which tests/test_oracle_golden.py pins
to the reference's own UNet + autograd through tests/golden/tiny_train.pt.  CPU only."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

D = torch.float64


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=D) * scale).requires_grad_(True)


def _check(got, want, tol=1e-10):
    assert got.shape == want.shape and rel_l2(got, want) < tol, rel_l2(got, want)


def test_activation_backward_formulas():
    from oracle import storygen_backward as B
    x, dy = _rnd(5, 37, seed=1, scale=2.0), _rnd(5, 37, seed=2)
    _check(B.silu_bwd(x.detach(), dy.detach()), torch.autograd.grad(F.silu(x), x, dy)[0])
    _check(B.gelu_bwd(x.detach(), dy.detach()), torch.autograd.grad(F.gelu(x), x, dy)[0])


@pytest.mark.parametrize("b,c,h,w,groups,eps", [(2, 32, 5, 7, 8, 1e-5), (1, 64, 4, 4, 32, 1e-6), (3, 20, 3, 3, 2, 1e-5)])
def test_group_norm_backward(b, c, h, w, groups, eps):
    from oracle import storygen_backward as B
    x = _rnd(b, c, h, w, seed=1, scale=1.7)
    x.data += 0.8
    gamma, beta, dy = _rnd(c, seed=2), _rnd(c, seed=3), _rnd(b, c, h, w, seed=4)
    want = torch.autograd.grad(F.group_norm(x, groups, gamma, beta, eps), x, dy)[0]
    _check(B.group_norm_bwd(x.detach(), gamma.detach(), dy.detach(), groups, eps), want)


def test_layer_norm_backward():
    from oracle import storygen_backward as B
    x, gamma, beta, dy = _rnd(3, 11, 40, seed=1, scale=2.0), _rnd(40, seed=2), _rnd(40, seed=3), _rnd(3, 11, 40, seed=4)
    want = torch.autograd.grad(F.layer_norm(x, (40,), gamma, beta, 1e-5), x, dy)[0]
    _check(B.layer_norm_bwd(x.detach(), gamma.detach(), dy.detach()), want)


@pytest.mark.parametrize("stride,k", [(1, 3), (2, 3), (1, 1)])
def test_conv_dgrad_as_a_forward_convolution(stride, k):
    from oracle import storygen_backward as B
    x, w = _rnd(2, 6, 8, 10, seed=1), _rnd(5, 6, k, k, seed=2)
    y = F.conv2d(x, w, stride=stride, padding=k // 2)
    dy = _rnd(*y.shape, seed=3)
    _check(B.conv_dgrad(dy.detach(), w.detach(), stride), torch.autograd.grad(y, x, dy)[0])


def test_upsample_conv_backward():
    from oracle import storygen_backward as B
    x, w = _rnd(2, 4, 5, 6, seed=1), _rnd(3, 4, 3, 3, seed=2)
    y = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    dy = _rnd(*y.shape, seed=3)
    _check(B.upsample2x_bwd(B.conv_dgrad(dy.detach(), w.detach())), torch.autograd.grad(y, x, dy)[0])


@pytest.mark.parametrize("nq,nk,heads,d", [(10, 10, 2, 8), (7, 19, 4, 5)])
def test_attention_core_backward(nq, nk, heads, d):
    """dQ, dK, dV from (Q, K, V, O, log-sum-exp rows, dO) — the flash-attention backward the HIP kernel will implement."""
    from oracle import storygen_backward as B
    c = heads * d
    q, k, v, do = _rnd(2, nq, c, seed=1), _rnd(2, nk, c, seed=2), _rnd(2, nk, c, seed=3), _rnd(2, nq, c, seed=4)
    qh, kh, vh = (t.reshape(2, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    o_ref = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).transpose(1, 2).reshape(2, nq, c)
    want = torch.autograd.grad(o_ref, (q, k, v), do)
    with torch.no_grad():
        o, lse = B.attention_core(q, k, v, heads)
        _check(o, o_ref)
        got = B.attention_core_bwd(q, k, v, o, lse, do, heads)
    for g, wnt in zip(got, want):
        _check(g, wnt)


@pytest.mark.parametrize("self_attn", [True, False])
def test_attention_module_backward_and_weight_gradients(self_attn):
    from oracle import storygen_backward as B
    from oracle import storygen_oracle as O
    c, cc, heads = 16, 16 if self_attn else 24, 2
    sd = {"a.to_q.weight": _rnd(c, c, seed=1, scale=0.3), "a.to_k.weight": _rnd(c, cc, seed=2, scale=0.3),
          "a.to_v.weight": _rnd(c, cc, seed=3, scale=0.3), "a.to_out.0.weight": _rnd(c, c, seed=4, scale=0.3),
          "a.to_out.0.bias": _rnd(c, seed=5)}
    x, ctx, dy = _rnd(2, 9, c, seed=6), None if self_attn else _rnd(2, 13, cc, seed=7).detach(), _rnd(2, 9, c, seed=8)
    y = O.attention(sd, "a", x, ctx, heads)
    names = list(sd)
    want = torch.autograd.grad(y, [x] + [sd[n] for n in names], dy)
    with torch.no_grad():
        dx, grads = B.attention_module_bwd({k: v.detach() for k, v in sd.items()}, "a", x.detach(), ctx, heads, dy.detach(), True)
    _check(dx, want[0])
    for n, wnt in zip(names, want[1:]):
        _check(grads[n], wnt)


@pytest.mark.parametrize("use_refs", [(2,), pytest.param((0, 1, 2), marks=pytest.mark.skipif(os.environ.get("SG_SLOW_TESTS") != "1",
                                                                                               reason="~1 min of CPU; set SG_SLOW_TESTS=1"))])
def test_explicit_training_step_matches_autograd_and_the_reference_golden(use_refs):
    """The whole chain: loss and all 80 attn3 gradients of the hand-written backward vs storygen_oracle.train_step
    (autograd) and vs what the reference's own UNet + autograd produced (tests/golden/tiny_train.pt)."""
    from oracle import storygen_backward as B
    from oracle import storygen_oracle as O
    from storygen_amd.arch import build_arch
    from storygen_amd.synth import synthetic_state_dict, synthetic_train_batch
    from test_oracle_golden import _load
    gold = _load("tiny_train")
    arch = build_arch(gold["config"])
    sd = synthetic_state_dict(arch, gold["seed"])
    batch = synthetic_train_batch(gold["batch"], gold["hw"], arch.config["cross_attention_dim"], gold["seed"])
    loss, grads = B.train_step_explicit(sd, arch.config, batch, use_refs)
    loss_ag, grads_ag = O.train_step(sd, arch.config, batch, use_refs)
    assert abs(float(loss) - float(loss_ag)) <= 1e-6 * abs(float(loss_ag))
    assert set(grads) == set(grads_ag) and len(grads) == 5 * len(arch.feature_keys)
    errs = {k: rel_l2(grads[k], grads_ag[k]) for k in grads}
    assert max(errs.values()) < 2e-4, max(errs.items(), key=lambda kv: kv[1])       # fp32, different summation orders
    g = gold["cases"]["refs_" + "".join(map(str, use_refs))]
    assert abs(float(loss) - g["loss"]) <= 1e-5 * abs(g["loss"])
    errs = {k: rel_l2(grads[k].flatten()[e["idx"]], e["values"]) for k, e in g["grads"].items()}
    assert max(errs.values()) < 3e-4, max(errs.items(), key=lambda kv: kv[1])
