"""TEST INFRASTRUCTURE: CPU stand-ins for the storygen_amd.ops entry points, written from the contracts in
include/storygen_hip.h (layouts included: zero-bordered conv inputs, transposed V, 32/32-interleaved GEGLU columns,
transposed dK / dV) with plain fp32 torch.  They let the HOST-SIDE composition of the training step
(storygen_amd/train_blocks.py, storygen_amd/train.py: op order, weight transposes, residual wiring, weight-gradient
contractions, the tape) run and be checked against the oracle without a GPU.  They say nothing about the kernels
themselves — those have their own hardware tests (tests/test_backward_gpu.py) and, for the attention backward, the
lane-level emulation in tools/emulate_attention_bwd.py."""
import math

import torch
import torch.nn.functional as F

EPI_LINEAR, EPI_GEGLU = 0, 1
LOG2E = 1.4426950408889634


def _store(out, val):
    out.copy_(val.to(out.dtype))
    return out


def gemm_workspace_bytes(M, N, split_k=0):
    return 0


def groupnorm_workspace_bytes(B, groups):
    return 16


def groupnorm_bwd_workspace_bytes(B, groups):
    return 16


def gemm(a, w, out, *, bias=None, rowbias=None, rows_per_batch=1, res1=None, res2=None, epilogue=EPI_LINEAR, split_k=0,
         workspace=None, out2=None, tile=None):
    v = a.float() @ w.float().t()
    if bias is not None:
        v = v + bias.float()
    if epilogue == EPI_GEGLU:      # columns come in blocks of 64: 32 values then their 32 gates -> val * gelu(gate)
        M, N = v.shape
        v = v.view(M, N // 64, 2, 32)
        v = (v[:, :, 0] * F.gelu(v[:, :, 1])).reshape(M, N // 2)
    else:
        if res1 is not None:
            v = v + res1.float()
        if res2 is not None:
            v = v + res2.float()
    if out2 is not None:
        _store(out2, v)
    return _store(out, v)


def layernorm(x, g1, b1, y1, eps=1e-5, g2=None, b2=None, y2=None):
    C = x.shape[-1]
    _store(y1, F.layer_norm(x.float(), (C,), g1.float(), b1.float(), eps))
    if y2 is not None:
        _store(y2, F.layer_norm(x.float(), (C,), g2.float(), b2.float(), eps))


def _heads(t, H):
    B, N, C = t.shape
    return t.float().reshape(B, N, H, C // H).transpose(1, 2)


def attention_lse(q, k, vt, out, lse2, heads, scale, nk=None):
    B, Nq, C = q.shape
    Nk = k.shape[1] if nk is None else nk
    v = vt[:, :, :Nk].transpose(1, 2)                       # V^T [B, C, Nk'] -> [B, Nk, C]
    s = _heads(q, heads) @ _heads(k[:, :Nk], heads).transpose(-1, -2) * scale
    lse = torch.logsumexp(s, -1)
    o = torch.exp(s - lse[..., None]) @ _heads(v, heads)
    _store(lse2, lse * LOG2E)
    return _store(out, o.transpose(1, 2).reshape(B, Nq, C))


def attention_bwd_prep(o, dout, lse2, ld2, heads):
    B, Nq, C = o.shape
    delta = (o.float() * dout.float()).reshape(B, Nq, heads, C // heads).sum(-1).transpose(1, 2)
    ld2[..., 0] = lse2
    ld2[..., 1] = delta
    return ld2


def _p_ds(q, k, v, dout, ld2, heads, scale):
    qh, kh, vh, doh = (_heads(t, heads) for t in (q, k, v, dout))
    p = torch.exp2((qh @ kh.transpose(-1, -2)) * (scale * LOG2E) - ld2[..., 0][..., None])
    ds = p * (doh @ vh.transpose(-1, -2) - ld2[..., 1][..., None])
    return qh, kh, doh, p, ds


def attention_bwd_dq(q, k, kt, v, dout, ld2, dq, heads, scale):
    B, Nq, C = q.shape
    assert kt.shape[0] == B and kt.shape[1] == C and torch.equal(kt[:, :, :k.shape[1]].transpose(1, 2), k), "kt must be K transposed"
    _, kh, _, _, ds = _p_ds(q, k, v, dout, ld2, heads, scale)
    return _store(dq, ((ds @ kh) * scale).transpose(1, 2).reshape(B, Nq, C))


def attention_bwd_dkv(q, qt, k, v, dout, dot, ld2, dkt, dvt, heads, scale):
    B, Nk, C = k.shape
    assert torch.equal(qt[:, :, :q.shape[1]].transpose(1, 2), q) and torch.equal(dot[:, :, :q.shape[1]].transpose(1, 2), dout)
    qh, _, doh, p, ds = _p_ds(q, k, v, dout, ld2, heads, scale)
    dk = ((ds.transpose(-1, -2) @ qh) * scale).transpose(1, 2).reshape(B, Nk, C)
    dv = (p.transpose(-1, -2) @ doh).transpose(1, 2).reshape(B, Nk, C)
    _store(dkt, dk.transpose(1, 2))
    _store(dvt, dv.transpose(1, 2))
    return dkt, dvt


def _gn(x, gamma, beta, groups, eps, silu):
    B, HW, C = x.shape
    y = F.group_norm(x.float().transpose(1, 2), groups, gamma.float(), beta.float(), eps)
    return (F.silu(y) if silu else y).transpose(1, 2)


def groupnorm(x, gamma, beta, out, groups, eps, silu, workspace, xcopy=None):
    y = _gn(x, gamma, beta, groups, eps, silu)
    if xcopy is not None:
        _store(xcopy, x)
    if out.dim() == 4:                                       # zero-bordered image: interior only
        B, Hp, Wp, C = out.shape
        out[:, 1:-1, 1:-1] = y.reshape(B, Hp - 2, Wp - 2, C).to(out.dtype)
        return out
    return _store(out, y)


def groupnorm_bwd(x, dy, gamma, beta, out, groups, eps, silu, workspace, res=None):
    xr = x.float().detach().requires_grad_(True)
    with torch.enable_grad():
        y = _gn(xr, gamma, beta, groups, eps, silu)
        (dx,) = torch.autograd.grad(y, xr, dy.float())
    if res is not None:
        dx = dx + res.float()
    if out.dim() == 4:
        B, Hp, Wp, C = out.shape
        out[:, 1:-1, 1:-1] = dx.reshape(B, Hp - 2, Wp - 2, C).to(out.dtype)
        return out
    return _store(out, dx)


def layernorm_bwd(x, dy1, g1, out, eps=1e-5, dy2=None, g2=None, res=None, res_scale=1.0):
    C = x.shape[-1]
    xr = x.float().detach().requires_grad_(True)
    with torch.enable_grad():
        xhat = F.layer_norm(xr, (C,), None, None, eps)
        g = dy1.float() * g1.float() + (dy2.float() * g2.float() if dy2 is not None else 0.0)
        (dx,) = torch.autograd.grad(xhat, xr, g)
    if res is not None:
        dx = dx + res_scale * res.float()
    return _store(out, dx)


def geglu_bwd(proj_il, du, dproj_il):
    M, N8 = proj_il.shape
    pr = proj_il.float().view(M, N8 // 64, 2, 32).detach()
    val, gate = pr[:, :, 0].clone(), pr[:, :, 1].clone().requires_grad_(True)
    with torch.enable_grad():
        gl = F.gelu(gate)
        (dgelu,) = torch.autograd.grad(gl, gate, torch.ones_like(gl))
    d = du.float().view(M, N8 // 64, 32)
    return _store(dproj_il, torch.stack([d * gl.detach(), d * val * dgelu], dim=2).reshape(M, N8))


def conv3x3(x, w_krsc, out, *, stride=1, upsample2x=False, bias=None, rowbias=None, res1=None, split_k=0, workspace=None,
            x_padded=False, tile=None):
    xi = (x[:, 1:-1, 1:-1] if x_padded else x).float().permute(0, 3, 1, 2)
    if upsample2x:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    w = w_krsc.float().permute(0, 3, 1, 2)                   # [Cout, 3, 3, Cin] -> [Cout, Cin, 3, 3]
    y = F.conv2d(xi, w, None if bias is None else bias.float(), stride=stride, padding=1)
    if rowbias is not None:
        y = y + rowbias.float()[:, :, None, None]
    y = y.permute(0, 2, 3, 1)
    if res1 is not None:
        y = y + res1.float()
    return _store(out, y)


def conv_in(x_nchw, w_kn, bias, out):
    cin, cout = x_nchw.shape[1], w_kn.shape[1]
    w = w_kn.float().reshape(3, 3, cin, cout).permute(3, 2, 0, 1)     # k = (ky*3 + kx)*Cin + ci
    return _store(out, F.conv2d(x_nchw.float(), w, bias.float(), padding=1).permute(0, 2, 3, 1))


def conv_out(x, w_krsc, bias, out_nchw):
    w = w_krsc.float().permute(0, 3, 1, 2)
    return _store(out_nchw, F.conv2d(x.float().permute(0, 3, 1, 2), w, bias.float(), padding=1))


def pad_cast(x, out_padded):
    out_padded[:, 1:-1, 1:-1] = x.to(out_padded.dtype)
    return out_padded


def zero_stuff(dy, out_padded):
    inner = out_padded[:, 1:-1, 1:-1]
    inner.zero_()
    inner[:, ::2, ::2] = dy.to(out_padded.dtype)
    return out_padded


def sum2x2(du, dx, accumulate=False):
    B, H, W, C = dx.shape
    s = du.float().reshape(B, H, 2, W, 2, C).sum(dim=(2, 4))
    return _store(dx, dx + s if accumulate else s)


def transpose(src, dst):
    return _store(dst, src.t())


def transpose_batched(src, dst):
    return _store(dst, src.transpose(1, 2))


def copy_rows(dst, src):
    return _store(dst, src)


def timestep_embed(t, freqs, out, flip_sin_to_cos):
    e = t.float()[:, None] * freqs.float()[None]
    parts = [torch.cos(e), torch.sin(e)] if flip_sin_to_cos else [torch.sin(e), torch.cos(e)]
    return _store(out, torch.cat(parts, -1))


def linear_rows(x, w, bias, out, act_in=False, act_out=False):
    v = F.linear(F.silu(x.float()) if act_in else x.float(), w.float(), None if bias is None else bias.float())
    return _store(out, F.silu(v) if act_out else v)


def mse_grad(pred, noise, mask, d_pred, loss):
    keep = 1.0 - mask
    d = (pred - noise) * keep
    loss[0] = (d * d).mean()
    return _store(d_pred, 2.0 * d * keep / d.numel())


def install(monkeypatch):
    """Replace every entry point the training composition uses on the real storygen_amd.ops module."""
    from storygen_amd import ops
    for name, fn in list(globals().items()):
        if callable(fn) and not name.startswith("_") and name not in ("install",) and hasattr(ops, name):
            monkeypatch.setattr(ops, name, fn)
    assert math.isclose(LOG2E, 1.0 / math.log(2.0))
