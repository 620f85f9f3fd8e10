"""TEST INFRASTRUCTURE — torch/CPU emulation of the storygen_amd.ops entry points the encoder engines call, so that the HOST logic of
storygen_amd/encoders.py (weight folds, layouts, buffer reuse, the shifted-view trick of the asymmetric stride-2 convolution) can be
checked against the oracle without a GPU.  `patched_ops()` swaps the functions in for the duration of a `with` block; nothing in the
product imports this file, and the emulation is never a fallback: the GPU tests run the real kernels through the same engine code."""
import contextlib

import torch
import torch.nn.functional as F

from storygen_amd import ops


def _store(out: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
    out.copy_(val.to(out.dtype))
    return out


def gemm(a, w, out, *, bias=None, rowbias=None, rows_per_batch=1, res1=None, res2=None, epilogue=0, split_k=0, workspace=None, out2=None,
         tile=None, use_table=True, stats=None):
    assert epilogue == 0 and rowbias is None and stats is None
    assert a.dtype == torch.float16 and w.dtype == torch.float16 and a.shape[1] == w.shape[1]
    assert a.shape[1] % 8 == 0 and w.shape[0] % 8 == 0, "sg_gemm_f16: K and N must be multiples of 8"
    v = a.float() @ w.float().t()
    if bias is not None:
        v = v + bias.float()[None]
    if res1 is not None:
        v = v + res1.float()
    if res2 is not None:
        v = v + res2.float()
    assert tuple(out.shape) == tuple(v.shape)
    return _store(out, v)


def conv3x3(x, w_krsc, out, *, stride=1, upsample2x=False, bias=None, rowbias=None, res1=None, split_k=0, workspace=None, x_padded=False,
            tile=None, stats=None):
    assert x.dtype == torch.float16 and w_krsc.dtype == torch.float16 and rowbias is None and stats is None
    assert x.shape[3] % 64 == 0 and w_krsc.shape[0] % 8 == 0
    B, Ho, Wo, Co = out.shape
    wt = w_krsc.float().permute(0, 3, 1, 2)                       # [Co, Ci, 3, 3]
    xf = x.float().permute(0, 3, 1, 2)                            # [B, Ci, H(+2), W(+2)]
    if upsample2x:
        assert x_padded and stride == 1
        v = F.conv2d(F.interpolate(xf[:, :, 1:-1, 1:-1], scale_factor=2.0, mode="nearest"), wt, padding=1)
    elif x_padded:
        # the kernel addresses padded pixel (oy*stride + ky, ox*stride + kx) of the tensor it is handed
        need_h, need_w = (Ho - 1) * stride + 3, (Wo - 1) * stride + 3
        assert need_h <= xf.shape[2] and need_w <= xf.shape[3]
        v = F.conv2d(xf[:, :, :need_h, :need_w], wt, stride=stride, padding=0)
    else:
        v = F.conv2d(xf, wt, stride=stride, padding=1)
    v = v.permute(0, 2, 3, 1)
    if bias is not None:
        v = v + bias.float()
    if res1 is not None:
        v = v + res1.float()
    assert tuple(v.shape) == tuple(out.shape), (v.shape, out.shape)
    return _store(out, v)


def conv_in(x_nchw, w_kn, bias, out):
    B, Ci, H, W = x_nchw.shape
    Co = w_kn.shape[1]
    assert Ci <= 8 and Co % 8 == 0 and tuple(w_kn.shape) == (9 * Ci, Co)
    wt = w_kn.float().view(3, 3, Ci, Co).permute(3, 2, 0, 1)
    v = F.conv2d(x_nchw.float(), wt, bias.float(), padding=1).permute(0, 2, 3, 1)
    return _store(out, v)


def conv_out(x, w_krsc, bias, out_nchw):
    Co = w_krsc.shape[0]
    assert Co <= 4 and x.shape[3] % 8 == 0
    v = F.conv2d(x.float().permute(0, 3, 1, 2), w_krsc.float().permute(0, 3, 1, 2), bias.float()[:Co], padding=1)
    return _store(out_nchw, v)


def groupnorm(x, gamma, beta, out, groups, eps, silu, workspace, xcopy=None, pstats=None):
    B, HW, Cc = x.shape
    v = F.group_norm(x.float().transpose(1, 2), groups, gamma.float(), beta.float(), eps).transpose(1, 2)
    if silu:
        v = F.silu(v)
    if xcopy is not None:
        _store(xcopy, x.float())
    if out.dim() == 4:
        Hp, Wp = out.shape[1], out.shape[2]
        out[:, 1:-1, 1:-1, :] = v.reshape(B, Hp - 2, Wp - 2, Cc).to(out.dtype)   # interior only, like the kernel
        return out
    return _store(out, v)


def layernorm(x, g1, b1, y1, eps=1e-5, g2=None, b2=None, y2=None):
    _store(y1, F.layer_norm(x.float(), (x.shape[1],), g1.float(), b1.float(), eps))


def pad_cast(x, out_padded):
    out_padded[:, 1:-1, 1:-1, :] = x.to(out_padded.dtype)
    return out_padded


def copy_rows(dst, src):
    return _store(dst, src)


def softmax_rows(scores, probs, scale=1.0):
    N = scores.shape[1]
    probs.zero_()
    probs[:, :N] = torch.softmax(scores.float() * scale, dim=-1).to(probs.dtype)
    return probs


def attention_small(q, k, v, out, heads, scale, causal, key_bias=None):
    B, T, Cq = q.shape
    D = Cq // heads
    assert T <= 128 and D <= 64
    qh, kh, vh = (t.float().view(B, T, heads, D).transpose(1, 2) for t in (q, k, v))
    s = (qh * scale) @ kh.transpose(-1, -2)
    if causal:
        s = s + torch.full((T, T), float("-inf")).triu(1)
    if key_bias is not None:
        s = s + key_bias[:, None, None, :]
    return _store(out, (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, Cq))


def act_rows(x, act):
    xf = x.float()
    return _store(x, xf * torch.sigmoid(1.702 * xf) if act == ops.ACT_QUICK_GELU else F.gelu(xf))


def embed_tokens(ids, tok, pos, out, T):
    r = torch.arange(ids.numel())
    return _store(out, tok[ids] + pos[r % T])


def gaussian_sample(mean, logvar, noise, out, scale=1.0):
    v = mean if noise is None else mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
    return _store(out, v * scale)


_EMULATED = dict(gemm=gemm, conv3x3=conv3x3, conv_in=conv_in, conv_out=conv_out, groupnorm=groupnorm, layernorm=layernorm, pad_cast=pad_cast,
                 copy_rows=copy_rows, softmax_rows=softmax_rows, attention_small=attention_small, act_rows=act_rows, embed_tokens=embed_tokens,
                 gaussian_sample=gaussian_sample)


@contextlib.contextmanager
def patched_ops():
    saved = {k: getattr(ops, k) for k in _EMULATED}
    try:
        for k, fn in _EMULATED.items():
            setattr(ops, k, fn)
        yield
    finally:
        for k, fn in saved.items():
            setattr(ops, k, fn)
