#!/usr/bin/env python
"""Thread-level CPU emulation of the index arithmetic of the bandwidth-bound backward kernels (csrc/backward.hip and the
GroupNorm backward in csrc/norm.hip), which were written without access to a GPU: every kernel's (block, thread) -> element
mapping and reduction tree is transliterated in numpy and compared with the closed-form result / the oracle formulas.
Companion of tools/emulate_attention_bwd.py.      python tools/emulate_backward_kernels.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import storygen_backward as Bk  # noqa: E402

rng = np.random.default_rng(0)


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def groupnorm_bwd(B, HW, C, G, silu, eps=1e-5, NT=1024):
    """gn_stats_wide_kernel -> gn_bwd_sums_kernel -> gn_bwd_apply_kernel: chunk geometry, the (low group, high group) split of
    an 8-channel vector, the fixed-order reductions, the pivot-shifted statistics."""
    cpg, vpr = C // G, C // 8
    rpp = NT // vpr
    x = (rng.standard_normal((B, HW, C)) * 2 + 1.5).astype(np.float32)
    dy = rng.standard_normal((B, HW, C)).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C)).astype(np.float32)
    want = min(64, -(-320 // B))
    rpc = max(-(-HW // want), rpp)
    nch = -(-HW // rpc)

    def chunk_partials(b, ch, fn):
        p0, p1 = ch * rpc, min(HW, ch * rpc + rpc)
        part = np.zeros((NT, 4))
        for t in range(NT):
            my_row = t // vpr
            cv = t - my_row * vpr
            if my_row >= rpp:
                continue
            c0 = cv * 8
            nlo = min(8, (c0 // cpg + 1) * cpg - c0)
            a, q = fn(b, list(range(p0 + my_row, p1, rpp)), c0)
            part[t] = [a[:nlo].sum(), q[:nlo].sum(), a[nlo:].sum(), q[nlo:].sum()]
        col = np.zeros((vpr, 4))
        for t in range(vpr):
            for r in range(rpp):
                col[t] += part[r * vpr + t]
        out = np.zeros((G, 2))
        for g in range(G):
            for c in range((g * cpg) >> 3, (((g + 1) * cpg - 1) >> 3) + 1):
                lo, hi = (c * 8) // cpg, (c * 8 + 7) // cpg
                if lo == g:
                    out[g] += col[c, :2]
                if hi == g and hi != lo:
                    out[g] += col[c, 2:]
        return out

    def f_stats(b, rows, c0):
        piv = np.array([x[b, 0, ((c0 + j) // cpg) * cpg] for j in range(8)])
        d = x[b, rows, c0:c0 + 8] - piv if rows else np.zeros((0, 8))
        return d.sum(0), (d * d).sum(0)

    ws = np.array([[chunk_partials(b, ch, f_stats) for ch in range(nch)] for b in range(B)])
    n = HW * cpg
    tot = ws.sum(1)
    md = tot[..., 0] / n
    mean = x[:, 0, ::cpg] + md
    rstd = 1 / np.sqrt(np.maximum(tot[..., 1] / n - md * md, 0) + eps)

    def g_xhat(b, rows, c0):
        grp = np.array([(c0 + j) // cpg for j in range(8)])
        xh = (x[b, rows, c0:c0 + 8] - mean[b, grp]) * rstd[b, grp]
        d = dy[b, rows, c0:c0 + 8].copy()
        if silu:
            nn = xh * gamma[c0:c0 + 8] + beta[c0:c0 + 8]
            sg = 1 / (1 + np.exp(-nn))
            d = d * sg * (1 + nn * (1 - sg))
        return d * gamma[c0:c0 + 8], xh

    def f_sums(b, rows, c0):
        g, xh = g_xhat(b, rows, c0)
        return g.sum(0), (g * xh).sum(0)

    ws2 = np.array([[chunk_partials(b, ch, f_sums) for ch in range(nch)] for b in range(B)])
    t2 = ws2.sum(1)
    m1, m2 = t2[..., 0] / n, t2[..., 1] / n
    dx = np.zeros_like(x)
    for b in range(B):
        for c0 in range(0, C, 8):
            g, xh = g_xhat(b, list(range(HW)), c0)
            grp = np.array([(c0 + j) // cpg for j in range(8)])
            dx[b, :, c0:c0 + 8] = rstd[b, grp] * (g - m1[b, grp] - xh * m2[b, grp])
    xi = torch.tensor(x).transpose(1, 2).reshape(B, C, HW, 1)
    dyi = torch.tensor(dy).transpose(1, 2).reshape(B, C, HW, 1)
    if silu:
        dyi = Bk.silu_bwd(F.group_norm(xi, G, torch.tensor(gamma), torch.tensor(beta), eps), dyi)
    ref = Bk.group_norm_bwd(xi, torch.tensor(gamma), dyi, G, eps).reshape(B, C, HW).transpose(1, 2).numpy()
    return rel(dx, ref)


def layernorm_bwd(M, C, dual):
    """layernorm_bwd_kernel<NV>: one wave per row, lane l holds the 8-channel vectors l, l + 64, ...; wave sums."""
    NV = -(-(C // 8) // 64)
    x, dy1, dy2 = (rng.standard_normal((M, C)) for _ in range(3))
    g1, g2, res = 1 + 0.1 * rng.standard_normal(C), 1 + 0.1 * rng.standard_normal(C), rng.standard_normal((M, C))
    out = np.zeros((M, C))
    vpr = C // 8
    for row in range(M):
        xs = np.zeros((64, NV, 8)); gs = np.zeros((64, NV, 8))
        for lane in range(64):
            for i in range(NV):
                cv = lane + 64 * i
                if cv < vpr:
                    xs[lane, i] = x[row, cv * 8:cv * 8 + 8]
                    gs[lane, i] = dy1[row, cv * 8:cv * 8 + 8] * g1[cv * 8:cv * 8 + 8]
                    if dual:
                        gs[lane, i] += dy2[row, cv * 8:cv * 8 + 8] * g2[cv * 8:cv * 8 + 8]
        mask = np.array([[lane + 64 * i < vpr for i in range(NV)] for lane in range(64)])[:, :, None]
        mean = (xs * mask).sum() / C
        rstd = 1 / np.sqrt((((xs - mean) ** 2) * mask).sum() / C + 1e-5)
        xh = (xs - mean) * rstd
        m1, m2 = (gs * mask).sum() / C, (gs * xh * mask).sum() / C
        for lane in range(64):
            for i in range(NV):
                cv = lane + 64 * i
                if cv < vpr:
                    out[row, cv * 8:cv * 8 + 8] = rstd * (gs[lane, i] - m1 - xh[lane, i] * m2) + 2.0 * res[row, cv * 8:cv * 8 + 8]
    t = lambda a: torch.tensor(a)      # noqa: E731
    ref = 2.0 * res + Bk.layer_norm_bwd(t(x), t(g1), t(dy1)).numpy() + (Bk.layer_norm_bwd(t(x), t(g2), t(dy2)).numpy() if dual else 0)
    return rel(out, ref)


def transpose(M, C):
    """transpose_kernel: 64 x 64 tiles, thread t loads rows t/8 + 32 i chunk t%8, stores output rows t/8 + 32 i chunk t%8."""
    src = rng.standard_normal((M, C))
    dst = np.full((C, M), np.nan)
    for bx in range(-(-M // 64)):
        for by in range(-(-C // 64)):
            m0, c0 = bx * 64, by * 64
            tile = np.zeros((64, 72))
            for t in range(256):
                ch = t & 7
                for i in range(2):
                    r = (t >> 3) + 32 * i
                    if m0 + r < M and c0 + ch * 8 < C:
                        tile[r, ch * 8:ch * 8 + 8] = src[m0 + r, c0 + ch * 8:c0 + ch * 8 + 8]
            for t in range(256):
                ch = t & 7
                for i in range(2):
                    c = (t >> 3) + 32 * i
                    if c0 + c < C and m0 + ch * 8 < M:
                        dst[c0 + c, m0 + ch * 8:m0 + ch * 8 + 8] = tile[ch * 8:ch * 8 + 8, c]
    return float(np.abs(dst - src.T).max())


def geglu_bwd(M, N8):
    proj, du = rng.standard_normal((M, N8)), rng.standard_normal((M, N8 // 2))
    dproj = np.zeros((M, N8))
    och = N8 // 16
    for idx in range(M * och):
        m, j = divmod(idx, och)
        vcol = (j >> 2) * 64 + (j & 3) * 8
        v, g, d = proj[m, vcol:vcol + 8], proj[m, vcol + 32:vcol + 40], du[m, j * 8:j * 8 + 8]
        gt = torch.tensor(g, requires_grad=True)
        gl = F.gelu(gt)
        gl.sum().backward()
        dproj[m, vcol:vcol + 8] = d * gl.detach().numpy()
        dproj[m, vcol + 32:vcol + 40] = d * v * gt.grad.numpy()
    pr = proj.reshape(M, N8 // 64, 2, 32)
    val, gate = pr[:, :, 0].reshape(M, -1), pr[:, :, 1].reshape(M, -1)
    gt = torch.tensor(gate, requires_grad=True)
    (torch.tensor(val) * F.gelu(gt)).backward(torch.tensor(du))
    ref = np.stack([(du * F.gelu(torch.tensor(gate)).numpy()).reshape(M, -1, 32), gt.grad.numpy().reshape(M, -1, 32)], 2).reshape(M, N8)
    return float(np.abs(dproj - ref).max())


def zero_stuff_and_sum2x2():
    B, Ho, Wo, C = 2, 3, 4, 8
    dy = rng.standard_normal((B, Ho, Wo, C))
    H, W = 2 * Ho, 2 * Wo
    y = np.zeros((B, H + 2, W + 2, C))
    for pix in range(B * H * W):
        xx, yy, b = pix % W, (pix // W) % H, pix // (W * H)
        y[b, yy + 1, xx + 1] = dy[b, yy >> 1, xx >> 1] if ((xx | yy) & 1) == 0 else 0
    z = np.zeros((B, H, W, C))
    z[:, ::2, ::2] = dy
    e1 = float(np.abs(y[:, 1:-1, 1:-1] - z).max() + np.abs(y[:, 0]).max() + np.abs(y[:, :, 0]).max())
    du = rng.standard_normal((B, 2 * Ho, 2 * Wo, C))
    dx = np.zeros((B, Ho, Wo, C))
    flat = du.reshape(-1, C)
    for pix in range(B * Ho * Wo):
        xx, yy, b = pix % Wo, (pix // Wo) % Ho, pix // (Wo * Ho)
        r0 = (b * 2 * Ho + 2 * yy) * (2 * Wo) + 2 * xx
        dx[b, yy, xx] = sum(flat[r0 + (q >> 1) * (2 * Wo) + (q & 1)] for q in range(4))
    e2 = float(np.abs(dx - du.reshape(B, Ho, 2, Wo, 2, C).sum((2, 4))).max())
    return e1, e2


if __name__ == "__main__":
    for args in ((2, 50, 320, 32, True), (1, 37, 960, 32, False), (1, 16, 2560, 32, True), (3, 30, 640, 32, True)):
        e = groupnorm_bwd(*args)
        print("groupnorm_bwd", args, f"{e:.1e}")
        assert e < 1e-5
    for args in ((5, 320, True), (3, 640, False), (2, 1280, True), (2, 2048, True)):
        e = layernorm_bwd(*args)
        print("layernorm_bwd", args, f"{e:.1e}")
        assert e < 1e-10
    for args in ((128, 64), (200, 72), (64, 1280), (72, 40)):
        e = transpose(*args)
        print("transpose", args, e)
        assert e == 0.0
    e = geglu_bwd(3, 192)
    print("geglu_bwd", f"{e:.1e}")
    assert e < 1e-12
    e1, e2 = zero_stuff_and_sum2x2()
    print("zero_stuff", e1, "sum2x2", e2)
    assert e1 == 0.0 and e2 < 1e-12
    print("EMULATION_OK")
