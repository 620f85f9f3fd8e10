"""Weight layouts the kernels consume, derived from the PyTorch-layout checkpoint tensors (the
state dict keeps PyTorch layouts; these are cached *derived* copies so `state_dict()` round-trips)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def conv3x3_krsc(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [Cout, 3, 3, Cin] (K index = (ky*3 + kx)*Cin + ci, matching the NHWC gather)."""
    return w.permute(0, 2, 3, 1).contiguous()


def conv_in_kn(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [9*Cin, Cout] with k = (ky*3 + kx)*Cin + ci."""
    cout, cin = w.shape[:2]
    return w.permute(2, 3, 1, 0).reshape(9 * cin, cout).contiguous()


def conv1x1_nk(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 1, 1] -> [Cout, Cin]."""
    return w.reshape(w.shape[0], w.shape[1]).contiguous()


def interleave_geglu(w: torch.Tensor, b: Optional[torch.Tensor]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """GEGLU.proj weight [2*inner, K] = [value | gate] (model/attention.py:391-392 chunks the output in halves)
    -> rows interleaved in groups of 32 so one 64-column MFMA tile holds value j and gate j side by side:
    rows [64u, 64u+32) = value[32u : 32u+32], rows [64u+32, 64u+64) = gate[32u : 32u+32]."""
    inner = w.shape[0] // 2
    assert inner % 32 == 0
    wi = torch.stack([w[:inner].reshape(inner // 32, 32, -1), w[inner:].reshape(inner // 32, 32, -1)], dim=1)
    wi = wi.reshape(2 * inner, -1).contiguous()
    bi = None
    if b is not None:
        bi = torch.stack([b[:inner].reshape(inner // 32, 32), b[inner:].reshape(inner // 32, 32)], dim=1).reshape(-1).contiguous()
    return wi, bi


def fold_layernorm(w: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor
                   ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Weights of `LayerNorm(gamma, beta)` followed by `Linear(w [N,K], b)` for evaluation as ONE GEMM on the raw input plus a rank-1
    epilogue (tools/next_round/README.md item 1; not used by the engine yet):

        LN(x) w^T + b  =  rstd * (x w'^T)  -  rstd * mu * c  +  d        mu, rstd = row statistics of x

    Returns (w' = gamma (.) w rounded to fp16, c [N] fp32 = row sums of the ROUNDED w' — the correction must cancel what the MFMA
    actually multiplies —, d [N] fp32 = w beta + b)."""
    w32, g32, be32 = w.float(), gamma.float(), beta.float()
    wf = (w32 * g32[None]).to(torch.float16)
    c = wf.float().sum(dim=1)
    d = w32 @ be32 + (0.0 if b is None else b.float())
    return wf, c, d


# ---- fused GEGLU feed-forward (csrc/ff_fused.hip, include/storygen_hip.h: sg_ff_geglu_fused_f16) --------------------------------------
FF_CHUNK_HIDDEN = 32                     # hidden units per chunk = one 32-row MFMA tile of values + one of gates


def _pi32(i: int) -> int:
    """The row permutation of the value / gate tiles: swap bits 2 and 3 of the MFMA row.  With it, accumulator registers 8 ks .. 8 ks + 7
    of a lane are the 8 CONSECUTIVE hidden units 16 ks + 8 hi .. + 7 — the B-operand fragment of the second GEMM's k-step ks."""
    return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1)


def ff_fused_layout(C: int) -> dict:
    """Sizes of the packed weight stream of ff_fused_pack (all in bytes)."""
    assert C % 64 == 0 and (C * 64) % 1024 == 0, "the fused feed-forward kernel is built for C = 320"
    w1_img = (C // 64) * 64 * 128        # C/64 K slabs, each 64 rows (32 values + 32 gates) x 64 halves
    w1_part = w1_img + 64 * 32           # + one more 16-deep k-step: the chunk's d1 terms as fp16 (hi, lo) pairs (whole KiB: 2)
    return dict(C=C, chunks=4 * C // FF_CHUNK_HIDDEN, w1_img=w1_img, w1_part=w1_part, w2_part=C * 64, chunk=w1_part + C * 64)


def ff_fused_pack(w1f: torch.Tensor, d1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """The weight stream of the fused GEGLU feed-forward kernel: byte-for-byte the LDS images its workgroups consume, so that
    filling a ring slot is a linear LDS-DMA copy.

    w1f [8C, C] fp16 = gamma (.) W1 in the 32/32-interleaved GEGLU row order (interleave_geglu + fold_layernorm), d1 [8C] fp32 = W1 beta +
    b1 in the same order, w2 [C, 4C] fp16 (ff.net.2.weight).  Per chunk c of 32 hidden units (interleaved rows [64c, 64c + 64)):
      W1 part: for every 64-deep K slab, a [64 rows][64 halves] image — row t*32 + i holds interleaved row 64c + t*32 + pi(i) (t = 0
               values, 1 gates; pi = _pi32), 16-byte chunk j of row r at byte r*128 + ((j ^ ((r >> 1) & 7)) << 4) — then ONE MORE
               16-deep k-step, [64 rows][16 halves] (32 B per row, same row order): (hi, lo, 0, ..., 0) with hi = fp16(d1), lo = fp16(d1 -
               hi) of the row's d1 term.  The kernel multiplies it with the constant activation fragment (1, 1, 0, ...): the first MFMA
               of a chunk delivers d1 (to ~2^-22 relative) into the accumulators;
      W2 part: [C rows = output columns][32 halves = the chunk's hidden units] of W2 / 2 (the kernel evaluates 2 gelu(g); halving is
               exact in fp16 down to the subnormals), chunk j of row n at byte n*64 + ((j ^ ((n >> 2) & 3)) << 4).
    Returns a uint8 tensor [chunks * (w1_part + w2_part)] on w1f's device."""
    C = w2.shape[0]
    L = ff_fused_layout(C)
    assert tuple(w1f.shape) == (8 * C, C) and tuple(w2.shape) == (C, 4 * C) and d1.numel() == 8 * C
    dev = w1f.device
    nch, ks = L["chunks"], C // 64
    w1f = w1f.to(torch.float16).contiguous()
    w2 = (w2.to(torch.float16) * 0.5).contiguous()
    pi = torch.tensor([t * 32 + _pi32(i) for t in range(2) for i in range(32)], device=dev)
    # W1: [chunk][slab][row][slot(8)][8 halves]; slot s of row r holds logical chunk s ^ ((r >> 1) & 7)
    a = w1f.view(nch, 64, ks, 8, 8)[:, pi]                                   # [chunk, row, slab, logical chunk, 8]
    r = torch.arange(64, device=dev)
    logical = torch.arange(8, device=dev)[None, :] ^ ((r[:, None] >> 1) & 7)    # [row, slot] -> logical chunk
    a = a.permute(0, 2, 1, 3, 4)                                             # [chunk, slab, row, logical, 8]
    a = torch.gather(a, 3, logical[None, None, :, :, None].expand(nch, ks, 64, 8, 8))
    w1_img = a.reshape(nch, L["w1_img"] // 2)
    # W2: [chunk][row n][slot(4)][8 halves]
    b = w2.view(C, nch, 4, 8).permute(1, 0, 2, 3)                            # [chunk, n, logical chunk, 8]
    n = torch.arange(C, device=dev)
    logical2 = torch.arange(4, device=dev)[None, :] ^ ((n[:, None] >> 2) & 3)
    b = torch.gather(b, 2, logical2[None, :, :, None].expand(nch, C, 4, 8))
    w2_img = b.reshape(nch, C * 32)
    out = torch.zeros(nch, L["chunk"], dtype=torch.uint8, device=dev)
    out[:, : L["w1_img"]] = w1_img.contiguous().view(torch.uint8).view(nch, -1)
    dd = d1.to(device=dev, dtype=torch.float32).contiguous().view(nch, 64)[:, pi]            # image row order
    hi = dd.to(torch.float16)
    lo = (dd - hi.float()).to(torch.float16)
    dimg = torch.zeros(nch, 64, 16, dtype=torch.float16, device=dev)
    dimg[:, :, 0], dimg[:, :, 1] = hi, lo
    out[:, L["w1_img"]: L["w1_part"]] = dimg.view(torch.uint8).view(nch, -1)
    out[:, L["w1_part"]:] = w2_img.contiguous().view(torch.uint8).view(nch, -1)
    return out.reshape(-1)
