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
