"""Tensor-level wrappers over the C ABI (include/storygen_hip.h).

Each wrapper borrows `data_ptr()`s of caller-owned torch tensors for the duration of the enqueue and launches on
torch's *current* HIP stream, so PyTorch's caching allocator keeps stream ordering and `torch.cuda.graph` capture
just works.  Nothing here computes: no op has a torch fallback, and importing this module without the built
library raises (storygen_amd/_lib.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import AttnDesc, ConvDesc, FfDesc, GemmDesc, GroupNormDesc, check

lib = _lib.load()

EPI_LINEAR = 0
EPI_GEGLU = 1
F_OUT_F32, F_RES1_F32, F_RES2_F32 = 1, 2, 4

# Per-shape tile / split-K choices measured on an MI355X by tools/tune_tiles.py (storygen_amd/tuning/mi355x_tiles.json);
# anything not listed uses the library's cost model.  TUNE_SINK: when set, every gemm/conv3x3 call appends its
# signature and a recipe to rebuild it (the tuning tool's shape census).
TILE_TABLE = {}
TUNE_SINK = None


def _apply_tile(d, tile, split_k, sig, mn):
    """tile = (bm, bn) or (bm, bn, waves) from the caller, else the tuned table entry (bm, bn, split[, waves]).  mn = M * N of the launch:
    a tuned split-K count is taken only when the CALLER's workspace holds its split * M * N fp32 partials — the table is keyed by shape,
    and a caller of the same shape without split-K scratch (the training blocks) keeps the library's own choice for its workspace."""
    if tile is not None:
        d.tile_m, d.tile_n = tile[0], tile[1]
        if len(tile) > 2:
            d.tile_waves = tile[2]
    elif split_k == 0 and sig in TILE_TABLE:
        e = TILE_TABLE[sig]
        d.tile_m, d.tile_n = e[0], e[1]
        d.split_k = e[2] if e[2] <= 1 or d.workspace_bytes >= e[2] * mn * 4 else 0
        if len(e) > 3:
            d.tile_waves = e[3]


def load_tile_table(path: Optional[str] = None) -> int:
    """Loads the per-shape tile table (default: tuning/mi355x_tiles.json); path="none" empties it (heuristic tiles only)."""
    import json
    import os
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "mi355x_tiles.json")
    TILE_TABLE.clear()
    if os.path.exists(path) and path != "none":
        with open(path) as f:
            TILE_TABLE.update({k: tuple(v) for k, v in json.load(f)["tiles"].items()})
    return len(TILE_TABLE)


# Optional in-situ kernel timer (bench.py): when set, every MFMA-class launch is bracketed by HIP events recorded
# on the launch stream and reported as (family, algorithmic_flops, start_event, end_event).
PROFILE_SINK = None
LN_GUARD_RANGE, LN_GUARD_OFFSET, LN_GUARD_RATIO = 1, 2, 16.0      # include/storygen_hip.h SG_LN_GUARD_*
# Measurement tooling (bench.py --dump-algorithmic): when a list, every GEMM / convolution launch reports the profiler class it
# will fall into and its ALGORITHMIC bytes: (kernel instantiation as rocprofv3 prints it, grid size in threads, bytes, family, shape)
PLAN_SINK = None
# Same for the bandwidth-bound kernels (tools/profile_step.py): (family, algorithmic_bytes, start, end, shape).
AUX_SINK = None


class _timed:
    __slots__ = ("family", "flops", "start", "shape", "aux")

    def __init__(self, family: str, flops: float, shape: str = "", aux: bool = False):
        self.family, self.flops, self.shape, self.aux = family, flops, shape, aux

    def __enter__(self):
        if (AUX_SINK if self.aux else PROFILE_SINK) is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        sink = AUX_SINK if self.aux else PROFILE_SINK
        if sink is not None and exc[0] is None:
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream())
            sink.append((self.family, self.flops, self.start, end, self.shape))
        return False


load_tile_table()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f16(t: torch.Tensor, name: str):
    if t.dtype != torch.float16 or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA/HIP float16 tensor, got {t.dtype} on {t.device}")


def _act(t: torch.Tensor, name: str) -> bool:
    """Activation tensors may be fp16 (MFMA operands) or fp32 (residual stream); returns True for fp32."""
    if t.dtype not in (torch.float16, torch.float32) or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA/HIP float16/float32 tensor, got {t.dtype} on {t.device}")
    return t.dtype == torch.float32


def _f32(t: torch.Tensor, name: str):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA/HIP float32 tensor, got {t.dtype} on {t.device}")


def _row_stride(t: torch.Tensor, name: str) -> int:
    """t is [..., rows, cols] with unit column stride; returns the row stride in elements."""
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t.stride(-2)


def device_arch() -> int:
    return lib.sg_device_arch()


def gemm_workspace_bytes(M: int, N: int, split_k: int = 0) -> int:
    return lib.sg_gemm_workspace_bytes(M, N, split_k)


# Development: when set to an int64 CUDA tensor, gemm() / conv3x3() launch the instrumented mainloop (sg_debug_*_anatomy) and
# the tensor receives 10 cycle counters per wave (tools/anatomy.py).
ANATOMY: Optional[torch.Tensor] = None


def new_workspace(nbytes: int, device) -> torch.Tensor:
    """A split-K workspace for gemm / conv3x3 (fp32 partial tiles; no initialisation needed)."""
    return torch.empty((int(nbytes) + 15) & ~15, dtype=torch.uint8, device=device)


# Split-K scratch of the gemm / conv3x3 calls that do not pass `workspace=` themselves (round 6).  The library splits K only when the caller
# owns scratch for the partial tiles; the training classes (train.py, train_blocks.py: ~50 call sites, ONE stream) had none, so their
# small-M / long-K launches — weight gradients (M = N = 320, K = 12 288 tokens), the 8x8 and 16x16 convolutions at batch 4 — ran 25 - 80
# workgroups down 180-slab chains (110 us where the sampler's batch-3 twin takes 21).  `with ops.default_workspace(ws):` lends them one
# buffer for the duration of a step; only for callers whose launches are ordered on one stream.
DEFAULT_WORKSPACE: Optional[torch.Tensor] = None


class default_workspace:
    def __init__(self, ws: Optional[torch.Tensor]):
        self.ws, self.prev = ws, None

    def __enter__(self):
        global DEFAULT_WORKSPACE
        self.prev, DEFAULT_WORKSPACE = DEFAULT_WORKSPACE, self.ws
        return self.ws

    def __exit__(self, *exc):
        global DEFAULT_WORKSPACE
        DEFAULT_WORKSPACE = self.prev
        return False


def _ws(workspace: Optional[torch.Tensor], d):
    if workspace is None:
        workspace = DEFAULT_WORKSPACE
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()


def _gemm_desc(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
               rowbias: Optional[torch.Tensor] = None, rows_per_batch: int = 1, res1: Optional[torch.Tensor] = None,
               res2: Optional[torch.Tensor] = None, epilogue: int = EPI_LINEAR, split_k: int = 0,
               workspace: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None,
               tile: Optional[tuple] = None, use_table: bool = True, stats: Optional[tuple] = None,
               ln: Optional[tuple] = None, ln_out: Optional[torch.Tensor] = None, guard: Optional[torch.Tensor] = None):
    """Builds the sg_gemm_desc of one problem; returns (desc, flops, shape string).  stats = (fp32 buffer, rows per image): the
    epilogue also writes the GroupNorm partial statistics of the output (sg_gemm_desc.stats).
    ln = (mode, stats [tokens, K/64 rounded up to even, 2] fp32, c fp32, d fp32, eps): LayerNorm folded into this GEMM (sg_gemm_desc.ln_*: mode 1 =
    the rows of `a` are the normalised tokens, 2 = the rows of `w` are); ln_out = fp32 [M, N/64 rounded up to even, 2]: also write the LayerNorm
    partials of THIS output (sg_gemm_desc.ln_stats_out); guard = int32 [1] CUDA tensor receiving the sticky SG_LN_GUARD_* flags of
    a launch with ln / ln_out (LN_GUARD_RANGE: the raw fp16 copy saturated; LN_GUARD_OFFSET: a token with |mean| / sigma > 16)."""
    _f16(a, "a"), _f16(w, "w")
    flags = F_OUT_F32 if _act(out, "out") else 0
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm: a is [{M},{K}] but w is {tuple(w.shape)}")
    n_out = N // 2 if epilogue == EPI_GEGLU else N
    if tuple(out.shape) != (M, n_out):
        raise ValueError(f"gemm: out must be [{M},{n_out}], got {tuple(out.shape)}")
    d = GemmDesc()
    d.A, d.lda = a.data_ptr(), _row_stride(a, "a")
    d.W, d.ldw = w.data_ptr(), _row_stride(w, "w")
    d.C, d.ldc = out.data_ptr(), _row_stride(out, "out")
    d.M, d.N, d.K = M, N, K
    d.epilogue = epilogue
    d.bias = _p(bias)
    if bias is not None:
        _f16(bias, "bias")
    if rowbias is not None:
        _f32(rowbias, "rowbias")
        d.rowbias, d.rowbias_ld = rowbias.data_ptr(), _row_stride(rowbias, "rowbias")
    d.rows_per_batch = rows_per_batch
    d.split_k = split_k
    if res1 is not None:
        flags |= F_RES1_F32 if _act(res1, "res1") else 0
        d.res1, d.ldr1 = res1.data_ptr(), _row_stride(res1, "res1")
    if res2 is not None:
        flags |= F_RES2_F32 if _act(res2, "res2") else 0
        d.res2, d.ldr2 = res2.data_ptr(), _row_stride(res2, "res2")
    if out2 is not None:
        _f16(out2, "out2")
        d.C2, d.ldc2 = out2.data_ptr(), _row_stride(out2, "out2")
    d.flags = flags
    _ws(workspace, d)
    if stats is not None:
        _f32(stats[0], "stats")
        d.stats, d.stats_batch_rows = stats[0].data_ptr(), int(stats[1])
    if ln is not None:
        mode, lst, lc, ld_, eps = ln
        _f32(lst, "ln stats"), _f32(lc, "ln c"), _f32(ld_, "ln d")
        tokens = M if mode == 1 else N
        if tuple(lst.shape) != (tokens, (K // 64 + 1) & ~1, 2) or not lst.is_contiguous() or lc.numel() != (N if mode == 1 else M) or ld_.numel() != lc.numel():
            raise ValueError(f"gemm: ln operands do not match mode {mode}, {tokens} tokens, K = {K}")
        d.ln_mode, d.ln_parts, d.ln_eps = int(mode), K // 64, float(eps)
        d.ln_stats, d.ln_c, d.ln_d = lst.data_ptr(), lc.data_ptr(), ld_.data_ptr()
    if ln_out is not None:
        _f32(ln_out, "ln_out")
        if ln_out.numel() != M * ((N // 64 + 1) & ~1) * 2 or not ln_out.is_contiguous():
            raise ValueError(f"gemm: ln_out must hold [{M}, {(N // 64 + 1) & ~1}, 2] floats (N / 64 blocks rounded up to even)")
        d.ln_stats_out = ln_out.data_ptr()
    if guard is not None:
        if guard.dtype != torch.int32 or not guard.is_cuda or guard.numel() < 1:
            raise TypeError("gemm: guard must be a CUDA int32 tensor")
        d.ln_guard = guard.data_ptr()
    sig = f"g:{M}:{N}:{K}:{epilogue}:{flags}:{int(bias is not None)}{int(rowbias is not None)}{int(res1 is not None)}{int(res2 is not None)}{int(out2 is not None)}"
    if ln is not None:
        sig += f":ln{ln[0]}"
    if ln_out is not None:
        sig += ":lo"
    if use_table or tile is not None:
        _apply_tile(d, tile, split_k, sig, M * N)
    if TUNE_SINK is not None and use_table:
        TUNE_SINK.append((sig, dict(kind="gemm", M=M, N=N, K=K, epilogue=epilogue, out_f32=bool(flags & F_OUT_F32), bias=bias is not None,
                                    rowbias=rowbias is not None, rows_per_batch=rows_per_batch, out2=out2 is not None,
                                    res1=None if res1 is None else str(res1.dtype), res2=None if res2 is None else str(res2.dtype),
                                    ln=0 if ln is None else int(ln[0]), ln_out=ln_out is not None)))
    return d, 2.0 * M * N * K, f"M{M} N{N} K{K}{' geglu' if epilogue else ''}"


def _gemm_bytes(d) -> float:
    """Algorithmic bytes of one GEMM problem: each operand once (A, W fp16; C fp32 / fp16; residuals; the second fp16 output)."""
    n_out = d.N // 2 if d.epilogue == EPI_GEGLU else d.N
    osz = 4 if d.flags & F_OUT_F32 else 2
    b = 2.0 * d.M * d.K + 2.0 * d.N * d.K + osz * d.M * n_out
    if d.res1:
        b += (4 if d.flags & F_RES1_F32 else 2) * d.M * n_out
    if d.res2:
        b += (4 if d.flags & F_RES2_F32 else 2) * d.M * n_out
    if d.C2:
        b += 2.0 * d.M * n_out
    return b


def _plan_of(fn, d):
    out = (C.c_int32 * 6)()
    check(fn(C.byref(d), out), "launch plan")
    return list(out)


def _kernel_name(bm: int, bn: int, pipe: int, conv: bool) -> str:
    """The instantiation a plan launches, as rocprofv3 prints it (pipe: 0 = register-staged, 1 = LDS-DMA ring of 64x64-per-wave tiles,
    16 + S = the 32x32-per-wave kernel on an S-stage ring)."""
    c = "true" if conv else "false"
    if pipe == 2:
        return f"mma_fat_kernel<{bm // 128}, {bn // 64}, {c}>"
    if pipe >= 16:
        return f"mma_lat_kernel<{bm // 32}, {bn // 32}, {c}, {pipe - 16}>"
    return f"mma_pipe_kernel<{bm // 64}, {bn // 64}, {c}, 3>" if pipe else f"mma_kernel<{bm}, {bn}, {c}>"


def _report_plan(kind: str, d, nbytes: float, family: str, shape: str, second=None):
    """PLAN_SINK record of one launch (see PLAN_SINK).  second = (desc, bytes) of the other problem of a paired launch.  Plans are
    queried on COPIES of the descriptors: the instrumented step must launch exactly what the uninstrumented one does."""
    if kind == "conv":
        bm, bn, splits, wgs, threads, pipe = _plan_of(lib.sg_conv3x3_launch_plan, d)
        name = _kernel_name(bm, bn, pipe, True)
    else:
        bm, bn, splits, wgs, threads, pipe = _plan_of(lib.sg_gemm_launch_plan, d)
        name = _kernel_name(bm, bn, pipe, False)
        if second is not None:
            d1, b1 = second
            q = type(d1).from_buffer_copy(d1)
            q.tile_m, q.tile_n, q.tile_waves = bm, bn, ((8 if bn == 128 else 4) if pipe >= 16 else 0)
            bm1, bn1, _, wgs1, threads1, pipe1 = _plan_of(lib.sg_gemm_launch_plan, q)
            if pipe and pipe1 and (bm1, bn1) == (bm, bn) and (pipe >= 16) == (pipe1 >= 16):
                if pipe >= 16:
                    name = f"mma_lat_pair_kernel<{pipe - 16}>"
                else:
                    name = f"mma_pipe_pair_kernel<{bm // 64}, {bn // 64}>"
                wgs, nbytes = ((max(wgs, wgs1) + 7) & ~7) * 2, nbytes + b1
            else:           # not pairable: two plain launches, each on its own plan
                bm1, bn1, _, wgs1, threads1, pipe1 = _plan_of(lib.sg_gemm_launch_plan, d1)
                PLAN_SINK.append((_kernel_name(bm1, bn1, pipe1, False), wgs1 * threads1, b1, family, shape + " [2nd]"))
    PLAN_SINK.append((name, wgs * threads, nbytes, family, shape))


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, **kw) -> torch.Tensor:
    """out[M, N'] = epi(a[M,K] @ w[N,K]^T); a/out/res* may be row-strided 2-D views; out/res* fp16 or fp32;
    out2 = optional extra fp16 copy of the result.  Keywords: bias, rowbias, rows_per_batch, res1, res2, epilogue, split_k,
    workspace, out2, tile."""
    d, flops, shape = _gemm_desc(a, w, out, **kw)
    if PLAN_SINK is not None:
        _report_plan("gemm", d, _gemm_bytes(d), "gemm", shape)
    with _timed("gemm", flops, shape):
        if ANATOMY is not None:
            check(lib.sg_debug_gemm_anatomy(C.byref(d), ANATOMY.data_ptr(), ANATOMY.numel() * ANATOMY.element_size(), _stream()),
                  "sg_debug_gemm_anatomy")
        else:
            check(lib.sg_gemm_f16(C.byref(d), _stream()), "sg_gemm_f16")
    return out


def ff_fused_supported(Cc: int) -> bool:
    """Is there a fused GEGLU feed-forward kernel for this width (sg_ff_fused_pack_bytes != 0)?"""
    return lib.sg_ff_fused_pack_bytes(int(Cc)) != 0


def ff_fused(x: torch.Tensor, wpack: torch.Tensor, b2: torch.Tensor, out: torch.Tensor, eps: float = 1e-5, split: bool = False) -> torch.Tensor:
    """out[M, C] (fp16) = Linear2(a * gelu(g)) + b2 + x with [a | g] = Linear1(LayerNorm(x)) + b1 in ONE launch (sg_ff_geglu_fused_f16):
    x fp32 [M, C] (row-strided view allowed), wpack = repack.ff_fused_pack(...) (uint8), b2 fp16 [C].
    split=True (sg_ff_desc.hidden_split): out is [M, 2C] — two workgroups per 128 tokens, each over half of the hidden units; columns
    [0, C) + columns [C, 2C) = the result (the consumer contracts [y_a | y_b] with [W | W])."""
    _f32(x, "x"), _f16(b2, "b2"), _f16(out, "out")
    M, Cc = x.shape
    if tuple(out.shape) != (M, 2 * Cc if split else Cc) or b2.numel() != Cc or wpack.dtype != torch.uint8 or not wpack.is_cuda or not wpack.is_contiguous():
        raise ValueError("ff_fused: out must be [M, C] fp16 ([M, 2C] with split), b2 [C], wpack a contiguous CUDA uint8 tensor")
    d = FfDesc()
    d.x, d.ldx = x.data_ptr(), _row_stride(x, "x")
    d.wpack, d.wpack_bytes = wpack.data_ptr(), wpack.numel()
    d.b2 = b2.data_ptr()
    d.y, d.ldy = out.data_ptr(), _row_stride(out, "out")
    d.M, d.C, d.eps = M, Cc, float(eps)
    d.hidden_split = 2 if split else 0
    with _timed("ff_fused", 24.0 * M * Cc * Cc, f"M{M} C{Cc}" + (" split" if split else "")):
        if ANATOMY is not None:
            check(lib.sg_debug_ff_anatomy(C.byref(d), ANATOMY.data_ptr(), ANATOMY.numel() * ANATOMY.element_size(), _stream()), "sg_debug_ff_anatomy")
        else:
            check(lib.sg_ff_geglu_fused_f16(C.byref(d), _stream()), "sg_ff_geglu_fused_f16")
    return out


def gemm_stats_rows(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, **kw) -> int:
    """Rows per partial (= the tile height) a gemm() call with these arguments (incl. stats=...) will write statistics for, or 0
    when that launch cannot emit them (sg_gemm_stats_tile_rows; no launch happens)."""
    d, _, _ = _gemm_desc(a, w, out, **kw)
    rc = lib.sg_gemm_stats_tile_rows(C.byref(d))
    if rc < 0:
        check(rc, "sg_gemm_stats_tile_rows")
    return rc


def gemm_pair(first: tuple, second: tuple) -> None:
    """Two independent GEMMs in one launch (sg_gemm_pair_f16).  Each argument is ((a, w, out), {keywords of gemm()}); the
    outputs must not overlap and the two workspaces, if given, must be different buffers."""
    (a0, kw0), (a1, kw1) = first, second
    d0, f0, s0 = _gemm_desc(*a0, use_table=False, **kw0)
    d1, f1, s1 = _gemm_desc(*a1, use_table=False, **kw1)
    if PLAN_SINK is not None:
        _report_plan("gemm", d0, _gemm_bytes(d0), "gemm", f"{s0} + {s1}", second=(d1, _gemm_bytes(d1)))
    with _timed("gemm", f0 + f1, f"{s0} + {s1}"):
        check(lib.sg_gemm_pair_f16(C.byref(d0), C.byref(d1), _stream()), "sg_gemm_pair_f16")


def _conv_desc(x: torch.Tensor, w_krsc: torch.Tensor, out: torch.Tensor, *, stride: int = 1, upsample2x: bool = False,
               bias: Optional[torch.Tensor] = None, rowbias: Optional[torch.Tensor] = None,
               res1: Optional[torch.Tensor] = None, split_k: int = 0, workspace: Optional[torch.Tensor] = None,
               x_padded: bool = False, tile: Optional[tuple] = None, stats: Optional[torch.Tensor] = None,
               defer_reduce: bool = False):
    """Builds the sg_conv3x3_desc; returns (desc, flops, shape string)."""
    _f16(x, "x"), _f16(w_krsc, "w")
    flags = F_OUT_F32 if _act(out, "out") else 0
    B, H, W, Cin = x.shape
    if x_padded:
        H, W = H - 2, W - 2
    Cout = w_krsc.shape[0]
    if tuple(w_krsc.shape) != (Cout, 3, 3, Cin) or not w_krsc.is_contiguous():
        raise ValueError(f"conv3x3: w must be contiguous [Cout,3,3,{Cin}], got {tuple(w_krsc.shape)}")
    hin, win = (H * 2, W * 2) if upsample2x else (H, W)
    Ho, Wo = (hin - 1) // stride + 1, (win - 1) // stride + 1
    if tuple(out.shape) != (B, Ho, Wo, Cout):
        raise ValueError(f"conv3x3: out must be {(B, Ho, Wo, Cout)}, got {tuple(out.shape)}")

    def pix_stride(t, name):
        if t.stride(-1) != 1 or t.stride(1) != t.shape[2] * t.stride(2) or t.stride(0) != t.shape[1] * t.stride(1):
            raise ValueError(f"{name}: must be [B,H,W,C] with a single pixel stride")
        return t.stride(2)

    d = ConvDesc()
    d.x, d.ldx = x.data_ptr(), pix_stride(x, "x")
    d.w = w_krsc.data_ptr()
    d.y, d.ldy = out.data_ptr(), pix_stride(out, "out")
    d.B, d.H, d.W, d.Cin, d.Cout = B, H, W, Cin, Cout
    d.stride, d.upsample2x, d.x_padded = stride, int(upsample2x), int(x_padded)
    if bias is not None:
        _f16(bias, "bias")
        d.bias = bias.data_ptr()
    if rowbias is not None:
        _f32(rowbias, "rowbias")
        d.rowbias, d.rowbias_ld = rowbias.data_ptr(), _row_stride(rowbias, "rowbias")
    if res1 is not None:
        flags |= F_RES1_F32 if _act(res1, "res1") else 0
        d.res1, d.ldr1 = res1.data_ptr(), pix_stride(res1, "res1")
    d.flags = flags
    d.split_k = split_k
    _ws(workspace, d)
    if stats is not None:
        _f32(stats, "stats")
        d.stats = stats.data_ptr()
    d.defer_reduce = int(bool(defer_reduce))
    sig = f"c:{B}:{H}:{W}:{Cin}:{Cout}:{stride}:{int(upsample2x)}:{int(x_padded)}:{flags}:{int(bias is not None)}{int(rowbias is not None)}{int(res1 is not None)}"
    _apply_tile(d, tile, split_k, sig, B * Ho * Wo * Cout)
    if TUNE_SINK is not None:
        TUNE_SINK.append((sig, dict(kind="conv", B=B, H=H, W=W, Cin=Cin, Cout=Cout, stride=stride, ups=bool(upsample2x), padded=bool(x_padded),
                                    out_f32=bool(flags & F_OUT_F32), bias=bias is not None, rowbias=rowbias is not None,
                                    res1=None if res1 is None else str(res1.dtype))))
    return d, 2.0 * B * Ho * Wo * Cout * 9 * Cin, f"B{B} {Ho}x{Wo} {Cin}->{Cout} s{stride}{' up' if upsample2x else ''}"


def conv3x3(x: torch.Tensor, w_krsc: torch.Tensor, out: torch.Tensor, **kw) -> torch.Tensor:
    """x [B,H,W,Cin] (channels-last, pixel-strided view allowed; or the zero-bordered [B,H+2,W+2,Cin] with
    x_padded=True) -> out [B,Ho,Wo,Cout] (fp16 or fp32); w_krsc [Cout,3,3,Cin]; res1 fp16 or fp32.  Keywords: stride, upsample2x,
    bias, rowbias, res1, split_k, workspace, x_padded, tile, stats (fp32 buffer: GroupNorm partial statistics of the output),
    defer_reduce (a split-K launch leaves its partial tiles in the workspace for groupnorm(split=...): sg_conv3x3_desc.defer_reduce)."""
    d, flops, shape = _conv_desc(x, w_krsc, out, **kw)
    if PLAN_SINK is not None:
        Ho, Wo = out.shape[1], out.shape[2]
        nb = 2.0 * d.B * d.H * d.W * d.Cin + 2.0 * d.Cout * 9 * d.Cin + (4 if d.flags & F_OUT_F32 else 2) * d.B * Ho * Wo * d.Cout
        if d.res1:
            nb += (4 if d.flags & F_RES1_F32 else 2) * d.B * Ho * Wo * d.Cout
        _report_plan("conv", d, nb, "conv3x3", shape)
    with _timed("conv3x3", flops, shape):
        if ANATOMY is not None:
            check(lib.sg_debug_conv_anatomy(C.byref(d), ANATOMY.data_ptr(), ANATOMY.numel() * ANATOMY.element_size(), _stream()),
                  "sg_debug_conv_anatomy")
        else:
            check(lib.sg_conv3x3_nhwc_f16(C.byref(d), _stream()), "sg_conv3x3_nhwc_f16")
    return out


def conv3x3_planned_splits(x: torch.Tensor, w_krsc: torch.Tensor, out: torch.Tensor, **kw) -> int:
    """Number of K slices a conv3x3() call with these arguments will use (sg_conv3x3_planned_splits; no launch).  > 1: the call may
    pass defer_reduce=True and hand its workspace to groupnorm(split=...)."""
    d, _, _ = _conv_desc(x, w_krsc, out, **kw)
    rc = lib.sg_conv3x3_planned_splits(C.byref(d))
    if rc < 0:
        check(rc, "sg_conv3x3_planned_splits")
    return rc


def conv3x3_stats_rows(x: torch.Tensor, w_krsc: torch.Tensor, out: torch.Tensor, **kw) -> int:
    """Rows per partial a conv3x3() call with these arguments (incl. stats=...) will write statistics for, or 0 (no launch)."""
    d, _, _ = _conv_desc(x, w_krsc, out, **kw)
    rc = lib.sg_conv3x3_stats_tile_rows(C.byref(d))
    if rc < 0:
        check(rc, "sg_conv3x3_stats_tile_rows")
    return rc


def conv_in(x_nchw: torch.Tensor, w_kn: torch.Tensor, bias: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _f32(x_nchw, "x"), _f16(w_kn, "w"), _f16(bias, "bias")
    y_f32 = _act(out, "out")
    B, Cin, H, W = x_nchw.shape
    Cout = w_kn.shape[1]
    if not x_nchw.is_contiguous() or tuple(w_kn.shape) != (9 * Cin, Cout) or tuple(out.shape) != (B, H, W, Cout):
        raise ValueError("conv_in: bad shapes")
    check(lib.sg_conv_in_f16(x_nchw.data_ptr(), w_kn.data_ptr(), bias.data_ptr(), out.data_ptr(), out.stride(2), int(y_f32),
                             B, H, W, Cin, Cout, _stream()), "sg_conv_in_f16")
    return out


def conv_out(x: torch.Tensor, w_krsc: torch.Tensor, bias: torch.Tensor, out_nchw: torch.Tensor) -> torch.Tensor:
    _f16(x, "x"), _f16(w_krsc, "w"), _f16(bias, "bias"), _f32(out_nchw, "out")
    B, H, W, Cin = x.shape
    Cout = w_krsc.shape[0]
    if tuple(out_nchw.shape) != (B, Cout, H, W) or not out_nchw.is_contiguous() or not w_krsc.is_contiguous():
        raise ValueError("conv_out: bad shapes")
    check(lib.sg_conv_out_f16(x.data_ptr(), x.stride(2), w_krsc.data_ptr(), bias.data_ptr(), out_nchw.data_ptr(), B, H, W,
                              Cin, Cout, _stream()), "sg_conv_out_f16")
    return out_nchw


def _attn_fwd_desc(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, heads: int, scale: float, nk: Optional[int],
                   short: Optional[tuple] = None):
    for n, t in (("q", q), ("k", k), ("vt", vt), ("out", out)):
        _f16(t, n)
        if t.dim() != 3 or t.stride(-1) != 1:
            raise ValueError(f"attention: {n} must be 3-D with a contiguous last dimension")
    B, Nq, Cq = q.shape
    Bk = k.shape[0]
    Nk = k.shape[1] if nk is None else nk
    D = Cq // heads
    if vt.shape[0] != Bk or vt.shape[1] != Cq or k.shape[2] != Cq or Nk > k.shape[1] or vt.shape[2] < ((Nk + 7) & ~7):
        raise ValueError(f"attention: k {tuple(k.shape)} / vt {tuple(vt.shape)} do not match q {tuple(q.shape)}, nk={Nk}")
    d = AttnDesc()
    d.q, d.ldq, d.bsq = q.data_ptr(), q.stride(1), q.stride(0)
    d.k, d.ldk, d.bsk = k.data_ptr(), k.stride(1), k.stride(0)
    d.vt, d.ldvt, d.bsvt = vt.data_ptr(), vt.stride(1), vt.stride(0)
    d.o, d.ldo, d.bso = out.data_ptr(), out.stride(1), out.stride(0)
    d.B, d.H, d.Nq, d.Nk, d.D = B, heads, Nq, Nk, D
    d.kv_batches = Bk
    d.scale = scale
    keys = [Nk] * B
    if short is not None:          # leading K/V rows with their own key count (sg_attn_desc.k2 ...)
        k2, vt2 = short
        _f16(k2, "k2"), _f16(vt2, "vt2")
        n2, Nk2 = k2.shape[0], k2.shape[1]
        if (k2.dim() != 3 or vt2.dim() != 3 or k2.shape[2] != Cq or tuple(vt2.shape[:2]) != (n2, Cq) or vt2.shape[2] < ((Nk2 + 7) & ~7)
                or k2.stride(-1) != 1 or vt2.stride(-1) != 1 or k2.stride(1) != k.stride(1) or vt2.stride(1) != vt.stride(1)):
            raise ValueError("attention: short rows must be [n, Nk2, H*D] / [n, H*D, Nk2] views with the token / row strides of k / vt")
        d.k2, d.bsk2, d.vt2, d.bsvt2 = k2.data_ptr(), k2.stride(0), vt2.data_ptr(), vt2.stride(0)
        d.Nk2, d.kv2_batches = Nk2, n2
        d.kv_batches = Bk = Bk + n2
        keys = [Nk2 if (b if b < Bk else b - (B - Bk)) < n2 else Nk for b in range(B)]
    if d.kv_batches > B:
        raise ValueError(f"attention: {d.kv_batches} K/V rows for {B} query batches")
    shape = f"B{B} H{heads} Nq{Nq} Nk{Nk}" + (f" ({d.kv2_batches}x Nk{d.Nk2})" if short is not None else "")
    return d, 4.0 * heads * Nq * D * float(sum(keys)), shape


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, heads: int, scale: float,
              nk: Optional[int] = None, short: Optional[tuple] = None) -> torch.Tensor:
    """q [B,Nq,H*D], k [Bk,Nk',H*D] (token- and batch-strided views allowed), vt [Bk,H*D,Nk''] = V transposed (keys
    contiguous; rows finite up to nk rounded up to 8), out [B,Nq,H*D].  nk = number of valid keys (default k.shape[1]).
    Bk < B: query batch b uses K/V batch (b if b < Bk else b - (B - Bk)).
    short = (k2 [n, Nk2, H*D], vt2 [n, H*D, Nk2]): n more K/V rows IN FRONT of k's, with Nk2 keys each (all of them valid)."""
    d, flops, shape = _attn_fwd_desc(q, k, vt, out, heads, scale, nk, short)
    with _timed(f"attention_d{d.D}", flops, shape):
        check(lib.sg_attn_fwd_f16(C.byref(d), _stream()), "sg_attn_fwd_f16")
    return out


def attention_pair(a: tuple, b: tuple, heads: int, scale: float) -> None:
    """Two attentions of the same query geometry in one launch (sg_attn_fwd_pair_f16): a, b = (q, k, vt, out, nk) as in attention()
    — the text and the image cross-attention of one transformer block (attention.py:271-276,285-290)."""
    da, fa, sa = _attn_fwd_desc(a[0], a[1], a[2], a[3], heads, scale, a[4])
    db, fb, sb = _attn_fwd_desc(b[0], b[1], b[2], b[3], heads, scale, b[4])
    with _timed(f"attention_d{da.D}", fa + fb, f"{sa} + {sb}"):
        check(lib.sg_attn_fwd_pair_f16(C.byref(da), C.byref(db), _stream()), "sg_attn_fwd_pair_f16")


def attention_f8_bytes(B: int, heads: int, N: int, transposed: bool) -> int:
    return lib.sg_attn_f8_bytes(B, heads, N, int(transposed))


def attention_f8(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, heads: int, scale: float,
                 scratch: torch.Tensor, nk: Optional[int] = None) -> torch.Tensor:
    """`attention` for head dim 40 on the fp8 (e4m3) MFMA path: packs q / k / vt (same operands and layouts as attention()) into
    e4m3 images inside `scratch` (uint8, >= the three attention_f8_bytes) and runs sg_attn_fwd_f8_d40."""
    for n, t in (("q", q), ("k", k), ("vt", vt), ("out", out)):
        _f16(t, n)
        if t.dim() != 3 or t.stride(-1) != 1:
            raise ValueError(f"attention_f8: {n} must be 3-D with a contiguous last dimension")
    B, Nq, Cq = q.shape
    Bk = k.shape[0]
    Nk = k.shape[1] if nk is None else nk
    if Cq != heads * 40:
        raise ValueError("attention_f8: head dim must be 40")
    if vt.shape[0] != Bk or vt.shape[1] != Cq or k.shape[2] != Cq or Nk > k.shape[1] or vt.shape[2] < ((Nk + 7) & ~7):
        raise ValueError(f"attention_f8: k {tuple(k.shape)} / vt {tuple(vt.shape)} do not match q {tuple(q.shape)}, nk={Nk}")
    nq8, nk8, nv8 = attention_f8_bytes(B, heads, Nq, False), attention_f8_bytes(Bk, heads, Nk, False), attention_f8_bytes(Bk, heads, Nk, True)
    if scratch.dtype != torch.uint8 or not scratch.is_cuda or scratch.numel() < nq8 + nk8 + nv8 + 512:
        raise ValueError(f"attention_f8: scratch must be a CUDA uint8 tensor of at least {nq8 + nk8 + nv8 + 512} bytes")
    base = (scratch.data_ptr() + 255) & ~255
    q8, k8, v8 = base, base + ((nq8 + 255) & ~255), base + ((nq8 + 255) & ~255) + ((nk8 + 255) & ~255)
    if v8 + nv8 > scratch.data_ptr() + scratch.numel():
        raise ValueError("attention_f8: scratch too small after alignment")
    st = _stream()
    check(lib.sg_attn_f8_pack(q.data_ptr(), q.stride(1), q.stride(0), q8, B, heads, Nq, 0, st), "sg_attn_f8_pack(q)")
    check(lib.sg_attn_f8_pack(k.data_ptr(), k.stride(1), k.stride(0), k8, Bk, heads, Nk, 0, st), "sg_attn_f8_pack(k)")
    check(lib.sg_attn_f8_pack(vt.data_ptr(), vt.stride(1), vt.stride(0), v8, Bk, heads, Nk, 1, st), "sg_attn_f8_pack(vt)")
    with _timed("attention_f8_d40", 4.0 * B * heads * Nq * Nk * 40, f"B{B} H{heads} Nq{Nq} Nk{Nk}"):
        check(lib.sg_attn_fwd_f8_d40(q8, k8, v8, out.data_ptr(), out.stride(1), out.stride(0), B, heads, Nq, Nk, Bk, scale, st),
              "sg_attn_fwd_f8_d40")
    return out


def softmax_rows(scores: torch.Tensor, probs: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """probs[m, :N] = softmax(scale * scores[m, :N]) (fp32 -> fp16); probs has N rounded up to 8 columns, the extra ones zeroed."""
    _f32(scores, "scores"), _f16(probs, "probs")
    M, N = scores.shape
    if probs.shape[0] != M or probs.shape[1] != ((N + 7) & ~7):
        raise ValueError(f"softmax_rows: probs must be [{M},{(N + 7) & ~7}], got {tuple(probs.shape)}")
    with _timed("softmax", M * N * 6.0, f"M{M} N{N}", aux=True):
        check(lib.sg_softmax_rows_f16(scores.data_ptr(), _row_stride(scores, "scores"), probs.data_ptr(), _row_stride(probs, "probs"), M, N,
                                      float(scale), _stream()), "sg_softmax_rows_f16")
    return probs


def attention_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, heads: int, scale: float, causal: bool,
                    key_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Short-sequence attention (T <= 128, head dim <= 64), q/k/v/out [B,T,H*D] token-/batch-strided fp16 views, V not transposed;
    key_bias = optional fp32 [B,T] additive term (padding mask)."""
    for n, t in (("q", q), ("k", k), ("v", v), ("out", out)):
        _f16(t, n)
        if t.dim() != 3 or t.stride(-1) != 1 or t.shape != q.shape:
            raise ValueError(f"attention_small: {n} must be [B,T,H*D] with a contiguous last dimension")
    B, T, Cq = q.shape
    if key_bias is not None:
        _f32(key_bias, "key_bias")
        if tuple(key_bias.shape) != (B, T) or not key_bias.is_contiguous():
            raise ValueError("attention_small: key_bias must be a contiguous [B,T] tensor")
    check(lib.sg_attn_small_f16(q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), k.stride(1), k.stride(0), v.data_ptr(), v.stride(1),
                                v.stride(0), out.data_ptr(), out.stride(1), out.stride(0), _p(key_bias), B, heads, T, Cq // heads,
                                float(scale), int(causal), _stream()), "sg_attn_small_f16")
    return out


ACT_QUICK_GELU, ACT_GELU = 0, 1


def act_rows(x: torch.Tensor, act: int) -> torch.Tensor:
    """In-place quick_gelu / gelu over a row-strided fp16 [M,N] view."""
    _f16(x, "x")
    check(lib.sg_act_rows_f16(x.data_ptr(), _row_stride(x, "x"), x.shape[0], x.shape[1], act, _stream()), "sg_act_rows_f16")
    return x


def embed_tokens(ids: torch.Tensor, tok: torch.Tensor, pos: torch.Tensor, out: torch.Tensor, T: int) -> torch.Tensor:
    """out[r] = tok[ids[r]] + pos[r % T]; ids int64 [rows] (already validated against the vocabulary), fp32 tables / output."""
    _f32(tok, "tok"), _f32(pos, "pos"), _f32(out, "out")
    if ids.dtype != torch.int64 or not ids.is_cuda or not ids.is_contiguous() or ids.numel() != out.shape[0]:
        raise ValueError("embed_tokens: ids must be a contiguous CUDA int64 tensor with one id per output row")
    if not tok.is_contiguous() or not pos.is_contiguous() or tok.shape[1] != out.shape[1] or pos.shape[1] != out.shape[1] or pos.shape[0] < T:
        raise ValueError("embed_tokens: table shapes do not match the output")
    check(lib.sg_embed_tokens_f32(ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), out.data_ptr(), _row_stride(out, "out"), out.shape[0], T,
                                  out.shape[1], _stream()), "sg_embed_tokens_f32")
    return out


def gaussian_sample(mean: torch.Tensor, logvar: Optional[torch.Tensor], noise: Optional[torch.Tensor], out: torch.Tensor,
                    scale: float = 1.0) -> torch.Tensor:
    """out = (mean + exp(0.5 clamp(logvar, -30, 20)) * noise) * scale; noise None -> mean * scale.  Contiguous fp32 tensors."""
    for n, t in (("mean", mean), ("logvar", logvar), ("noise", noise), ("out", out)):
        if t is not None:
            _f32(t, n)
            if not t.is_contiguous() or t.numel() != mean.numel():
                raise ValueError(f"gaussian_sample: {n} must be contiguous with {mean.numel()} elements")
    check(lib.sg_gaussian_sample_f32(mean.data_ptr(), _p(logvar), _p(noise), out.data_ptr(), float(scale), mean.numel(), _stream()),
          "sg_gaussian_sample_f32")
    return out


def groupnorm_workspace_bytes(B: int, groups: int) -> int:
    return lib.sg_groupnorm_workspace_bytes(B, groups)


def groupnorm_uses_pstats(HW: int, Cc: int, groups: int) -> bool:
    """True when a GroupNorm of this shape consumes producer statistics (the two-launch wide variant)."""
    return bool(lib.sg_groupnorm_uses_pstats(HW, Cc, groups))


def groupnorm_is_fused(HW: int, Cc: int, groups: int) -> bool:
    """True when a GroupNorm of this shape runs as the one-launch kernel (the only variant that takes split=...)."""
    return bool(lib.sg_groupnorm_is_fused(HW, Cc, groups))


def groupnorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, groups: int, eps: float,
              silu: bool, workspace: torch.Tensor, xcopy: Optional[torch.Tensor] = None, pstats: Optional[list] = None,
              split: Optional[dict] = None) -> torch.Tensor:
    """x [B, HW, C] channels-last fp16/fp32 (row-strided views allowed).  out is either [B, HW, C] fp16 or the
    zero-bordered image [B, H+2, W+2, C] (then only the interior is written); xcopy = optional fp16 [B, HW, C] raw copy.
    pstats = [(buffer, rows per partial, channels), ...] (one or two sources, in channel order): statistics written by the
    producers' epilogues (gemm / conv3x3 stats=...) — the wide variant then skips its own statistics pass.
    split = dict(ws=workspace of a conv3x3(defer_reduce=True) launch, splits=its K slices, bias=, rowbias= [B, >= C] fp32, res1= [B, HW, C]
    fp16 / fp32, store=bool): x is that launch's UNREDUCED output — the kernel sums the slices and applies bias / rowbias / res1 while it
    loads (sg_groupnorm_desc.split_*); `x` itself is then only the destination of the reduced tensor, written when store=True (its dtype
    also says whether the normalisation sees fp16-rounded values)."""
    x_f32 = _act(x, "x")
    _f16(gamma, "gamma"), _f16(beta, "beta"), _f16(out, "out")
    B, HW, Cc = x.shape
    if x.stride(0) != HW * x.stride(1):
        raise ValueError("groupnorm: batch stride of x must equal HW * row stride")
    d = GroupNormDesc()
    d.x, d.ldx, d.x_f32 = x.data_ptr(), x.stride(1), int(x_f32)
    if out.dim() == 4:
        Hp, Wp = out.shape[1], out.shape[2]
        if (Hp - 2) * (Wp - 2) != HW or out.stride(1) != Wp * out.stride(2) or out.stride(0) != Hp * out.stride(1):
            raise ValueError("groupnorm: padded output must be a dense [B, H+2, W+2, C] image")
        d.y, d.ldy, d.y_pad_w = out.data_ptr(), out.stride(2), Wp - 2
    else:
        if out.stride(0) != HW * out.stride(1):
            raise ValueError("groupnorm: batch stride of out must equal HW * row stride")
        d.y, d.ldy, d.y_pad_w = out.data_ptr(), out.stride(1), 0
    if xcopy is not None:
        _f16(xcopy, "xcopy")
        if xcopy.stride(0) != HW * xcopy.stride(1):
            raise ValueError("groupnorm: batch stride of xcopy must equal HW * row stride")
        d.xcopy, d.ldxc = xcopy.data_ptr(), xcopy.stride(1)
    d.gamma, d.beta = gamma.data_ptr(), beta.data_ptr()
    d.B, d.HW, d.C, d.groups, d.eps, d.silu = B, HW, Cc, groups, eps, int(silu)
    d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    if pstats:
        c0 = 0
        for i, (buf, rows, nc) in enumerate(pstats):
            _f32(buf, "pstats")
            d.pstats[i], d.pstats_rows[i], d.pstats_c0[i], d.pstats_nc[i] = buf.data_ptr(), int(rows), c0, int(nc)
            c0 += int(nc)
    nbytes = B * HW * Cc * ((4 if x_f32 else 2) + 2 + (2 if xcopy is not None else 0))
    shape = f"B{B} HW{HW} C{Cc}"
    if split is not None:
        ws, ns = split["ws"], int(split["splits"])
        if ws.numel() * ws.element_size() < ns * B * HW * Cc * 4:
            raise ValueError("groupnorm: split workspace smaller than splits * B * HW * C floats")
        d.split_ws, d.split_count = ws.data_ptr(), ns
        if split.get("bias") is not None:
            _f16(split["bias"], "split bias")
            d.split_bias = split["bias"].data_ptr()
        if split.get("rowbias") is not None:
            _f32(split["rowbias"], "split rowbias")
            d.split_rowbias, d.split_rowbias_ld = split["rowbias"].data_ptr(), _row_stride(split["rowbias"], "split rowbias")
        r = split.get("res1")
        if r is not None:
            if tuple(r.shape) != (B, HW, Cc) or r.stride(0) != HW * r.stride(1) or r.stride(2) != 1:
                raise ValueError("groupnorm: split res1 must be a [B, HW, C] view with one row stride")
            d.split_res, d.split_ldr, d.split_res_f32 = r.data_ptr(), r.stride(1), int(_act(r, "split res1"))
        if split.get("store"):
            d.split_out, d.split_ldo, d.split_out_f32 = x.data_ptr(), x.stride(1), int(x_f32)
        d.split_round_f16 = int(not x_f32)
        nbytes = B * HW * Cc * (4 * ns + 2 + (4 if r is not None and r.dtype == torch.float32 else 0) + ((4 if x_f32 else 2) if split.get("store") else 0))
        shape += f" split{ns}"
    with _timed("groupnorm", nbytes, shape, aux=True):
        check(lib.sg_groupnorm_nhwc_f16(C.byref(d), _stream()), "sg_groupnorm_nhwc_f16")
    return out


def layernorm(x: torch.Tensor, g1: torch.Tensor, b1: torch.Tensor, y1: torch.Tensor, eps: float = 1e-5,
              g2: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None,
              y2: Optional[torch.Tensor] = None) -> None:
    x_f32 = _act(x, "x")
    _f16(y1, "y1")
    M, Cc = x.shape
    with _timed("layernorm", M * Cc * ((4 if x_f32 else 2) + (2 if y2 is None else 4)), f"M{M} C{Cc}{' x2' if y2 is not None else ''}", aux=True):
        check(lib.sg_layernorm_f16(x.data_ptr(), _row_stride(x, "x"), int(x_f32), M, Cc, eps, g1.data_ptr(), b1.data_ptr(), y1.data_ptr(),
                                   _row_stride(y1, "y1"), _p(g2), _p(b2), _p(y2),
                                   0 if y2 is None else _row_stride(y2, "y2"), _stream()), "sg_layernorm_f16")


def timestep_embed(t: torch.Tensor, freqs: torch.Tensor, out: torch.Tensor, flip_sin_to_cos: bool) -> torch.Tensor:
    _f32(t, "t"), _f32(freqs, "freqs"), _f32(out, "out")
    B, dim = out.shape
    check(lib.sg_timestep_embed_f32(t.data_ptr(), freqs.data_ptr(), out.data_ptr(), B, dim, int(flip_sin_to_cos),
                                    _stream()), "sg_timestep_embed_f32")
    return out


def lookup_rows(keys: torch.Tensor, table_keys: torch.Tensor, table: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[b] = table[j] for table_keys[j] == keys[b] (fp32, exact match; NaN rows otherwise) — sg_lookup_rows_f32."""
    _f32(keys, "keys"), _f32(table_keys, "table_keys"), _f32(table, "table"), _f32(out, "out")
    B, N = out.shape
    if keys.numel() != B or table.shape[0] != table_keys.numel() or table.shape[1] != N:
        raise ValueError("lookup_rows: keys [B], table_keys [T], table [T, N], out [B, N]")
    check(lib.sg_lookup_rows_f32(keys.data_ptr(), B, table_keys.data_ptr(), table_keys.numel(), table.data_ptr(), _row_stride(table, "table"),
                                 out.data_ptr(), _row_stride(out, "out"), N, _stream()), "sg_lookup_rows_f32")
    return out


def linear_rows(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, act_in: bool = False,
                act_out: bool = False) -> torch.Tensor:
    _f32(x, "x"), _f16(w, "w"), _f32(out, "out")
    B, K = x.shape
    N = w.shape[0]
    check(lib.sg_linear_rows_f32(x.data_ptr(), _row_stride(x, "x"), w.data_ptr(), _row_stride(w, "w"), _p(bias),
                                 out.data_ptr(), _row_stride(out, "out"), B, N, K, int(act_in), int(act_out), _stream()),
          "sg_linear_rows_f32")
    return out


def add_noise(src: torch.Tensor, noise: torch.Tensor, coef: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[u] = coef[u,0]*src[u] + coef[u,1]*noise[u % N]; src/out [U,...], noise [N,...], coef [U,2] (device)."""
    for n, t in (("src", src), ("noise", noise), ("coef", coef), ("out", out)):
        _f32(t, n)
        if not t.is_contiguous():
            raise ValueError(f"add_noise: {n} must be contiguous")
    U, N = src.shape[0], noise.shape[0]
    if out.shape != src.shape or src.shape[1:] != noise.shape[1:] or coef.numel() != 2 * U:
        raise ValueError("add_noise: shape mismatch")
    check(lib.sg_add_noise_f32(src.data_ptr(), noise.data_ptr(), coef.data_ptr(), out.data_ptr(), U, N, src[0].numel(),
                               _stream()), "sg_add_noise_f32")
    return out


def cfg_ddim_step(eps3: torch.Tensor, latents: torch.Tensor, latents3: Optional[torch.Tensor], coef: torch.Tensor
                  ) -> torch.Tensor:
    _f32(eps3, "eps3"), _f32(latents, "latents"), _f32(coef, "coef")
    N = latents.shape[0]
    check(lib.sg_cfg_ddim_step_f32(eps3.data_ptr(), latents.data_ptr(), _p(latents3), coef.data_ptr(), N,
                                   latents[0].numel(), _stream()), "sg_cfg_ddim_step_f32")
    return latents


def cfg_plms_step(eps3: torch.Tensor, latents: torch.Tensor, latents3: Optional[torch.Tensor], history: torch.Tensor,
                  kept: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    """Guidance combine + PNDM/PLMS update (sg_cfg_plms_step_f32); history fp32 [4, *latents.shape], kept like latents."""
    for n, t in (("eps3", eps3), ("latents", latents), ("history", history), ("kept", kept), ("coef", coef)):
        _f32(t, n)
    if history.numel() != 4 * latents.numel() or kept.numel() != latents.numel() or coef.numel() != 15:
        raise ValueError("cfg_plms_step: history must hold 4 latents, kept 1, coef 15 floats")
    N = latents.shape[0]
    check(lib.sg_cfg_plms_step_f32(eps3.data_ptr(), latents.data_ptr(), _p(latents3), history.data_ptr(), kept.data_ptr(),
                                   coef.data_ptr(), N, latents[0].numel(), _stream()), "sg_cfg_plms_step_f32")
    return latents


def copy_rows(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """dst[b, r, :cols] = src[b, r, :cols] for 3-D views with unit channel stride; fp16->fp16, fp32->fp32 or
    fp32->fp16 (cast)."""
    s32, d32 = _act(src, "src"), _act(dst, "dst")
    if d32 and not s32:
        raise TypeError("copy_rows: fp16 -> fp32 is not supported")
    mode = 0 if not s32 else (1 if d32 else 2)
    Bn, rows, cols = src.shape
    if tuple(dst.shape) != (Bn, rows, cols) or dst.stride(-1) != 1 or src.stride(-1) != 1:
        raise ValueError("copy_rows: shape/stride mismatch")
    with _timed("copy_rows", Bn * rows * cols * ((4 if s32 else 2) + (4 if d32 else 2)), f"B{Bn} R{rows} C{cols}", aux=True):
        check(lib.sg_copy_rows(dst.data_ptr(), dst.stride(1), dst.stride(0), src.data_ptr(), src.stride(1), src.stride(0),
                               Bn, rows, cols, mode, _stream()), "sg_copy_rows")
    return dst


def pad_cast(x: torch.Tensor, out_padded: torch.Tensor) -> torch.Tensor:
    """x [B,H,W,C] fp16/fp32 (single pixel stride) -> interior of the zero-bordered fp16 [B,H+2,W+2,C]."""
    x_f32 = _act(x, "x")
    _f16(out_padded, "out")
    B, H, W, Cc = x.shape
    if tuple(out_padded.shape) != (B, H + 2, W + 2, Cc) or not out_padded.is_contiguous():
        raise ValueError("pad_cast: out must be a contiguous [B,H+2,W+2,C]")
    if x.stride(-1) != 1 or x.stride(1) != W * x.stride(2) or x.stride(0) != H * x.stride(1):
        raise ValueError("pad_cast: x must be [B,H,W,C] with a single pixel stride")
    check(lib.sg_pad_cast_f16(x.data_ptr(), x.stride(2), int(x_f32), out_padded.data_ptr(), Cc, B, H, W, Cc, _stream()),
          "sg_pad_cast_f16")
    return out_padded


# ------------------------------------------------------------------------------------------------ backward pass (config 4)
# Validated on MI355X in round 2 (tests/test_backward_gpu.py).  Kernels: csrc/backward.hip, attention_bwd.hip; formulas: oracle/storygen_backward.py.
def layernorm_bwd(x: torch.Tensor, dy1: torch.Tensor, g1: torch.Tensor, out: torch.Tensor, eps: float = 1e-5,
                  dy2: Optional[torch.Tensor] = None, g2: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                  res_scale: float = 1.0) -> torch.Tensor:
    """out[M,C] (fp32) = res_scale * res + dLayerNorm(x; dy1 * g1 (+ dy2 * g2)).  x, dy fp16 or fp32 (both dy alike)."""
    x_f32, dy_f32 = _act(x, "x"), _act(dy1, "dy1")
    _f32(out, "out"), _f16(g1, "g1")
    if dy2 is not None and _act(dy2, "dy2") != dy_f32:
        raise TypeError("layernorm_bwd: dy1 and dy2 must have the same dtype")
    if res is not None:
        _f32(res, "res")
    M, Cc = x.shape
    check(lib.sg_layernorm_bwd_f16(x.data_ptr(), _row_stride(x, "x"), int(x_f32), dy1.data_ptr(), _row_stride(dy1, "dy1"),
                                   g1.data_ptr(), _p(dy2), 0 if dy2 is None else _row_stride(dy2, "dy2"), _p(g2), int(dy_f32),
                                   _p(res), 0 if res is None else _row_stride(res, "res"), float(res_scale), out.data_ptr(),
                                   _row_stride(out, "out"), M, Cc, eps, _stream()), "sg_layernorm_bwd_f16")
    return out


def geglu_bwd(proj_il: torch.Tensor, du: torch.Tensor, dproj_il: torch.Tensor) -> torch.Tensor:
    """proj_il / dproj_il [M, N8] fp16 in the interleaved GEGLU column layout (repack.py), du [M, N8/2] fp16."""
    _f16(proj_il, "proj"), _f16(du, "du"), _f16(dproj_il, "dproj")
    M, N8 = proj_il.shape
    if tuple(du.shape) != (M, N8 // 2) or tuple(dproj_il.shape) != (M, N8):
        raise ValueError("geglu_bwd: shape mismatch")
    check(lib.sg_geglu_bwd_f16(proj_il.data_ptr(), _row_stride(proj_il, "proj"), du.data_ptr(), _row_stride(du, "du"),
                               dproj_il.data_ptr(), _row_stride(dproj_il, "dproj"), M, N8, _stream()), "sg_geglu_bwd_f16")
    return dproj_il


def groupnorm_bwd_workspace_bytes(B: int, groups: int) -> int:
    return lib.sg_groupnorm_bwd_workspace_bytes(B, groups)


def groupnorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, groups: int,
                  eps: float, silu: bool, workspace: torch.Tensor, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x, dy [B, HW, C] (fp16 / fp32, row-strided views allowed).  out: fp32 [B, HW, C] (optionally + res, fp32 [B, HW, C]),
    fp16 [B, HW, C], or the zero-bordered fp16 image [B, H+2, W+2, C] (interior written)."""
    from ._lib import GroupNormBwdDesc
    x_f32, dy_f32 = _act(x, "x"), _act(dy, "dy")
    out_f32 = _act(out, "out")
    _f16(gamma, "gamma"), _f16(beta, "beta")
    B, HW, Cc = x.shape
    if tuple(dy.shape) != (B, HW, Cc) or x.stride(0) != HW * x.stride(1) or dy.stride(0) != HW * dy.stride(1):
        raise ValueError("groupnorm_bwd: x / dy must be [B, HW, C] with a single row stride")
    d = GroupNormBwdDesc()
    d.x, d.ldx, d.x_f32 = x.data_ptr(), x.stride(1), int(x_f32)
    d.dy, d.lddy, d.dy_f32 = dy.data_ptr(), dy.stride(1), int(dy_f32)
    d.gamma, d.beta = gamma.data_ptr(), beta.data_ptr()
    if res is not None:
        _f32(res, "res")
        if tuple(res.shape) != (B, HW, Cc) or res.stride(0) != HW * res.stride(1):
            raise ValueError("groupnorm_bwd: res must be [B, HW, C] with a single row stride")
        d.res, d.ldr = res.data_ptr(), res.stride(1)
    if out.dim() == 4:
        Hp, Wp = out.shape[1], out.shape[2]
        if out_f32 or (Hp - 2) * (Wp - 2) != HW or not out.is_contiguous():
            raise ValueError("groupnorm_bwd: a padded output must be a contiguous fp16 [B, H+2, W+2, C] image")
        d.out, d.ldo, d.out_f32, d.out_pad_w = out.data_ptr(), out.stride(2), 0, Wp - 2
    else:
        if tuple(out.shape) != (B, HW, Cc) or out.stride(0) != HW * out.stride(1):
            raise ValueError("groupnorm_bwd: out must be [B, HW, C] with a single row stride")
        d.out, d.ldo, d.out_f32, d.out_pad_w = out.data_ptr(), out.stride(1), int(out_f32), 0
    d.B, d.HW, d.C, d.groups, d.eps, d.silu = B, HW, Cc, groups, eps, int(silu)
    d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    check(lib.sg_groupnorm_bwd_nhwc_f16(C.byref(d), _stream()), "sg_groupnorm_bwd_nhwc_f16")
    return out


def transpose(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst [C, M] fp16 = src [M, C]^T (src fp16 or fp32)."""
    s32 = _act(src, "src")
    _f16(dst, "dst")
    M, Cc = src.shape
    if tuple(dst.shape) != (Cc, M):
        raise ValueError("transpose: dst must be [C, M]")
    check(lib.sg_transpose_f16(src.data_ptr(), _row_stride(src, "src"), int(s32), dst.data_ptr(), _row_stride(dst, "dst"), M, Cc,
                               _stream()), "sg_transpose_f16")
    return dst


def transpose_batched(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst [B, C, M] fp16 = src [B, M, C] transposed per batch row, ONE launch (src fp16 or fp32; row- and batch-strided views allowed)."""
    s32 = _act(src, "src")
    _f16(dst, "dst")
    B, M, Cc = src.shape
    if tuple(dst.shape) != (B, Cc, M) or src.stride(2) != 1 or dst.stride(2) != 1:
        raise ValueError("transpose_batched: dst must be [B, C, M], last dimensions contiguous")
    check(lib.sg_transpose_batched_f16(src.data_ptr(), src.stride(1), src.stride(0), int(s32), dst.data_ptr(), dst.stride(1), dst.stride(0),
                                       B, M, Cc, _stream()), "sg_transpose_batched_f16")
    return dst


def sum2x2(du: torch.Tensor, dx: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """du [B, 2H, 2W, C] fp32 -> dx [B, H, W, C] fp32 (single pixel stride each)."""
    _f32(du, "du"), _f32(dx, "dx")
    B, H, W, Cc = dx.shape
    if tuple(du.shape) != (B, 2 * H, 2 * W, Cc) or du.stride(-1) != 1 or dx.stride(-1) != 1:
        raise ValueError("sum2x2: shape mismatch")
    for t, hh, ww in ((du, 2 * H, 2 * W), (dx, H, W)):
        if t.stride(1) != ww * t.stride(2) or t.stride(0) != hh * t.stride(1):
            raise ValueError("sum2x2: tensors must have a single pixel stride")
    check(lib.sg_sum2x2_f32(du.data_ptr(), du.stride(2), dx.data_ptr(), dx.stride(2), B, H, W, Cc, int(accumulate), _stream()),
          "sg_sum2x2_f32")
    return dx


def zero_stuff(dy: torch.Tensor, out_padded: torch.Tensor) -> torch.Tensor:
    """dy [B, Ho, Wo, C] (fp16 / fp32, single pixel stride) -> interior of the zero-bordered fp16 [B, 2Ho+2, 2Wo+2, C]."""
    d32 = _act(dy, "dy")
    _f16(out_padded, "out")
    B, Ho, Wo, Cc = dy.shape
    if tuple(out_padded.shape) != (B, 2 * Ho + 2, 2 * Wo + 2, Cc) or not out_padded.is_contiguous():
        raise ValueError("zero_stuff: out must be a contiguous [B, 2Ho+2, 2Wo+2, C]")
    if dy.stride(-1) != 1 or dy.stride(1) != Wo * dy.stride(2) or dy.stride(0) != Ho * dy.stride(1):
        raise ValueError("zero_stuff: dy must be [B, Ho, Wo, C] with a single pixel stride")
    check(lib.sg_zero_stuff_f16(dy.data_ptr(), dy.stride(2), int(d32), out_padded.data_ptr(), Cc, B, Ho, Wo, Cc, _stream()),
          "sg_zero_stuff_f16")
    return out_padded


def mse_grad(pred: torch.Tensor, noise: torch.Tensor, mask: torch.Tensor, d_pred: torch.Tensor, loss: torch.Tensor) -> torch.Tensor:
    """loss[0] = mean(((pred - noise) * (1 - mask))^2), d_pred = its gradient; all fp32, contiguous, same shape."""
    for n, t in (("pred", pred), ("noise", noise), ("mask", mask), ("d_pred", d_pred), ("loss", loss)):
        _f32(t, n)
        if not t.is_contiguous():
            raise ValueError(f"mse_grad: {n} must be contiguous")
    check(lib.sg_mse_grad_f32(pred.data_ptr(), noise.data_ptr(), mask.data_ptr(), d_pred.data_ptr(), loss.data_ptr(),
                              pred.numel(), _stream()), "sg_mse_grad_f32")
    return d_pred


def _attn_desc(q, k, vt, out, heads, scale, nk):
    for n, t in (("q", q), ("k", k), ("vt", vt), ("out", out)):
        _f16(t, n)
        if t.dim() != 3 or t.stride(2) != 1:
            raise ValueError(f"attention: {n} must be 3-D with a contiguous last dimension")
    B, Nq, Cq = q.shape
    Bk = k.shape[0]
    Nk = k.shape[1] if nk is None else nk
    D = Cq // heads
    d = AttnDesc()
    d.q, d.ldq, d.bsq = q.data_ptr(), q.stride(1), q.stride(0)
    d.k, d.ldk, d.bsk = k.data_ptr(), k.stride(1), k.stride(0)
    d.vt, d.ldvt, d.bsvt = vt.data_ptr(), vt.stride(1), vt.stride(0)
    d.o, d.ldo, d.bso = out.data_ptr(), out.stride(1), out.stride(0)
    d.B, d.H, d.Nq, d.Nk, d.D, d.kv_batches, d.scale = B, heads, Nq, Nk, D, (0 if Bk == B else Bk), scale
    return d


def attention_lse(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, lse2: torch.Tensor, heads: int,
                  scale: float, nk: Optional[int] = None) -> torch.Tensor:
    """Training forward: `attention` that also stores lse2 [B, H, Nq] fp32 (log2-domain log-sum-exp rows)."""
    _f32(lse2, "lse2")
    if tuple(lse2.shape) != (q.shape[0], heads, q.shape[1]) or not lse2.is_contiguous():
        raise ValueError("attention_lse: lse2 must be a contiguous [B, H, Nq]")
    d = _attn_desc(q, k, vt, out, heads, scale, nk)
    with _timed(f"attention_d{d.D}", 4.0 * d.B * d.H * d.Nq * d.Nk * d.D, f"B{d.B} H{d.H} Nq{d.Nq} Nk{d.Nk} lse"):
        check(lib.sg_attn_fwd_lse_f16(C.byref(d), lse2.data_ptr(), _stream()), "sg_attn_fwd_lse_f16")
    return out


def attention_bwd_prep(o: torch.Tensor, dout: torch.Tensor, lse2: torch.Tensor, ld2: torch.Tensor, heads: int) -> torch.Tensor:
    """ld2 [B, H, Nq, 2] fp32 = (lse2, delta = rowsum_d(dO * O)) per (batch, head, query); o, dout [B, Nq, H*D] fp16."""
    _f16(o, "o"), _f16(dout, "dout"), _f32(lse2, "lse2"), _f32(ld2, "ld2")
    B, Nq, Cq = o.shape
    if tuple(ld2.shape) != (B, heads, Nq, 2) or not ld2.is_contiguous() or not lse2.is_contiguous():
        raise ValueError("attention_bwd_prep: ld2 must be a contiguous [B, H, Nq, 2], lse2 a contiguous [B, H, Nq]")
    check(lib.sg_attn_bwd_prep_f32(o.data_ptr(), o.stride(1), o.stride(0), dout.data_ptr(), dout.stride(1), dout.stride(0),
                                   lse2.data_ptr(), ld2.data_ptr(), B, heads, Nq, Cq // heads, _stream()), "sg_attn_bwd_prep_f32")
    return ld2


def _attn_bwd_desc(q, k, v, dout, ld2, heads, scale):
    from ._lib import AttnBwdDesc
    for n, t in (("q", q), ("k", k), ("v", v), ("dout", dout)):
        _f16(t, n)
        if t.dim() != 3 or t.stride(2) != 1:
            raise ValueError(f"attention_bwd: {n} must be [B, N, H*D] with a contiguous last dimension")
    _f32(ld2, "ld2")
    B, Nq, Cq = q.shape
    d = AttnBwdDesc()
    d.q, d.ldq, d.bsq = q.data_ptr(), q.stride(1), q.stride(0)
    d.k, d.ldk, d.bsk = k.data_ptr(), k.stride(1), k.stride(0)
    d.v, d.ldv, d.bsv = v.data_ptr(), v.stride(1), v.stride(0)
    d.dout, d.lddo, d.bsdo = dout.data_ptr(), dout.stride(1), dout.stride(0)
    d.ld2 = ld2.data_ptr()
    d.B, d.H, d.Nq, d.Nk, d.D, d.scale = B, heads, Nq, k.shape[1], Cq // heads, scale
    return d


def attention_bwd_dq(q: torch.Tensor, k: torch.Tensor, kt: torch.Tensor, v: torch.Tensor, dout: torch.Tensor, ld2: torch.Tensor,
                     dq: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """dq [B, Nq, H*D] = scale * dS K.  kt [B, H*D, Nk] = K transposed (keys contiguous)."""
    d = _attn_bwd_desc(q, k, v, dout, ld2, heads, scale)
    _f16(kt, "kt"), _f16(dq, "dq")
    d.kt, d.ldkt, d.bskt = kt.data_ptr(), kt.stride(1), kt.stride(0)
    d.dq, d.lddq, d.bsdq = dq.data_ptr(), dq.stride(1), dq.stride(0)
    with _timed("attention_bwd", 6.0 * d.B * d.H * d.Nq * d.Nk * d.D, f"dq B{d.B} H{d.H} Nq{d.Nq} Nk{d.Nk} D{d.D}"):     # S, dP, dQ: 1.5 x forward
        check(lib.sg_attn_bwd_dq_f16(C.byref(d), _stream()), "sg_attn_bwd_dq_f16")
    return dq


def attention_bwd_dkv(q: torch.Tensor, qt: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dout: torch.Tensor, dot: torch.Tensor,
                      ld2: torch.Tensor, dkt: torch.Tensor, dvt: torch.Tensor, heads: int, scale: float):
    """dkt, dvt [B, H*D, Nk] (transposed: keys contiguous).  qt, dot [B, H*D, Nq] = Q, dO transposed."""
    d = _attn_bwd_desc(q, k, v, dout, ld2, heads, scale)
    for n, t in (("qt", qt), ("dot", dot), ("dkt", dkt), ("dvt", dvt)):
        _f16(t, n)
    d.qt, d.ldqt, d.bsqt = qt.data_ptr(), qt.stride(1), qt.stride(0)
    d.dot, d.lddot, d.bsdot = dot.data_ptr(), dot.stride(1), dot.stride(0)
    d.dkt, d.lddkt, d.bsdkt = dkt.data_ptr(), dkt.stride(1), dkt.stride(0)
    d.dvt, d.lddvt, d.bsdvt = dvt.data_ptr(), dvt.stride(1), dvt.stride(0)
    with _timed("attention_bwd", 8.0 * d.B * d.H * d.Nq * d.Nk * d.D, f"dkv B{d.B} H{d.H} Nq{d.Nq} Nk{d.Nk} D{d.D}"):    # S, dP, dK, dV: 2 x forward
        check(lib.sg_attn_bwd_dkv_f16(C.byref(d), _stream()), "sg_attn_bwd_dkv_f16")
    return dkt, dvt


def debug_set_tile(bm: int = 0, bn: int = 0, no_pipe: bool = False) -> None:
    """Test hook: force the GEMM/conv tile shape / kernel family (0, 0 = automatic)."""
    check(lib.sg_debug_set_tile(bm, bn, int(no_pipe)), "sg_debug_set_tile")


def debug_set_option(name: str, value: int) -> None:
    """Development option of the library (sg_debug_set_option; names in include/storygen_hip.h)."""
    check(lib.sg_debug_set_option(name.encode(), int(value)), "sg_debug_set_option")


# The development tools (tools/*.py, tests) historically selected kernel variants through SG_* environment variables.  The
# library no longer reads the environment: this maps the variables onto sg_debug_set_option, and only when a tool asks for it.
_ENV_OPTIONS = {"SG_NO_NMAJOR": "no_nmajor", "SG_NO_PIPE": "no_pipe", "SG_NO_SPLIT": "no_split",
                "SG_ATTN_SUB2": "attn_sub2", "SG_ATTN_PRIO": "attn_prio", "SG_ATTN_D80": "attn_d80", "SG_ATTN_D160": "attn_d160",
                "SG_FF_VARIANT": "ff_variant", "SG_PIPE_STAGES": "pipe_stages", "SG_GN_FUSED_NT": "gn_fused_nt", "SG_GN_CHUNKS": "gn_chunks", "SG_ATTN_LEAN": "attn_lean", "SG_ATTN_D40_GENERAL": "attn_d40_general", "SG_NO_GN_FUSED": "gn_no_fused", "SG_GN_WIDE": "gn_wide", "SG_GN_FUSED_MAX": "gn_fused_max",
                "SG_LAT_TILES": "lat_tiles", "SG_LAT_MIN_KT": "lat_min_kt", "SG_LAT_MAX_KT": "lat_max_kt", "SG_LAT_STAGES": "lat_stages", "SG_LAT_WIDE": "lat_wide", "SG_LAT_MASK": "lat_mask", "SG_LAT_WIDE_M": "lat_wide_m", "SG_FAT_M": "fat_m", "SG_BIG_M": "big_m", "SG_BIG_BM": "big_bm", "SG_BIG_BN": "big_bn"}


def apply_env_options() -> dict:
    """Development tools only (never called by the product path): SG_* variables -> sg_debug_set_option.  Returns what was set."""
    import os
    done = {}
    for var, name in _ENV_OPTIONS.items():
        if os.environ.get(var, "") != "":
            debug_set_option(name, int(os.environ[var]))
            done[name] = int(os.environ[var])
    if os.environ.get("SG_TILE"):
        bm, bn = (int(v) for v in os.environ["SG_TILE"].split(","))
        debug_set_option("tile_m", bm), debug_set_option("tile_n", bn)
        done["tile"] = (bm, bn)
    return done


def debug_mfma_f8(a_bytes: torch.Tensor, b_bytes: torch.Tensor, scale_a: int = 127, scale_b: int = 127) -> torch.Tensor:
    """a_bytes, b_bytes: uint8 [64, 32] (raw fp8 e4m3 operand bytes per lane) -> fp32 [64, 16] accumulator registers."""
    if a_bytes.dtype != torch.uint8 or tuple(a_bytes.shape) != (64, 32) or not a_bytes.is_cuda or not a_bytes.is_contiguous():
        raise TypeError("debug_mfma_f8: a must be a contiguous CUDA uint8 [64, 32]")
    if b_bytes.dtype != torch.uint8 or tuple(b_bytes.shape) != (64, 32) or not b_bytes.is_cuda or not b_bytes.is_contiguous():
        raise TypeError("debug_mfma_f8: b must be a contiguous CUDA uint8 [64, 32]")
    out = torch.empty(64, 16, dtype=torch.float32, device=a_bytes.device)
    check(lib.sg_debug_mfma_f8_32x32x64(a_bytes.data_ptr(), b_bytes.data_ptr(), out.data_ptr(), scale_a, scale_b, _stream()),
          "sg_debug_mfma_f8_32x32x64")
    return out


def debug_mfma(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty(64, 16, dtype=torch.float32, device=a.device)
    check(lib.sg_debug_mfma_32x32x16(a.data_ptr(), b.data_ptr(), out.data_ptr(), _stream()), "sg_debug_mfma_32x32x16")
    return out
