"""UNetEngine — StoryGen's UNet forward on the HIP kernels (fp16 channels-last activations, fp32 accumulate).

Re-design of /root/reference/model/unet_2d_condition.py:338-485 (+ unet_2d_blocks.py forward paths and
attention.py:85-128,236-302) around what the MI355X wants rather than around nn.Module calls:

  * every activation is a channels-last matrix [B*H*W, C] — fp32 for the residual stream, fp16 for everything that
    feeds an MFMA; the transformer's token layout IS that matrix, so the reference's permute/reshape pairs
    (attention.py:103,117) disappear;
  * weights are repacked once and shared by every engine (EngineWeights): KRSC convs, fused q|k projections, V
    projections kept separate because they are computed TRANSPOSED (VT = Wv . X^T, the attention kernel's operand
    layout), [Wo2 | Wo3] for the single out-projection of the two cross-attentions, 32/32-interleaved GEGLU, all 22
    time_emb_proj stacked into one GEMV bundle whose output, plus the conv1 bias, becomes conv1's per-sample
    channel bias;
  * residual adds, biases, the `(a2+h)+(a3+h)` combine (attention.py:277,291-293) and GEGLU live in GEMM/conv
    epilogues; nearest-2x upsampling and stride-2 are folded into the conv gather; `torch.cat([h, skip])`
    costs nothing: the producer of h and the down-path producer of the skip both write their columns of the
    consuming resnet's concat buffer;
  * harvested features (attention.py:263) are written straight into slot r of the preallocated context buffers
    [B, R*HW, C] that the main pass cross-attends to (replaces the clones at unet_2d_condition.py:428-429,445,
    468-470 and the token-axis concat at pipeline.py:440-443);
  * all buffers are allocated up front, nothing allocates or synchronises inside `forward`, timestep values are
    read from device memory — so a whole pass (or a whole denoising step) can be captured in one hipGraph;
  * independent branches inside a pass (1x1 shortcut beside conv1 -> norm2, text attention beside image attention)
    can run on a side stream (`forward(side=...)`), and a reference pass can also run the attn3 K / V^T projections
    of each context it completes (HarvestPlan.kv), taking them off the main pass's critical path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import ops
from .arch import BlockSpec, ResnetSpec, UNetArch, XfSpec, feature_shapes
from .repack import conv1x1_nk, conv3x3_krsc, conv_in_kn, ff_fused_pack, fold_layernorm, interleave_geglu

F16 = torch.float16
GN_EPILOGUE_STATS = True   # GroupNorm statistics from the producing conv / GEMM epilogues (False = every GroupNorm makes its own pass)
# Experiment, default off: the residual stream INSIDE a transformer block (h0: proj_in output, h1: after
# self-attention, h2/h3: after the cross-attentions) in fp16 instead of fp32; block inputs / outputs and the resnets stay fp32.  Halves
# the bytes of 3 GEMM outputs, 3 residual reads and 3 LayerNorm reads per block; costs fp16 rounding of the stream.  tools/exp_fp16_stream.py
FP16_BLOCK_STREAM = False
FP16_RESNET_STREAM = False   # same experiment for the rest of the stream: resnet / transformer / shortcut outputs and the skip-concat buffers
PAIR_GEMMS = True     # q|k + V^T, q2 + q3, k3 + v3^T as one launch each (sg_gemm_pair_f16); False = two launches (A/B switch)
# LayerNorm folded into the GEMM that consumes it (round 3; model/attention.py:250,268,283,298): the producers of h0 / h1 / h3 also
# write the raw fp16 copy and per-token partial statistics, the q|k / V^T / q2 / q3 / GEGLU projections run on the raw copy with
# gamma-scaled weights and normalise in their epilogues (sg_gemm_desc.ln_*).  No LayerNorm launch is left in a pass (-94 launches
# per step).  False = the separate layernorm kernel (A/B switch).
LN_FOLD = True
# Text and image cross-attention of a block in one launch (sg_attn_fwd_pair_f16).  Measured on the contract workload (r3c12: 14.69 /
# 14.70 ms paired vs 14.62 ms): the text attention already runs on the side stream beside the context K / V^T projections, and one
# grid after them gives that overlap up — so the default stays two launches (A/B: bench.py --attn-pair).
ATTN_PAIR = False
# Split-K convolutions at the 16x16 / 8x8 levels leave their second pass to the one-launch GroupNorm that consumes them (round 4;
# sg_conv3x3_desc.defer_reduce + sg_groupnorm_desc.split_*): conv1 -> norm2 of every ResnetBlock2D there, and conv2 -> Transformer2DModel.norm
# where a transformer follows.  Bit-identical values, one launch and one round trip of the tensor less per site.  False = separate
# splitk_reduce launches (A/B switch).
SPLITK_IN_GN = True
# GEGLU feed-forward of a transformer block as ONE launch where the kernel exists (C = 320, the 64x64 level: sg_ff_geglu_fused_f16;
# SURVEY 8(f) rank 2 — cross-layer fusion): norm3 -> Linear(C, 8C) -> a gelu(g) -> Linear(4C, C) -> + h with the [M, 4C] intermediate
# kept in registers.  False = LayerNorm-folded GEGLU GEMM + second GEMM (A/B switch: bench.py --no-ff-fused).
FF_FUSED = True
# Fused feed-forward launches of at most this many tokens split the hidden dimension over two workgroups per 128 tokens (round 5;
# sg_ff_desc.hidden_split): 12 288 tokens of the main pass are 96 workgroups on 256 CUs and the launch lasts as long as ONE wave's walk
# over the 40 hidden chunks.  The two partial sums land in the halves of an [M, 2C] buffer and proj_out contracts them with [W | W].
# 0 = never (A/B switch: bench.py --no-ff-split).
FF_SPLIT_MAX_TOKENS = 16384
# ff.net.2 and proj_out are two linear maps with nothing but a residual add between them (model/attention.py:300 `ff(norm3(h)) + h`, then
# :121-123 proj_out + the block's input): out = W_out (W_2 g + b_2 + h3) + b_out + x = [g | h3] [W_out W_2 | W_out]^T + (W_out b_2 + b_out) + x
# — ONE GEMM with K = 4C + C over the GEGLU output and the raw fp16 copy of h3 side by side in one buffer, the same FLOPs, one launch
# (and one fp16 rounding of an intermediate) less per transformer block at C = 640 / 1280 (round 6; the 64x64 level runs the fused
# feed-forward kernel instead).  False = the two GEMMs as written (A/B switch: bench.py --no-ff-proj-merge).
FF_PROJ_MERGE = True
LN_EPS = 1e-5         # torch.nn.LayerNorm default, as the reference constructs norm1..norm4 (model/attention.py:213-233)


PAIR_FALLBACK_LAT = False   # development (tools/exp_determinism.py): let the unpaired form of a paired projection take the latency kernel by size
VT_LAT_TILE = (64, 64, 4)   # development: the hint VT_LAT_FILTER applies
VT_CHECK = None             # development: int64 device counter (see VT_LAT_FILTER)
VT_MASK = VT_DIFF = None    # development: per-element mismatch count / last difference of the checked V^T launches
VT_HASH = None              # development: dict(buf=[K, 4] int64, ctr=[1] int64): per hinted V^T launch the integer sums of its inputs (raw copy, LayerNorm partials) and outputs (q|k, V^T), in launch order
VT_LAT_FILTER = None        # development: callable(prefix, consume) -> bool; with PAIR_GEMMS off, the V^T projection (columns-are-tokens fold) of the
                            # transformers it selects is HINTED onto the latency kernel, every other unpaired projection is kept off it


def _pair(first, second):
    if PAIR_GEMMS:
        ops.gemm_pair(first, second)
    else:       # (the projections that are paired stay off the latency kernel either way: tile hint (0, 0, -1))
        for args, kw in (first, second):
            ops.gemm(*args, **(kw if PAIR_FALLBACK_LAT else {**kw, "tile": (0, 0, -1)}))


class _Resnet:
    __slots__ = ("spec", "n1g", "n1b", "w1", "n2g", "n2b", "w2", "b2", "wsc", "bsc", "temb_off")


class _Xf:
    __slots__ = ("spec", "ng", "nb", "w_in", "b_in", "w_out", "b_out", "ln", "w_qk1", "w_v1", "w_o1", "b_o1", "w_q2", "w_k2",
                 "w_v2", "w_o2", "b_o2", "w_q3", "w_k3", "w_v3", "w_o3", "b_o3", "w_o23", "b_o23", "w_ff1", "b_ff1", "w_ff2", "b_ff2",
                 # LayerNorm-folded copies (repack.fold_layernorm): weight gamma (.) W in fp16, c / d in fp32
                 "w_qk1f", "c_qk1", "d_qk1", "w_v1f", "c_v1", "d_v1", "w_q2f", "c_q2", "d_q2", "w_q3f", "c_q3", "d_q3",
                 "w_ff1f", "c_ff1", "d_ff1",
                 "ff_pack",     # weight stream of the fused feed-forward kernel (repack.ff_fused_pack), or None
                 "w_out2",      # [W_out | W_out]: proj_out over the two partial sums of a hidden-split fused feed-forward
                 "w_ffo", "b_ffo")   # FF_PROJ_MERGE: [W_out W_ff2 | W_out] (fp32 product rounded once) and W_out b_ff2 + b_out, or None


class EngineWeights:
    """The checkpoint repacked once into the layouts the kernels consume (fp16, on the device).  Shared by every
    UNetEngine built on it (the reference-pass engine and the main-pass engine of the sampler use one copy)."""

    def __init__(self, arch: UNetArch, state_dict: Dict[str, torch.Tensor], device):
        self.arch, self.dev, self.cfg = arch, torch.device(device), arch.config
        self.version = 0                 # bumped by reload_(): anything derived from the weights and cached by an engine (time tables) is stale then
        self._load_weights(state_dict)

    def _w(self, sd, key) -> torch.Tensor:
        return sd[key].detach().to(device=self.dev, dtype=F16).contiguous()

    def _load_weights(self, sd):
        dev, arch = self.dev, self.arch
        g = lambda k: self._w(sd, k)  # noqa: E731
        self.w_conv_in = conv_in_kn(g("conv_in.weight"))
        self.b_conv_in = g("conv_in.bias")
        self.w_t1, self.b_t1 = g("time_embedding.linear_1.weight"), g("time_embedding.linear_1.bias")
        self.w_t2, self.b_t2 = g("time_embedding.linear_2.weight"), g("time_embedding.linear_2.bias")
        half = cfg_half = self.cfg["block_out_channels"][0] // 2
        # diffusers Timesteps: exp(-ln(1e4) * j / (half - shift)) computed in fp32 by torch, as the reference does
        self.freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32)
                               / (cfg_half - self.cfg["freq_shift"])).to(dev)
        self.resnets: Dict[str, _Resnet] = {}
        temb_w, temb_b, off = [], [], 0
        for r in arch.resnets:
            p = r.prefix
            o = _Resnet()
            o.spec = r
            o.n1g, o.n1b = g(f"{p}.norm1.weight"), g(f"{p}.norm1.bias")
            o.w1 = conv3x3_krsc(g(f"{p}.conv1.weight"))
            o.n2g, o.n2b = g(f"{p}.norm2.weight"), g(f"{p}.norm2.bias")
            o.w2, o.b2 = conv3x3_krsc(g(f"{p}.conv2.weight")), g(f"{p}.conv2.bias")
            if r.has_shortcut:
                o.wsc, o.bsc = conv1x1_nk(g(f"{p}.conv_shortcut.weight")), g(f"{p}.conv_shortcut.bias")
            else:
                o.wsc = o.bsc = None
            temb_w.append(g(f"{p}.time_emb_proj.weight"))
            # conv1 bias + time_emb_proj bias are both per-channel constants of the same add (resnet forward)
            temb_b.append((sd[f"{p}.time_emb_proj.bias"].float() + sd[f"{p}.conv1.bias"].float()).to(dev, F16))
            o.temb_off = off
            off += r.cout
            self.resnets[p] = o
        self.w_temb = torch.cat(temb_w, 0).contiguous()
        self.b_temb = torch.cat(temb_b, 0).contiguous()
        self.temb_total = off
        self.xfs: Dict[str, _Xf] = {}
        for blk in arch.down + [arch.mid] + arch.up:
            for a in blk.attns:
                if a is None:
                    continue
                p, t = a.prefix, f"{a.prefix}.transformer_blocks.0"
                o = _Xf()
                o.spec = a
                o.ng, o.nb = g(f"{p}.norm.weight"), g(f"{p}.norm.bias")
                o.w_in, o.b_in = conv1x1_nk(g(f"{p}.proj_in.weight")), g(f"{p}.proj_in.bias")
                o.w_out, o.b_out = conv1x1_nk(g(f"{p}.proj_out.weight")), g(f"{p}.proj_out.bias")
                o.ln = {n: (g(f"{t}.{n}.weight"), g(f"{t}.{n}.bias")) for n in ("norm1", "norm2", "norm3", "norm4")}
                # q and k projections fused (one GEMM, token-major output); v separate: it is computed TRANSPOSED
                # (VT = Wv . X^T, the same GEMM with its operands swapped), the layout the attention kernel streams
                o.w_qk1 = torch.cat([g(f"{t}.attn1.to_q.weight"), g(f"{t}.attn1.to_k.weight")], 0).contiguous()
                o.w_v1 = g(f"{t}.attn1.to_v.weight")
                o.w_o1, o.b_o1 = g(f"{t}.attn1.to_out.0.weight"), g(f"{t}.attn1.to_out.0.bias")
                o.w_q2 = g(f"{t}.attn2.to_q.weight")
                o.w_k2, o.w_v2 = g(f"{t}.attn2.to_k.weight"), g(f"{t}.attn2.to_v.weight")
                o.w_o2, o.b_o2 = g(f"{t}.attn2.to_out.0.weight"), g(f"{t}.attn2.to_out.0.bias")
                o.w_q3 = g(f"{t}.attn3.to_q.weight")
                o.w_k3, o.w_v3 = g(f"{t}.attn3.to_k.weight"), g(f"{t}.attn3.to_v.weight")
                o.w_o3, o.b_o3 = g(f"{t}.attn3.to_out.0.weight"), g(f"{t}.attn3.to_out.0.bias")
                # main pass: (attn2.to_out(a2) + h) + (attn3.to_out(a3) + h) (attention.py:277,291-293) is ONE GEMM over the
                # concatenated heads [a2 | a3] with weights [Wo2 | Wo3] (K = 2C), bias b2 + b3 and the residual h twice
                o.w_o23 = torch.cat([o.w_o2, o.w_o3], dim=1).contiguous()
                o.b_o23 = (sd[f"{t}.attn2.to_out.0.bias"].float() + sd[f"{t}.attn3.to_out.0.bias"].float()).to(dev, F16)
                o.w_ff1, o.b_ff1 = interleave_geglu(g(f"{t}.ff.net.0.proj.weight"), g(f"{t}.ff.net.0.proj.bias"))
                o.w_ff2, o.b_ff2 = g(f"{t}.ff.net.2.weight"), g(f"{t}.ff.net.2.bias")
                for name, val in self._pack_folds(o).items():
                    setattr(o, name, val)
                o.ff_pack = ff_fused_pack(o.w_ff1f, o.d_ff1, o.w_ff2) if (ops.ff_fused_supported(a.channels) and dev.type == "cuda") else None
                o.w_out2 = torch.cat([o.w_out, o.w_out], dim=1).contiguous() if o.ff_pack is not None else None
                if o.ff_pack is None:         # ff.net.2 and proj_out as one K = 5C GEMM (FF_PROJ_MERGE)
                    wo32 = o.w_out.float()
                    o.w_ffo = torch.cat([(wo32 @ o.w_ff2.float()).to(F16), o.w_out], dim=1).contiguous()
                    o.b_ffo = (wo32 @ o.b_ff2.float() + o.b_out.float()).to(F16).contiguous()
                else:
                    o.w_ffo = o.b_ffo = None
                self.xfs[p] = o
        self.samplers = {}
        for blk in arch.down + arch.up:
            if blk.sampler_prefix:
                self.samplers[blk.sampler_prefix] = (conv3x3_krsc(g(f"{blk.sampler_prefix}.weight")),
                                                     g(f"{blk.sampler_prefix}.bias"))
        self.gn_out = (g("conv_norm_out.weight"), g("conv_norm_out.bias"))
        self.w_conv_out = conv3x3_krsc(g("conv_out.weight"))
        self.b_conv_out = g("conv_out.bias")

    @staticmethod
    def _pack_folds(xf: "_Xf", which=("qk1", "v1", "q2", "q3", "ff1")) -> Dict[str, torch.Tensor]:
        """LayerNorm-folded operand copies of the projections that follow norm1 (q|k, v), norm2 (q2), norm4 (q3) and norm3 (GEGLU):
        W' = gamma (.) W in fp16, c = row sums of W', d = W beta (+ bias)."""
        src = {"qk1": (xf.w_qk1, None, "norm1"), "v1": (xf.w_v1, None, "norm1"), "q2": (xf.w_q2, None, "norm2"),
               "q3": (xf.w_q3, None, "norm4"), "ff1": (xf.w_ff1, xf.b_ff1, "norm3")}
        out = {}
        for k in which:
            w, b, n = src[k]
            wf, c, d = fold_layernorm(w, b, *xf.ln[n])
            out[f"w_{k}f"], out[f"c_{k}"], out[f"d_{k}"] = wf.contiguous(), c.contiguous(), d.contiguous()
        return out

    # ---- in-place refresh: captured hipGraphs and engines hold the ADDRESSES of the repacked tensors, so new parameter
    # values are copied into the existing storage (same shapes) instead of re-allocating
    def _pack_attn3(self, sd, xf: "_Xf") -> Dict[str, torch.Tensor]:
        t = f"{xf.spec.prefix}.transformer_blocks.0"
        g = lambda k: self._w(sd, k)  # noqa: E731
        w_o3 = g(f"{t}.attn3.to_out.0.weight")
        return {"w_q3": g(f"{t}.attn3.to_q.weight"), "w_k3": g(f"{t}.attn3.to_k.weight"), "w_v3": g(f"{t}.attn3.to_v.weight"),
                "w_o3": w_o3, "b_o3": g(f"{t}.attn3.to_out.0.bias"), "w_o23": torch.cat([xf.w_o2, w_o3], dim=1),
                "b_o23": (sd[f"{t}.attn2.to_out.0.bias"].float() + sd[f"{t}.attn3.to_out.0.bias"].float()).to(self.dev, F16)}

    def refresh_attn3_(self, sd, prefixes=None):
        """Stage-2 training changes only the attn3 modules (train_StorySalon_stage2.py:170-177): re-pack those of the given
        Transformer2DModel prefixes (default: all) from `sd`, in place."""
        for p, xf in self.xfs.items():
            if prefixes is None or p in prefixes:
                for name, val in self._pack_attn3(sd, xf).items():
                    getattr(xf, name).copy_(val)
                for name, val in self._pack_folds(xf, ("q3",)).items():
                    getattr(xf, name).copy_(val)

    def _pack_attn1(self, sd, xf: "_Xf") -> Dict[str, torch.Tensor]:
        t = f"{xf.spec.prefix}.transformer_blocks.0"
        g = lambda k: self._w(sd, k)  # noqa: E731
        return {"w_qk1": torch.cat([g(f"{t}.attn1.to_q.weight"), g(f"{t}.attn1.to_k.weight")], 0), "w_v1": g(f"{t}.attn1.to_v.weight"),
                "w_o1": g(f"{t}.attn1.to_out.0.weight"), "b_o1": g(f"{t}.attn1.to_out.0.bias")}

    def refresh_attn1_(self, sd, prefixes=None):
        """Stage-1 training changes only the attn1 modules (train_StorySalon_stage1.py:175-179): the same in-place re-pack for them."""
        for p, xf in self.xfs.items():
            if prefixes is None or p in prefixes:
                for name, val in self._pack_attn1(sd, xf).items():
                    getattr(xf, name).copy_(val)
                for name, val in self._pack_folds(xf, ("qk1", "v1")).items():
                    getattr(xf, name).copy_(val)

    def reload_(self, sd):
        """Re-pack the whole checkpoint into the existing tensors (any parameter may have changed)."""
        fresh = EngineWeights.__new__(EngineWeights)
        fresh.arch, fresh.dev, fresh.cfg = self.arch, self.dev, self.cfg
        fresh._load_weights(sd)

        def walk(old, new):
            if torch.is_tensor(old):
                old.copy_(new)
            elif isinstance(old, dict):
                for k in old:
                    walk(old[k], new[k])
            elif isinstance(old, (tuple, list)):
                for a, b in zip(old, new):
                    walk(a, b)
            elif isinstance(old, (_Resnet, _Xf)):
                for k in old.__slots__:
                    if k != "spec" and getattr(old, k, None) is not None:
                        walk(getattr(old, k), getattr(new, k))
        for k, v in self.__dict__.items():
            if k not in ("arch", "dev", "cfg", "version"):
                walk(v, getattr(fresh, k))
        self.version += 1


class HarvestPlan:
    """Where a reference pass puts the features it harvests (attention.py:263).

    `ctx` are the context buffers of the engine that will consume them ([rows, R*HW, C] fp16 per feature key) and each
    op (src, src_step, dst_row, dst_slot, count) means: for j < count, slot dst_slot+j of ctx row dst_row <- the
    feature of sample src + j*src_step of this pass.  One strided copy per op (src_step = 0 broadcasts one sample
    into `count` frame slots)."""

    def __init__(self, ctx: Dict[str, torch.Tensor], ops_: List[tuple], kv: Optional[Dict[str, tuple]] = None,
                 src_offset: int = 0, short: int = 0, slots_per_row: Optional[int] = None, direct: bool = False,
                 identity: bool = False):
        # kv: feature key -> (K [rows*R*HW, C], VT [C, rows*R*HW]) buffers: when given, the reference pass also runs the
        # attn3 K / V^T projections of each finished context (they depend on nothing else), taking them off the main
        # pass's critical path.
        # src_offset: added to every op's `src` — a reference pass that batches the samples of several denoising steps
        # (sampler ref_ahead > 1) passes one plan per step, each pointing at that step's slice of the batch.
        # short / slots_per_row: layout of the DESTINATION buffers (UNetEngine ctx_short): the first `short` context rows hold one
        # frame slot, the others slots_per_row; flat slot of (row, slot) = row if row < short else short + (row - short) * R + slot.
        # direct: the pass's sample u belongs in flat slot u for every u (sampler._plan orders the batch that way), so the
        # producer of the feature writes its fp16 copy straight into the context buffer and no copy kernel runs.
        # identity: the CALLER guarantees that mapping for buffers it laid out itself — `ctx` / `kv` are flat [samples * HW, C]
        # matrices in the pass's sample order (the sampler's group schedule: the context sets of G steps end to end); no ops.
        self.ctx, self.ops, self.kv, self.src_offset = ctx, list(ops_), kv, int(src_offset)
        self.short, self.slots_per_row, self.direct = int(short), slots_per_row, bool(direct) or bool(identity)
        self.identity = bool(identity)
        if self.identity and (self.ops or self.src_offset):
            raise ValueError("an identity harvest plan has no copy ops and no source offset")

    def flat_slot(self, row: int, slot: int, R: int) -> int:
        return row if row < self.short else self.short + (row - self.short) * R + slot

    def is_direct(self, n_samples: int, R: int) -> bool:
        """True when this plan alone maps sample u -> flat slot u for all n_samples samples of the pass."""
        if self.identity:
            return all(c.dim() == 2 and c.shape[0] % n_samples == 0 for c in self.ctx.values())
        if not self.direct or self.src_offset:
            return False
        seen = {}
        for src, step, row, slot, cnt in self.ops:
            for j in range(cnt):
                seen[self.flat_slot(row, slot + j, R)] = src + j * step
        return seen == {u: u for u in range(n_samples)}


class UNetEngine:
    def __init__(self, arch: UNetArch, state_dict: Optional[Dict[str, torch.Tensor]], device, batch: int, height: int,
                 width: int, n_ref: int = 0, seq_len: int = 77, splitk_workspace_mb: int = 96,
                 weights: Optional[EngineWeights] = None, ctx_rows: Optional[int] = None,
                 attn3_groups: Optional[List[tuple]] = None, fp8_attention: bool = False, ctx_short: int = 0,
                 cfg_shared_head: bool = False, fp16_stream: Optional[tuple] = None):
        """fp16_stream = (block, resnet) or None (= the module switches FP16_BLOCK_STREAM / FP16_RESNET_STREAM): per ENGINE, which
        parts of the residual stream are stored as fp16 — block: the transformer state h0 / h1 (h2 / h3 stay fp32: they feed the fused
        feed-forward and the merged proj_out); resnet: block inputs / outputs, shortcut outputs, skip / concat buffers.
        batch = samples per UNet call; n_ref = R prior frames (sizes the context buffers; 0 = an engine that only
        harvests, into another engine's buffers).  ctx_rows = number of distinct context rows (default: one per
        sample); attn3_groups = [(q0, n, c0), ...]: samples [q0, q0+n) cross-attend to context rows [c0, c0+n) — lets
        samples whose prior-frame features are identical (the two image-conditioned CFG branches) share
        one copy of the context and of its K/V projection.  ctx_short = S > 0: the first S context rows hold ONE frame slot instead
        of R (the zero-image rows of `multi-image-condition`, whose R slots would be R copies of one feature map: softmax over R
        copies of the same keys is softmax over one); the context buffers are then flat [(S + (rows - S) R) HW, C] matrices.
        cfg_shared_head: the caller guarantees the classifier-free-guidance pattern of the loop's main pass (pipeline.py:448-453) with
        one story frame: batch 3 = the SAME latent and timestep three times, text rows [uncond, uncond, text], context rows [zero-image,
        frames, frames].  Until the first cross-attention the three samples are then the same computation, so forward(consume=True) runs
        conv_in, the first ResnetBlock2D and the first transformer's norm / proj_in / self-attention / to_out / query projections ONCE
        (batch 1), the text attention for (uncond, text) and the image attention for (zero-image, frames) only, and copies the results to
        the samples that share them — the same arithmetic on the same values (bit-exact redundancy removal)."""
        self.arch, self.dev = arch, torch.device(device)
        self.B, self.H, self.W, self.R, self.S = batch, height, width, n_ref, seq_len
        cfg = arch.config
        self.cfg = cfg
        nlev = len(cfg["block_out_channels"])
        if height % (1 << (nlev - 1)) or width % (1 << (nlev - 1)):
            raise ValueError(f"latent size {height}x{width} must be divisible by {1 << (nlev - 1)}")
        self.groups, self.eps = cfg["norm_num_groups"], cfg["norm_eps"]
        self.cad = cfg["cross_attention_dim"]
        self.hw = [(height >> l) * (width >> l) for l in range(nlev)]
        self.Sp = (seq_len + 7) & ~7          # text tokens padded with zero rows to a multiple of 8 (V^T row alignment)
        for k, lv in arch.feature_level.items():
            if self.hw[lv] % 8:
                raise ValueError(f"latent {height}x{width} leaves {self.hw[lv]} tokens at attention block {k}; the V^T layout "
                                 "needs a multiple of 8 (use a latent of at least 32x32 with the 4-level SD-1.5 UNet)")
        self.ctx_rows = batch if ctx_rows is None else ctx_rows
        self.attn3_groups = [(0, batch, 0)] if attn3_groups is None else list(attn3_groups)
        covered = sorted(q for q0, n, _ in self.attn3_groups for q in range(q0, q0 + n))
        if n_ref and (covered != list(range(batch)) or any(c0 + n > self.ctx_rows for _, n, c0 in self.attn3_groups)):
            raise ValueError(f"attn3_groups {self.attn3_groups} must cover samples 0..{batch - 1} once, within {self.ctx_rows} context rows")
        # the common sharing pattern maps onto the kernel's kv_batches argument (one launch instead of one per group)
        rows = self.ctx_rows
        share = sorted(self.attn3_groups) == ([(0, batch, 0)] if rows == batch else [(0, rows, 0), (rows, batch - rows, 2 * rows - batch)])
        self.attn3_share = rows if share else None
        self.ctx_short = int(ctx_short)
        if self.ctx_short and (not n_ref or self.attn3_share is None or self.ctx_short > rows):
            raise ValueError("ctx_short needs context buffers, at most ctx_rows short rows and the shared-row attn3 pattern")
        self.ctx_slots = self.ctx_short + (rows - self.ctx_short) * n_ref      # frame slots per feature key
        first = arch.down[0]
        self.cfg_shared_head = bool(cfg_shared_head)
        if self.cfg_shared_head and not (batch == 3 and n_ref and rows == 2 and self.attn3_share == 2 and self.ctx_short in (0, 1)
                                         and first.attns and first.attns[0] is not None and not fp8_attention):
            raise ValueError("cfg_shared_head needs batch 3 on 2 shared context rows (the CFG main pass of one story frame), a transformer "
                             "behind the first resnet and the fp16 attention path")
        self._shared_back = False
        self._head_checked = False
        self.fp16_block, self.fp16_resnet = (FP16_BLOCK_STREAM, FP16_RESNET_STREAM) if fp16_stream is None else map(bool, fp16_stream)
        self.wts = weights if weights is not None else EngineWeights(arch, state_dict, device)
        # BASELINE config 5: the head-dim-40 self / image attentions (the 46 080-key context of the 96x96 level) on the fp8 MFMA
        # path (sg_attn_fwd_f8_d40); text attention and the D = 80 / 160 levels stay fp16
        self.fp8_attention = bool(fp8_attention)
        # GroupNorm statistics as producer epilogues (north-star; HISTORY.md 5): a conv / GEMM whose output feeds a wide GroupNorm also
        # writes per-(row tile, channel) sums, and that GroupNorm skips its own statistics pass.  Per producer site: one buffer and
        # the cached answer of sg_*_stats_tile_rows (0 = this launch cannot emit them -> the consumer makes its own pass).
        self.gn_epilogue_stats = GN_EPILOGUE_STATS
        self._stats_buf: Dict[str, torch.Tensor] = {}
        self._stats_rows: Dict[str, int] = {}
        self._stats_want: Dict[tuple, bool] = {}
        self._pstats: Dict[int, tuple] = {}           # data_ptr of a produced tensor -> (buffer, rows per partial, channels)
        self._splits: Dict[str, int] = {}             # conv site -> planned K slices (sg_conv3x3_planned_splits), asked once
        self._pending_split: Dict[int, dict] = {}     # data_ptr of a tensor whose split-K reduction its GroupNorm consumer will do
        self.text_cache: Dict[str, torch.Tensor] = {}
        # sticky flags of the LayerNorm fold's two assumptions (ops.LN_GUARD_*; include/storygen_hip.h): every producer / consumer of a
        # folded LayerNorm ORs into this word; check_ln_guard() reads it (one 4-byte D2H copy: call it where the host synchronises anyway)
        self.ln_guard = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._alloc(splitk_workspace_mb)

    def __getattr__(self, name):
        # repacked weights live in the shared EngineWeights object
        wts = self.__dict__.get("wts")
        if wts is not None and hasattr(wts, name):
            return getattr(wts, name)
        raise AttributeError(name)

    # ------------------------------------------------------------------------------------------ buffers
    def _buf(self, *shape, dtype=F16) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def _alloc(self, splitk_mb: int):
        """Precision plan: the residual stream (block inputs/outputs, skips, concat buffers, the transformer's running
        state h0..h3, shortcut outputs) is fp32; every MFMA operand (norm outputs, q/k/v, attention outputs, FF inner,
        conv inputs) is fp16.  Conv inputs live in zero-bordered [B,H+2,W+2,C] buffers so padding needs no predicate."""
        B, arch = self.B, self.arch
        boc = self.cfg["block_out_channels"]
        nlev = len(boc)
        F32 = torch.float32
        self.x_in = self._buf(B, self.cfg["in_channels"], self.H, self.W, dtype=F32)
        self.t_in = self._buf(B, dtype=F32)
        self.text_pad = torch.zeros(B, self.Sp, self.cad, dtype=F16, device=self.dev)    # rows >= S stay zero
        self.text_in = self.text_pad[:, : self.S]
        self.eps_out = self._buf(B, self.cfg["out_channels"], self.H, self.W, dtype=F32)
        self.temb0 = self._buf(B, boc[0], dtype=F32)
        self.temb1 = self._buf(B, arch.temb_dim, dtype=F32)
        self.temb2 = self._buf(B, arch.temb_dim, dtype=F32)
        self.tproj = self._buf(B, self.temb_total, dtype=F32)
        self.time_table: Optional[tuple] = None                         # (timesteps [T], rows [T, temb_total]) — build_time_table()
        self.time_table_on = True                                       # False: forward() runs the chain itself (set_inputs); the buffers stay alive
        self.ws_split = ops.new_workspace(splitk_mb << 20, self.dev)      # split-K partial tiles (fp32, reduced by a second launch: no initialisation needed)
        self.ws_side = ops.new_workspace(splitk_mb << 20, self.dev)       # split-K scratch of the side-stream branches
        self.ws_pair = ops.new_workspace(splitk_mb << 20, self.dev)       # second problem of a paired GEMM launch
        self.side: Optional[torch.cuda.Stream] = None                   # set by forward(side=...)
        self.kv_ext: Optional[Dict[str, tuple]] = None                  # attn3 K / V^T computed by the reference pass
        self.ws_gn = self._buf(ops.groupnorm_workspace_bytes(B, self.groups), dtype=torch.uint8)
        self.f8_scratch = None
        if self.fp8_attention:
            heads = self.cfg["attention_head_dim"] if isinstance(self.cfg["attention_head_dim"], int) else self.cfg["attention_head_dim"][0]
            hw0 = self.hw[0]
            nk = max(hw0, self.R * hw0)
            kb = max(B, self.ctx_rows if self.R else B)
            need = (ops.attention_f8_bytes(B, heads, hw0, False) + ops.attention_f8_bytes(kb, heads, nk, False)
                    + ops.attention_f8_bytes(kb, heads, nk, True) + 4096)
            self.f8_scratch = self._buf(need, dtype=torch.uint8)
        # per level: widest resnet input (concat) and the level's channel count; every conv-input channel count
        cmax, cout = [0] * nlev, [0] * nlev
        conv_in_ch = [set() for _ in range(nlev)]
        lvl = 0
        for blk in arch.down:
            for r in blk.resnets:
                cmax[lvl], cout[lvl] = max(cmax[lvl], r.cin), max(cout[lvl], r.cout)
                conv_in_ch[lvl].update((r.cin, r.cout))
            if blk.sampler_prefix:
                conv_in_ch[lvl].add(blk.channels)
                lvl += 1
        for r in arch.mid.resnets:
            cmax[lvl], cout[lvl] = max(cmax[lvl], r.cin), max(cout[lvl], r.cout)
            conv_in_ch[lvl].update((r.cin, r.cout))
        for blk in arch.up:
            for r in blk.resnets:
                cmax[lvl], cout[lvl] = max(cmax[lvl], r.cin), max(cout[lvl], r.cout)
                conv_in_ch[lvl].update((r.cin, r.cout))
            if blk.sampler_prefix:
                conv_in_ch[lvl].add(blk.channels)
                lvl -= 1
        self.lv = []
        self.padded: Dict[tuple, torch.Tensor] = {}
        for l in range(nlev):
            M, C, Cw = B * self.hw[l], cout[l], max(cmax[l], cout[l])
            h, w = self.H >> l, self.W >> l
            for ch in conv_in_ch[l]:
                self.padded[(l, ch)] = torch.zeros(B, h + 2, w + 2, ch, dtype=F16, device=self.dev)
            f32 = lambda *sh: self._buf(*sh, dtype=F32)  # noqa: E731
            d = dict(
                # fp32 residual stream
                **{n: (self._buf(M, C) if self.fp16_resnet else f32(M, C)) for n in ("r", "t_out", "sc")},
                **{n: (self._buf(M, C) if self.fp16_block else f32(M, C)) for n in ("h0", "h1")},
                # h2 / h3 (the input of the feed-forward; in a reference pass h2 plays h3) stay fp32 unless the MODULE switch asks for
                # the round-2 experiment's all-fp16 block stream: the fused feed-forward kernel reads the fp32 stream
                **{n: (self._buf(M, C) if (self.fp16_block and FP16_BLOCK_STREAM) else f32(M, C)) for n in ("h2", "h3")},
                # fp16 MFMA operands
                gn=self._buf(M, C), x16=self._buf(M * Cw), c1=self._buf(M, C), h4=self._buf(M, C),
                ln=self._buf(M, C), ln4=self._buf(M, C), qk=self._buf(M, 2 * C),
                # LayerNorm fold: per-token (sum, M2) partials of h0 / h1 / h3 per 64-channel block
                **{n: torch.zeros(M, (C // 64 + 1) & ~1, 2, dtype=F32, device=self.dev) for n in ("lnst0", "lnst1", "lnst3")}, vt=self._buf(C, M), q=self._buf(M, C),
                att=self._buf(M, C), q2=self._buf(M, C), att23=self._buf(M, 2 * C), ffi=self._buf(M, 4 * C),
                ffo=self._buf(M, 5 * C),          # FF_PROJ_MERGE: [GEGLU output | raw fp16 copy of h3], the K = 5C operand of the merged GEMM
                kt=self._buf(B * self.Sp, C), vtt=self._buf(C, B * self.Sp),
                ki=self._buf(self.ctx_slots * self.hw[l], C) if self.R else None,
                vti=self._buf(C, self.ctx_slots * self.hw[l]) if self.R else None,
            )
            self.lv.append(d)
        # torch.cat([h, skip]) (unet_2d_blocks.py:609,626,716) without a copy: every up-path resnet owns its concat input
        # buffer [M, cin] (fp32); the producer of h writes columns [0, c_h) and the skip tensor (down_block_res_samples)
        # IS the column slice [c_h, cin) of the buffer of the resnet that will pop it — the down path writes it there.
        self.up_cats: List[torch.Tensor] = []
        skip_views: List[torch.Tensor] = []
        lvl = nlev - 1
        for blk in arch.up:
            for j, r in enumerate(blk.resnets):
                buf = self._buf(B * self.hw[lvl], r.cin, dtype=F16 if self.fp16_resnet else F32)
                self.up_cats.append(buf)
                skip_views.append(buf[:, r.cin - blk.skip_channels[j]:])
            if blk.sampler_prefix:
                lvl -= 1
        self.skips: List[torch.Tensor] = list(reversed(skip_views))       # skips[i] = i-th tensor the down path produces
        self.skip_meta = [(0, boc[0])]
        lvl = 0
        for blk in arch.down:
            for _ in blk.resnets:
                self.skip_meta.append((lvl, blk.channels))
            if blk.sampler_prefix:
                lvl += 1
                self.skip_meta.append((lvl, blk.channels))
        assert len(self.skips) == len(self.skip_meta)
        for sk, (l, c) in zip(self.skips, self.skip_meta):
            assert tuple(sk.shape) == (B * self.hw[l], c), (tuple(sk.shape), l, c)
        # context buffers: [ctx_rows, R*HW, C] fp16 per feature key (K/V projection operands of attn3); with short rows a flat
        # [ctx_slots * HW, C] matrix (short rows first)
        self.ctx: Dict[str, torch.Tensor] = {}
        if self.R:
            for k, (n, c) in feature_shapes(arch, self.H, self.W).items():
                self.ctx[k] = self._buf(self.ctx_slots * n, c) if self.ctx_short else self._buf(self.ctx_rows, self.R * n, c)

    # ------------------------------------------------------------------------------------------ layers
    def _img(self, x2d: torch.Tensor, lvl: int) -> torch.Tensor:
        h, w = self.H >> lvl, self.W >> lvl
        return x2d.unflatten(0, (self.B, h, w))

    def _fork(self) -> bool:
        """Start a side branch (independent kernels on the side stream) — inside a capture this becomes a parallel
        branch of the hipGraph.  Returns False when no side stream is configured (everything stays in order)."""
        if self.side is None:
            return False
        self.side.wait_stream(torch.cuda.current_stream(self.dev))
        return True

    def _join(self):
        torch.cuda.current_stream(self.dev).wait_stream(self.side)

    # ---- cfg_shared_head: run a stretch of the pass on the first n samples of every buffer
    def _enter_head(self, n: int):
        """Make self.B = n and point the level-0 buffers, the conv inputs and the time rows at their first n samples.  Returns the
        state _leave_head restores."""
        saved = (self.B, self.lv[0], self.padded, self.tproj)
        hw, B = self.hw[0], self.B
        cols = ("vt", "vtt", "vti")                          # transposed buffers [C, tokens]: the samples are column blocks
        head = {}
        for k, v in self.lv[0].items():
            if v is None:
                head[k] = None
            elif k in cols:
                head[k] = v[:, : v.shape[1] // B * n]
            else:
                head[k] = v[: v.shape[0] // B * n]
        self.B, self.lv[0] = n, head
        self.padded = {k: (v[:n] if k[0] == 0 else v) for k, v in saved[2].items()}
        self.tproj = saved[3][:n]
        return saved

    def _leave_head(self, saved):
        self.B, self.lv[0], self.padded, self.tproj = saved

    def _spread(self, t2d: torch.Tensor, n: int, copies: int):
        """Rows of samples [0, n) of a [B*hw, C] stream tensor (row-strided views allowed) -> the next `copies` groups of n samples."""
        rows = n * self.hw[0]
        src = t2d[:rows].unflatten(0, (1, rows)).expand(copies, rows, t2d.shape[1])
        ops.copy_rows(t2d[rows:(1 + copies) * rows].unflatten(0, (copies, rows)), src)

    # ---- GroupNorm statistics from producer epilogues
    def _stats_for(self, site: str, lvl: int, n_out: int, query) -> Optional[torch.Tensor]:
        """The statistics buffer a producer at `site` (output [B*hw[lvl], n_out]) should write, or None.  `query(buf)` -> rows per
        partial of exactly that launch (ops.*_stats_rows); asked once per site (the plan depends only on the shapes)."""
        if not self.gn_epilogue_stats or not self._stats_wanted(lvl, n_out):
            return None
        rows = self._stats_rows.get(site)
        if rows is None:
            M = self.B * self.hw[lvl]
            buf = torch.empty((M // 64) * 2 * n_out, dtype=torch.float32, device=self.dev)    # enough for any tile height >= 64
            rows = self._stats_rows[site] = int(query(buf))
            if rows:
                self._stats_buf[site] = buf
        return self._stats_buf.get(site) if rows else None

    def _stats_wanted(self, lvl: int, n_out: int) -> bool:
        """Does any GroupNorm that can read a tensor of n_out channels at this level take producer statistics?  Its own width, or the
        wider channel concats [h | skip] of the up blocks (unet_2d_blocks.py:600-601) — e.g. at 16x16 a 1280-channel GroupNorm is
        the one-launch kernel, the 2560- and 1920-channel concats are not.  (Statistics nobody reads cost the producer a few
        shuffles; a consumer that does not use them ignores them.)"""
        key = (lvl, n_out)
        if key not in self._stats_want:
            widths = [n_out] + sorted({r.cin for r in self.arch.resnets if r.cin > n_out})
            self._stats_want[key] = any(ops.groupnorm_uses_pstats(self.hw[lvl], c, self.groups) for c in widths)
        return self._stats_want[key]

    def _publish(self, out: torch.Tensor, site: str, n_out: int):
        self._pstats[out.data_ptr()] = (self._stats_buf[site], self._stats_rows[site], n_out)

    def _pstats_of(self, x: torch.Tensor) -> Optional[list]:
        """Producer statistics covering x [M, C] (one producer, or two for a channel concat [h | skip]), else None."""
        C = x.shape[1]
        a = self._pstats.get(x.data_ptr())
        if a is None:
            return None
        if a[2] == C:
            return [a]
        if a[2] < C:
            b = self._pstats.get(x[:, a[2]:].data_ptr())
            if b is not None and a[2] + b[2] == C:
                return [a, b]
        return None

    def _defer_splits(self, site: str, lvl: int, cout: int, query) -> int:
        """K slices (> 1) of the convolution at `site` when it may leave its split-K reduction to the GroupNorm that consumes its
        [B*hw, cout] output (the one-launch variant sums the slices while loading its slab), else 0."""
        if not SPLITK_IN_GN or not ops.groupnorm_is_fused(self.hw[lvl], cout, self.groups):
            return 0
        # asked on every call: the plan is host-only arithmetic, but it depends on process state that can change after the first
        # forward (ops.load_tile_table, sg_debug_set_option) — and the GroupNorm must be told the slice count THIS launch uses
        n = self._splits[site] = int(query())
        return n if n > 1 else 0

    def _attention(self, q, k, vt, out, heads: int, scale: float, nk: Optional[int] = None, short: Optional[tuple] = None):
        """softmax(scale q k^T) v on the HIP kernels: fp16 MFMA, or e4m3 MFMA for the D = 40 image / self attentions when the
        engine was built with fp8_attention (text attention — 77 keys — and every other head dim stay fp16).
        short = (k2, vt2): leading K/V rows with their own key count (ops.attention)."""
        n_keys = k.shape[1] if nk is None else nk
        if self.fp8_attention and q.shape[2] == heads * 40 and n_keys >= 256:
            if short is not None:          # the fp8 kernel has one key count per launch: short rows first, then the others
                ns = short[0].shape[0]
                ops.attention_f8(q[:ns], short[0], short[1], out[:ns], heads, scale, self.f8_scratch)
                q, out = q[ns:], out[ns:]
            ops.attention_f8(q, k, vt, out, heads, scale, self.f8_scratch, nk=nk)
        else:
            ops.attention(q, k, vt, out, heads, scale, nk=nk, short=short)

    def _resnet(self, rn: _Resnet, x: torch.Tensor, out: torch.Tensor, lvl: int, defer_out: bool = False):
        """diffusers ResnetBlock2D.  x fp32 [M,Cin] contiguous; out fp32 [M,Cout], possibly a column
        slice of a concat buffer.  defer_out: the only reader of `out` before anything else is the GroupNorm of the transformer that
        follows — a split-K conv2 may leave its reduction to it (SPLITK_IN_GN)."""
        L, M, r, B, hw = self.lv[lvl], x.shape[0], rn.spec, self.B, self.hw[lvl]
        ws = self.ws_split
        p_in, p_mid = self.padded[(lvl, r.cin)], self.padded[(lvl, r.cout)]
        x16 = L["x16"][: M * r.cin].view(M, r.cin) if rn.wsc is not None else None
        ops.groupnorm(x.unflatten(0, (B, hw)), rn.n1g, rn.n1b, p_in, self.groups, self.eps, True, self.ws_gn,
                      xcopy=None if x16 is None else x16.unflatten(0, (B, hw)), pstats=self._pstats_of(x))
        res, forked = x, False
        if rn.wsc is not None:                     # 1x1 shortcut: independent of conv1 -> norm2, runs beside them
            forked = self._fork()
            if forked:
                with torch.cuda.stream(self.side):
                    ops.gemm(x16, rn.wsc, L["sc"], bias=rn.bsc, workspace=self.ws_side)
            else:
                ops.gemm(x16, rn.wsc, L["sc"], bias=rn.bsc, workspace=ws)
            res = L["sc"]
        h1 = L["c1"]
        rb = self.tproj[:, rn.temb_off: rn.temb_off + r.cout]
        kw1 = dict(rowbias=rb, workspace=ws, x_padded=True)
        d1 = self._defer_splits(r.prefix + ".conv1", lvl, r.cout, lambda: ops.conv3x3_planned_splits(p_in, rn.w1, self._img(h1, lvl), **kw1))
        if d1:      # conv1 -> norm2: the GroupNorm sums the K slices itself; h1 is never stored (fp16 rounding kept: bit-identical)
            ops.conv3x3(p_in, rn.w1, self._img(h1, lvl), defer_reduce=True, **kw1)
            self._pstats.pop(h1.data_ptr(), None)
            ops.groupnorm(h1.unflatten(0, (B, hw)), rn.n2g, rn.n2b, p_mid, self.groups, self.eps, True, self.ws_gn,
                          split=dict(ws=ws, splits=d1, rowbias=rb, store=False))
        else:
            s1 = self._stats_for(r.prefix + ".conv1", lvl, r.cout,
                                 lambda buf: ops.conv3x3_stats_rows(p_in, rn.w1, self._img(h1, lvl), stats=buf, **kw1))
            ops.conv3x3(p_in, rn.w1, self._img(h1, lvl), stats=s1, **kw1)
            if s1 is not None:
                self._publish(h1, r.prefix + ".conv1", r.cout)
            else:
                self._pstats.pop(h1.data_ptr(), None)
            ops.groupnorm(h1.unflatten(0, (B, hw)), rn.n2g, rn.n2b, p_mid, self.groups, self.eps, True, self.ws_gn,
                          pstats=self._pstats_of(h1))
        if forked:
            self._join()
        kw2 = dict(bias=rn.b2, res1=self._img(res, lvl), workspace=ws, x_padded=True)
        d2 = 0
        if defer_out:
            d2 = self._defer_splits(r.prefix + ".conv2", lvl, r.cout, lambda: ops.conv3x3_planned_splits(p_mid, rn.w2, self._img(out, lvl), **kw2))
        if d2:      # conv2 -> Transformer2DModel.norm: that GroupNorm reduces, and stores `out` (the residual of proj_out)
            ops.conv3x3(p_mid, rn.w2, self._img(out, lvl), defer_reduce=True, **kw2)
            self._pstats.pop(out.data_ptr(), None)
            self._pending_split[out.data_ptr()] = dict(ws=ws, splits=d2, bias=rn.b2, res1=res.unflatten(0, (B, hw)), store=True)
            return
        s2 = self._stats_for(r.prefix + ".conv2", lvl, r.cout,
                             lambda buf: ops.conv3x3_stats_rows(p_mid, rn.w2, self._img(out, lvl), stats=buf, **kw2))
        ops.conv3x3(p_mid, rn.w2, self._img(out, lvl), stats=s2, **kw2)
        if s2 is not None:
            self._publish(out, r.prefix + ".conv2", r.cout)
        else:
            self._pstats.pop(out.data_ptr(), None)

    def _text_kv(self, xf: _Xf, lvl: int, use_cache: bool):
        """K and V^T projections of the text embeddings for attn2 (attention.py:192-199): timestep-invariant, so the
        sampler computes them once per prompt (cache_text_kv) instead of once per UNet call.  Returns (K [B,Sp,C],
        VT [B,C,Sp]) views; rows / columns >= S come from the zero padding tokens."""
        B, Sp, C = self.B, self.Sp, xf.spec.channels
        if use_cache:
            kt, vtt = self.text_cache[xf.spec.prefix]
        else:
            kt, vtt = self.lv[lvl]["kt"], self.lv[lvl]["vtt"]
            self._project_text(xf, kt, vtt)
        return kt.view(B, Sp, C), vtt.view(C, B, Sp).permute(1, 0, 2)

    def _project_text(self, xf: _Xf, kt: torch.Tensor, vtt: torch.Tensor):
        x = self.text_pad.view(self.B * self.Sp, self.cad)
        _pair(((x, xf.w_k2, kt), dict(workspace=self.ws_split)),
                      ((xf.w_v2, x, vtt), dict(workspace=self.ws_pair)))                   # VT = Wv . X^T

    def cache_text_kv(self):
        """Run every attn2 K / V^T projection on the current self.text_in and keep the results (use with
        forward(text_cache=True) while the prompts do not change)."""
        for blk in self.arch.down + [self.arch.mid] + self.arch.up:
            for a in blk.attns:
                if a is None:
                    continue
                bufs = self.text_cache.get(a.prefix)
                if bufs is None:
                    bufs = self.text_cache[a.prefix] = (self._buf(self.B * self.Sp, a.channels),
                                                        self._buf(a.channels, self.B * self.Sp))
                self._project_text(self.xfs[a.prefix], *bufs)

    def _transformer(self, xf: _Xf, x: torch.Tensor, out: Optional[torch.Tensor], lvl: int, text: Optional[torch.Tensor],
                     harvest: Optional[HarvestPlan], consume: bool, text_cache: bool = False, stop_after_harvest: bool = False,
                     phase: str = "all"):
        """Transformer2DModel.forward (attention.py:85-128) + BasicTransformerBlock.forward (:236-302).
        x, out, h0..h3 fp32; everything that feeds an MFMA fp16.
        phase: "front" = up to and including the query projections of the cross-attentions (everything that depends on the hidden
        states alone), "back" = from the cross-attentions on, "all" = both (cfg_shared_head runs the front once for the three
        CFG samples)."""
        L, B, hw, S = self.lv[lvl], self.B, self.hw[lvl], self.S
        M, C, heads = x.shape[0], xf.spec.channels, xf.spec.heads
        scale = xf.spec.dim_head ** -0.5
        ws = self.ws_split
        h0 = L["h0"]
        fold = LN_FOLD and C % 64 == 0 and C // 64 <= 20
        # fused feed-forward: needs the fp32 stream (it reads h3 itself: no raw copy, no LayerNorm partials from h3's producer)
        ff1 = FF_FUSED and xf.ff_pack is not None and L["h3"].dtype != F16
        # with the fold, the producer of a stream tensor also writes its raw fp16 copy and the LayerNorm partials of its rows
        raw = lambda t, buf: t if t.dtype == F16 else buf                                 # noqa: E731  (fp16 stream: it IS the copy)
        gd = self.ln_guard
        prod = lambda t, buf, st: dict(out2=None if t.dtype == F16 else buf, ln_out=st, guard=gd) if fold else {}   # noqa: E731
        h0r, h1r = raw(h0, L["ln"]), raw(L["h1"], L["ln4"])
        # ff.net.2 + proj_out as one GEMM (FF_PROJ_MERGE): h3's raw copy lives beside the GEGLU output in L["ffo"]
        merge = FF_PROJ_MERGE and fold and not ff1 and xf.w_ffo is not None and L["h3"].dtype != F16 and out is not None
        h3raw = L["ffo"][:M, 4 * C:5 * C] if merge else L["ln"]
        qk, vt = L["qk"], L["vt"]
        wp = self.ws_pair
        att = L["att"]
        h1 = L["h1"]
        q2, q3buf = L["q2"], L["q"]
        if phase != "back":
            ops.groupnorm(x.unflatten(0, (B, hw)), xf.ng, xf.nb, L["gn"].unflatten(0, (B, hw)), self.groups, 1e-6, False,
                          self.ws_gn, pstats=self._pstats_of(x), split=self._pending_split.pop(x.data_ptr(), None))   # :99 (eps 1e-6, :55)
            ops.gemm(L["gn"], xf.w_in, h0, bias=xf.b_in, workspace=ws, **prod(h0, L["ln"], L["lnst0"]))   # proj_in :101
            # --- self-attention :250-262
            # q|k (token-major) and V^T = Wv . X^T (the attention kernel's operand layout): two GEMMs on one LayerNorm output, one launch
            if fold and VT_LAT_FILTER is not None and not PAIR_GEMMS:      # development: bisecting by transformer
                ops.gemm(h0r, xf.w_qk1f, qk, ln=(1, L["lnst0"], xf.c_qk1, xf.d_qk1, LN_EPS), guard=gd, tile=(0, 0, -1))
                sel = VT_LAT_FILTER(xf.spec.prefix, consume)
                ops.gemm(xf.w_v1f, h0r, vt, ln=(2, L["lnst0"], xf.c_v1, xf.d_v1, LN_EPS), guard=gd, tile=VT_LAT_TILE if sel else (0, 0, -1))
                if sel and VT_HASH is not None:       # order-independent integer sums: are the INPUTS of a differing launch the same in both runs?
                    HS = VT_HASH[bool(consume)]          # (one record per engine: the two branches of the graph run concurrently)
                    isum = lambda t, dt: t.contiguous().view(dt).sum(dtype=torch.int64)      # noqa: E731
                    row = torch.stack([isum(h0r, torch.int16), isum(L["lnst0"], torch.int32), isum(qk, torch.int16), isum(vt, torch.int16),
                                       torch.full((), vt.shape[1], dtype=torch.int64, device=vt.device)])
                    HS["buf"].index_copy_(0, HS["ctr"], row.unsqueeze(0))
                    if "vts" in HS:                      # ... and the V^T images themselves (flattened, zero-padded to the row length), + the inputs
                        for key, t in (("vts", vt), ("xs", h0r), ("sts", L["lnst0"])):
                            flat = t.contiguous().reshape(1, -1)
                            HS[key].index_copy_(0, HS["ctr"], torch.nn.functional.pad(flat, (0, HS[key].shape[1] - flat.shape[1])))
                    HS["ctr"].add_(1)
                if sel and VT_CHECK is not None:      # the same projection again on the 64x64-per-wave kernel; count elements that differ by more than rounding
                    chk = torch.empty_like(vt)
                    ops.gemm(xf.w_v1f, h0r, chk, ln=(2, L["lnst0"], xf.c_v1, xf.d_v1, LN_EPS), guard=gd, tile=(0, 0, -1))
                    a, b = vt.float(), chk.float()
                    bad = (a - b).abs() > 0.01 + 0.01 * b.abs()
                    VT_CHECK.add_(bad.sum())
                    if VT_MASK is not None and VT_MASK.shape == bad.shape:
                        VT_MASK.add_(bad.to(VT_MASK.dtype))
                        VT_DIFF.copy_(torch.where(bad, a - b, VT_DIFF))
            elif fold:
                _pair(((h0r, xf.w_qk1f, qk), dict(ln=(1, L["lnst0"], xf.c_qk1, xf.d_qk1, LN_EPS), guard=gd)),
                      ((xf.w_v1f, h0r, vt), dict(ln=(2, L["lnst0"], xf.c_v1, xf.d_v1, LN_EPS), guard=gd)))
            else:
                ops.layernorm(h0, *xf.ln["norm1"], L["ln"])
                _pair(((L["ln"], xf.w_qk1, qk), dict(workspace=ws)), ((xf.w_v1, L["ln"], vt), dict(workspace=wp)))
            qk3 = qk.view(B, hw, 2 * C)
            self._attention(qk3[:, :, :C], qk3[:, :, C:], vt.view(C, B, hw).permute(1, 0, 2), att.view(B, hw, C), heads, scale)
            plans = () if harvest is None else (tuple(harvest) if isinstance(harvest, (list, tuple)) else (harvest,))
            # feature :263.  A plan whose sample order is the context's slot order (HarvestPlan.direct) gets the feature as the second,
            # fp16 output of this GEMM — written straight into the context buffer, which then also serves as the raw copy of h1 that the
            # folded query projections read; any other plan is served by strided copies.
            direct = None
            if len(plans) == 1 and plans[0].is_direct(B, plans[0].slots_per_row or self.R):
                c = plans[0].ctx[xf.spec.feature_key]
                direct = c if c.dim() == 2 else c.view(-1, C)
                assert direct.shape[0] == M, (direct.shape, M)
            kw1 = prod(h1, L["ln4"], L["lnst1"])
            if direct is not None:
                kw1["out2"] = direct
                h1r = direct
            ops.gemm(att, xf.w_o1, h1, bias=xf.b_o1, res1=h0, workspace=ws, **kw1)
            if harvest is not None:
                h1b = h1.view(B, hw, C)
                for plan in plans:
                    ctx = plan.ctx[xf.spec.feature_key]
                    c2d = ctx if ctx.dim() == 2 else ctx.view(-1, C)
                    if direct is None:
                        Rp = plan.slots_per_row or (ctx.shape[1] // hw)
                        for src, step, row, slot, cnt in plan.ops:
                            t0 = plan.flat_slot(row, slot, Rp) * hw
                            dst = c2d[t0:t0 + cnt * hw].view(cnt, hw, C)
                            ops.copy_rows(dst, h1b[plan.src_offset + src:].as_strided((cnt, hw, C), (step * hw * C, C, 1)))
                    if plan.kv is not None:            # attn3 K / V^T of the finished context (attention.py:215-223)
                        ki, vti = plan.kv[xf.spec.feature_key]
                        _pair(((c2d, xf.w_k3, ki), dict(workspace=ws)), ((xf.w_v3, c2d, vti), dict(workspace=wp)))   # VT[C, slots*hw]
                if stop_after_harvest:
                    return
            # --- query projections of the text cross-attention :266-277 (norm2) and the image cross-attention :281-291 (norm4): they
            # share statistics, and both in one launch
            if fold:
                pass
            elif consume:
                ops.layernorm(h1, *xf.ln["norm2"], L["ln"], 1e-5, *xf.ln["norm4"], L["ln4"])
            else:
                ops.layernorm(h1, *xf.ln["norm2"], L["ln"])
            if consume and fold:
                _pair(((h1r, xf.w_q2f, q2), dict(ln=(1, L["lnst1"], xf.c_q2, xf.d_q2, LN_EPS), guard=gd)),
                      ((h1r, xf.w_q3f, q3buf), dict(ln=(1, L["lnst1"], xf.c_q3, xf.d_q3, LN_EPS), guard=gd)))
            elif consume:
                _pair(((L["ln"], xf.w_q2, q2), dict(workspace=ws)), ((L["ln4"], xf.w_q3, q3buf), dict(workspace=wp)))
            elif fold:
                ops.gemm(h1r, xf.w_q2f, L["q"], ln=(1, L["lnst1"], xf.c_q2, xf.d_q2, LN_EPS), guard=gd)
            else:
                ops.gemm(L["ln"], xf.w_q2, L["q"], workspace=ws)
            if phase == "front":
                return
        kt3, vtt3 = self._text_kv(xf, lvl, text_cache)
        if consume:
            # the text branch (q2 -> attn2, :266-276) and the image branch (q3 -> attn3, :281-290) write the two halves
            # of one [M, 2C] buffer; their out-projections, biases and both residual adds (:277,291-293) are one GEMM
            att23 = L["att23"]
            a2v, a3v = att23[:, :C].unflatten(0, (B, hw)), att23[:, C:].unflatten(0, (B, hw))
            shared = self._shared_back      # cfg_shared_head, first transformer: ONE query tensor for the three CFG samples
            if shared:
                q2q = q2[:hw].view(1, hw, C).expand(2, hw, C)
                q3q = q3buf[:hw].view(1, hw, C).expand(2, hw, C)
                kt3, vtt3, a2o, a3o = kt3[::2], vtt3[::2], a2v[::2], a3v[:2]       # text rows (uncond, text) -> samples 0 and 2
            else:
                q2q, q3q, a2o, a3o = q2.view(B, hw, C), q3buf.view(B, hw, C), a2v, a3v
            # text + image attention as one launch when the image attention is one fp16 launch itself (else the text attention runs
            # beside the context projections on the side stream)
            paired = (ATTN_PAIR and self.attn3_share is not None and not self.ctx_short and not shared
                      and not (self.fp8_attention and C == heads * 40))
            forked = False if paired else self._fork()
            if paired:
                pass
            elif forked:
                with torch.cuda.stream(self.side):
                    ops.attention(q2q, kt3, vtt3, a2o, heads, scale, nk=S)
            else:
                ops.attention(q2q, kt3, vtt3, a2o, heads, scale, nk=S)
            ctx = self.ctx[xf.spec.feature_key]
            ns, rows = self.ctx_short, self.ctx_rows
            nk = self.R * hw
            c2d = ctx if ctx.dim() == 2 else ctx.view(-1, C)
            if self.kv_ext is not None:
                ki, vti = self.kv_ext[xf.spec.feature_key]
            else:
                ki, vti = L["ki"], L["vti"]
                _pair(((c2d, xf.w_k3, ki), dict(workspace=ws)), ((xf.w_v3, c2d, vti), dict(workspace=wp)))   # VT[C, slots*hw]
            # K / V^T rows: `ns` short ones (hw keys) in front of rows - ns long ones (R hw keys)
            ki3 = ki[ns * hw:].view(rows - ns, nk, C)
            vti3 = vti[:, ns * hw:].unflatten(1, (rows - ns, nk)).permute(1, 0, 2)
            short = None
            if ns:
                short = (ki[: ns * hw].view(ns, hw, C), vti[:, : ns * hw].unflatten(1, (ns, hw)).permute(1, 0, 2))
            q3 = q3q
            if paired:
                ops.attention_pair((q3, ki3, vti3, a3v, None), (q2q, kt3, vtt3, a2v, S), heads, scale)
            elif self.attn3_share is not None:    # one launch: batch b reads context row b (b < rows) or b - (B - rows)
                self._attention(q3, ki3, vti3, a3o, heads, scale, short=short)
            else:
                for q0, n, c0 in self.attn3_groups:
                    self._attention(q3[q0:q0 + n], ki3[c0:c0 + n], vti3[c0:c0 + n], a3v[q0:q0 + n], heads, scale)
            if forked:
                self._join()
            if shared:      # sample 1 = (uncond text, frames): its text attention is sample 0's, its image attention sample 2 shares
                ops.copy_rows(a2v[1:2], a2v[0:1])
                ops.copy_rows(a3v[2:3], a3v[1:2])
            h3 = L["h3"]
            ops.gemm(att23, xf.w_o23, h3, bias=xf.b_o23, res1=h1, res2=h1, workspace=ws,  # (a2 + h) + (a3 + h)
                     **({} if ff1 else prod(h3, h3raw, L["lnst3"])))
        else:
            ops.attention(L["q"].view(B, hw, C), kt3, vtt3, att.view(B, hw, C), heads, scale, nk=S)
            h3 = L["h2"]
            ops.gemm(att, xf.w_o2, h3, bias=xf.b_o2, res1=h1, workspace=ws, **({} if ff1 else prod(h3, h3raw, L["lnst3"])))   # :277,295
        # --- feed-forward :298-300
        h4, w_out = L["h4"], xf.w_out
        if ff1 and M <= FF_SPLIT_MAX_TOKENS:      # small launch: two workgroups per 128 tokens, partial sums side by side
            # L["att23"] as scratch: its only reader, the w_o23 / w_o2 GEMM above, is behind us on this stream, and the next writer (the
            # cross-attentions of the next block at this level) comes after proj_out below has consumed the partial sums (advisor r5).
            # The two halves are rounded to fp16 separately and re-added by proj_out's K = 2C contraction: one more rounding than the
            # unsplit kernel's single fp16 output, covered by the full-depth parity tests on the default (split) schedule.
            h4, w_out = L["att23"], xf.w_out2
            ops.ff_fused(h3, xf.ff_pack, xf.b_ff2, h4, LN_EPS, split=True)
        elif ff1:    # one launch: LayerNorm in registers, GEGLU intermediate never materialised (h4 is fp16: it only feeds proj_out)
            ops.ff_fused(h3, xf.ff_pack, xf.b_ff2, L["h4"], LN_EPS)
        elif merge:      # GEGLU into columns [0, 4C) of the buffer whose columns [4C, 5C) hold h3's raw copy; ff.net.2 happens inside proj_out's GEMM
            ops.gemm(h3raw, xf.w_ff1f, L["ffo"][:M, :4 * C], epilogue=ops.EPI_GEGLU, ln=(1, L["lnst3"], xf.c_ff1, xf.d_ff1, LN_EPS), guard=gd)
            h4, w_out = L["ffo"][:M], xf.w_ffo
        elif fold:
            ops.gemm(raw(h3, L["ln"]), xf.w_ff1f, L["ffi"], epilogue=ops.EPI_GEGLU, ln=(1, L["lnst3"], xf.c_ff1, xf.d_ff1, LN_EPS), guard=gd)
        else:
            ops.layernorm(h3, *xf.ln["norm3"], L["ln"])
            ops.gemm(L["ln"], xf.w_ff1, L["ffi"], bias=xf.b_ff1, epilogue=ops.EPI_GEGLU, workspace=ws)
        if not ff1 and not merge:
            ops.gemm(L["ffi"], xf.w_ff2, L["h4"], bias=xf.b_ff2, res1=h3, workspace=ws)   # fp16: only feeds proj_out
        kwo = dict(bias=xf.b_ffo if merge else xf.b_out, res1=x, workspace=ws)            # proj_out + residual :121-123
        site = xf.spec.prefix + ".proj_out"
        so = self._stats_for(site, lvl, C, lambda buf: ops.gemm_stats_rows(h4, w_out, out, stats=(buf, hw), **kwo))
        ops.gemm(h4, w_out, out, stats=None if so is None else (so, hw), **kwo)
        if so is not None:
            self._publish(out, site, C)
        else:
            self._pstats.pop(out.data_ptr(), None)

    def _sampler_conv(self, prefix: str, h: torch.Tensor, out: torch.Tensor, lvl: int, out_lvl: int, down: bool):
        """Downsample2D (3x3 stride 2) / Upsample2D (nearest 2x + 3x3): the fp32 stream tensor is cast into the
        zero-bordered fp16 conv input first."""
        w, b = self.samplers[prefix]
        pbuf = self.padded[(lvl, h.shape[1])]
        ops.pad_cast(self._img(h, lvl), pbuf)
        kw = dict(stride=2 if down else 1, upsample2x=not down, bias=b, workspace=self.ws_split, x_padded=True)
        n_out = out.shape[1]
        st = self._stats_for(prefix, out_lvl, n_out, lambda buf: ops.conv3x3_stats_rows(pbuf, w, self._img(out, out_lvl), stats=buf, **kw))
        ops.conv3x3(pbuf, w, self._img(out, out_lvl), stats=st, **kw)
        if st is not None:
            self._publish(out, prefix, n_out)
        else:
            self._pstats.pop(out.data_ptr(), None)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, harvest_slot: Optional[int] = None, consume: bool = False, harvest: Optional[HarvestPlan] = None,
                harvest_only: bool = False, text_cache: bool = False,
                side: Optional[torch.cuda.Stream] = None) -> Optional[torch.Tensor]:
        """One UNet call on the static inputs (self.x_in fp32 NCHW, self.t_in fp32 [B], self.text_in fp16).

        harvest_slot=r : reference pass — no image context; sample b's features go to slot r of row b of self.ctx.
        harvest=plan   : reference pass writing into another engine's context buffers (HarvestPlan, or a list of them:
                         one per slice of the batch, HarvestPlan.src_offset).
        harvest_only   : stop right after the last feature is harvested (the rest of the pass — text attention and FF
                         of the last block, conv_out — only produces an epsilon the loop discards, pipeline.py:433-435);
                         returns None.
        consume=True   : main pass — attn3 cross-attends to self.ctx.
        text_cache     : use the attn2 K/V projections stored by cache_text_kv().
        side           : a second stream for the pass's independent branches (1x1 shortcuts beside conv1/norm2, the text
                         attention beside the image attention); in a capture they become parallel graph branches.
        Returns self.eps_out (fp32 NCHW)."""
        if harvest_slot is not None:
            if harvest is not None or self.ctx_rows != self.B:
                raise ValueError("harvest_slot needs one context row per sample and no explicit plan")
            if not self.R:
                raise ValueError("engine was built without context buffers (n_ref=0)")
            if self.ctx_short:
                raise ValueError("harvest_slot: this engine's context rows have different slot counts (ctx_short); pass a HarvestPlan")
            harvest = HarvestPlan(self.ctx, [(b, 0, b, harvest_slot, 1) for b in range(self.B)])
        if harvest is not None and consume:
            raise ValueError("a pass either harvests features or consumes them")
        if consume and not self.R:
            raise ValueError("engine was built without context buffers (n_ref=0)")
        if harvest_only and harvest is None:
            raise ValueError("harvest_only needs a harvest plan")
        self.side = side
        last_xf = [a for blk in self.arch.up for a in blk.attns if a is not None][-1].prefix if harvest_only else None
        tk = dict(text_cache=text_cache)
        arch, text, skips = self.arch, self.text_in, self.skips
        # --- time embedding :392-398, then all 22 time_emb_proj(silu(emb)) in one GEMV bundle.  emb is only ever consumed through
        # silu (resnet.py time_emb_proj), so the second linear writes silu(emb) once instead of every wave of the bundle redoing it
        if self.time_table is not None and self.time_table_on:       # the loop's timesteps were tabulated at prepare() time: one row lookup instead of the chain
            ops.lookup_rows(self.t_in, self.time_table[0], self.time_table[1], self.tproj)
        else:
            self._time_chain(self.t_in, self.temb0, self.temb1, self.temb2, self.tproj)
        shared = self.cfg_shared_head and consume
        if shared and not self._head_checked and not torch.cuda.is_current_stream_capturing():
            # the caller's guarantee, verified on the first eager pass (the sampler's warm-up before capture): one latent, one timestep,
            # text rows 0 and 1 alike.  (One host synchronisation, once per engine.)
            self._head_checked = True
            if not (torch.equal(self.x_in[0], self.x_in[1]) and torch.equal(self.x_in[0], self.x_in[2])
                    and bool((self.t_in == self.t_in[0]).all()) and torch.equal(self.text_in[0], self.text_in[1])):
                raise ValueError("cfg_shared_head: the three samples must be the same latent at the same timestep with text rows "
                                 "[uncond, uncond, text] (the CFG main pass of one story frame, pipeline.py:448-453)")
        if shared:
            # CFG main pass of one story frame: the three samples are ONE computation up to the first cross-attention — conv_in :411,
            # the first ResnetBlock2D and the first transformer's front half run on sample 0 only, their results are copied to the
            # other two, and the cross-attentions of that transformer run once per DISTINCT (query, context) pair
            blk0 = arch.down[0]
            r0, xf0 = self.resnets[blk0.resnets[0].prefix], self.xfs[blk0.attns[0].prefix]
            # (conv_in at batch 3: a 0.2 GFLOP launch costs the same at either batch, and the skip tensor it writes is needed per sample)
            ops.conv_in(self.x_in, self.w_conv_in, self.b_conv_in, self._img(skips[0], 0))
            saved = self._enter_head(1)
            try:
                sk0 = skips[0][: self.hw[0]]
                self._resnet(r0, sk0, self.lv[0]["r"], 0, defer_out=True)
                self._transformer(xf0, self.lv[0]["r"], None, 0, text, None, True, phase="front", **tk)
            finally:
                self._leave_head(saved)
            for t2d in (self.lv[0]["r"], self.lv[0]["h1"]):                # proj_out's residual, the attentions' residual
                self._spread(t2d, 1, 2)
            self._shared_back = True
            try:
                self._transformer(xf0, self.lv[0]["r"], skips[1], 0, text, None, True, phase="back", **tk)
            finally:
                self._shared_back = False
            h, lvl, si = skips[1], 0, 2
        else:
            # --- conv_in :411
            ops.conv_in(self.x_in, self.w_conv_in, self.b_conv_in, self._img(skips[0], 0))
            h, lvl, si = skips[0], 0, 1
        # --- down :417-433 — every layer output is a skip tensor, so it is written straight into its skip buffer
        for bi, blk in enumerate(arch.down):
            for j, r in enumerate(blk.resnets):
                if shared and bi == 0 and j == 0:
                    continue                  # done above
                xf, out = blk.attns[j], skips[si]
                if xf is None:
                    self._resnet(self.resnets[r.prefix], h, out, lvl)
                else:
                    rt = self.lv[lvl]["r"]
                    self._resnet(self.resnets[r.prefix], h, rt, lvl, defer_out=True)
                    self._transformer(self.xfs[xf.prefix], rt, out, lvl, text, harvest, consume, **tk)
                h, si = out, si + 1
            if blk.sampler_prefix:
                out = skips[si]
                self._sampler_conv(blk.sampler_prefix, h, out, lvl, lvl + 1, down=True)
                h, lvl, si = out, lvl + 1, si + 1
        assert si == len(skips)
        # --- mid :436-445
        L = self.lv[lvl]
        m0, m1 = arch.mid.resnets
        self._resnet(self.resnets[m0.prefix], h, L["r"], lvl, defer_out=True)
        self._transformer(self.xfs[arch.mid.attns[0].prefix], L["r"], L["t_out"], lvl, text, harvest, consume, **tk)
        blk0 = arch.up[0]
        k = 0                                                                              # index of the up-path resnet
        cat = self.up_cats[k]
        self._resnet(self.resnets[m1.prefix], L["t_out"], cat[:, : blk0.resnets[0].cin - blk0.skip_channels[0]], lvl)
        # --- up :448-475 — `cat` = [h | skip]: h was written by the previous layer, the skip by the down path
        for bi, blk in enumerate(arch.up):
            L = self.lv[lvl]
            nres = len(blk.resnets)
            for j, r in enumerate(blk.resnets):
                si -= 1
                assert cat.shape[1] == r.cin and skips[si].data_ptr() == cat[:, r.cin - blk.skip_channels[j]:].data_ptr()
                if j + 1 < nres:
                    nxt = self.up_cats[k + 1]
                    out = nxt[:, : blk.resnets[j + 1].cin - blk.skip_channels[j + 1]]
                else:
                    nxt, out = None, L["t_out"]
                xf = blk.attns[j]
                if xf is None:
                    self._resnet(self.resnets[r.prefix], cat, out, lvl)
                else:
                    self._resnet(self.resnets[r.prefix], cat, L["r"], lvl, defer_out=True)
                    if xf.prefix == last_xf:
                        self._transformer(self.xfs[xf.prefix], L["r"], None, lvl, text, harvest, consume, stop_after_harvest=True)
                        return None
                    self._transformer(self.xfs[xf.prefix], L["r"], out, lvl, text, harvest, consume, **tk)
                k += 1
                if nxt is not None:
                    cat = nxt
                h = out
            if blk.sampler_prefix:                                                        # Upsample2D :656-658,730-732
                nblk = arch.up[bi + 1]
                cat = self.up_cats[k]
                c_h = nblk.resnets[0].cin - nblk.skip_channels[0]
                self._sampler_conv(blk.sampler_prefix, h, cat[:, :c_h], lvl, lvl - 1, down=False)
                lvl -= 1
        assert si == 0 and lvl == 0
        # --- out :477-480
        L = self.lv[0]
        ops.groupnorm(h.unflatten(0, (self.B, self.hw[0])), *self.gn_out, L["gn"].unflatten(0, (self.B, self.hw[0])),
                      self.groups, self.eps, True, self.ws_gn, pstats=self._pstats_of(h))
        ops.conv_out(self._img(L["gn"], 0), self.w_conv_out, self.b_conv_out, self.eps_out)
        return self.eps_out

    def check_ln_guard(self, clear: bool = True) -> int:
        """Flags raised by the folded LayerNorms since the last check (0 = both assumptions held; synchronises on a 4-byte copy).
        Raises when one did not: the pass that set it must be rerun with engine.LN_FOLD = False (the LayerNorm launch on the fp32
        stream, as the reference computes it)."""
        flags = int(self.ln_guard.item())
        if clear and flags:
            self.ln_guard.zero_()
        if flags:
            why = []
            if flags & ops.LN_GUARD_RANGE:
                why.append("a stream tensor reached |x| >= 65504 (its fp16 copy was saturated)")
            if flags & ops.LN_GUARD_OFFSET:
                why.append(f"a token had |mean| / sigma > {ops.LN_GUARD_RATIO:g} (fp16 rounding of the raw copy no longer negligible)")
            raise FloatingPointError("LayerNorm fold outside its range: " + "; ".join(why) + " — rerun with storygen_amd.engine.LN_FOLD = False")
        return flags

    # ------------------------------------------------------------------------------------------ convenience
    def _time_chain(self, t, e0, e1, e2, out):
        """Timesteps -> TimestepEmbedding -> all time_emb_proj(silu(emb)) rows (unet_2d_condition.py:392-398, ResnetBlock2D)."""
        ops.timestep_embed(t, self.freqs, e0, self.cfg["flip_sin_to_cos"])
        ops.linear_rows(e0, self.w_t1, self.b_t1, e1, act_out=True)
        ops.linear_rows(e1, self.w_t2, self.b_t2, e2, act_out=True)
        ops.linear_rows(e2, self.w_temb, self.b_temb, out)

    def build_time_table(self, timesteps, capacity: int = 256) -> bool:
        """Tabulate the time-embedding chain for every distinct value in `timesteps` (a sampler knows them all at prepare() time:
        pipeline.py:410-415).  forward() then reads the rows of self.t_in with ops.lookup_rows — one launch of a few microseconds instead
        of four (the last one streams the 22 stacked time_emb_proj matrices, ~45 MB, per UNet call) — and gets NaN rows, not stale ones,
        for a timestep that is not in the table.  Rows are computed by the same kernels four at a time, exactly as forward() would.
        The table lives in buffers of `capacity` rows (unused keys are NaN, which matches nothing) that later calls refill IN PLACE, so a
        captured graph stays valid; returns True when the buffers were (re)allocated — or dropped — i.e. when graphs captured before
        the call must be captured again.  Pass None to drop the table (training, set_inputs() with arbitrary timesteps)."""
        self.time_table_on = True
        if timesteps is None:
            had = self.time_table is not None
            self.time_table = None
            self._time_keys = None
            return had
        keys = sorted({float(t) for t in timesteps})
        if not keys:
            raise ValueError("build_time_table: no timesteps")
        tag = (keys, getattr(self.wts, "version", 0))
        if self.time_table is not None and getattr(self, "_time_keys", None) == tag:
            return False                          # the same schedule on the same weights as the last call (a pipeline called again): the rows are already there
        F32, dev, T = torch.float32, self.dev, len(keys)
        fresh = self.time_table is None or self.time_table[0].numel() < T
        if fresh:
            cap = max(int(capacity), T)
            self.time_table = (torch.empty(cap, dtype=F32, device=dev), torch.empty(cap, self.temb_total, dtype=F32, device=dev))
        tk, tab = self.time_table
        tk.fill_(float("nan"))
        tk[:T].copy_(torch.tensor(keys, dtype=F32))
        e0 = torch.empty(4, self.temb0.shape[1], dtype=F32, device=dev)
        e1, e2 = (torch.empty(4, self.arch.temb_dim, dtype=F32, device=dev) for _ in range(2))
        for i in range(0, T, 4):
            n = min(4, T - i)
            self._time_chain(tk[i:i + n], e0[:n], e1[:n], e2[:n], tab[i:i + n])
        self._time_keys = tag
        return fresh

    def set_inputs(self, sample: torch.Tensor, timestep, text: torch.Tensor):
        self.x_in.copy_(sample.to(self.dev, torch.float32))
        # arbitrary timesteps: the chain itself, not a sampler's table.  The table's buffers are kept (a hipGraph captured while it was
        # in use still reads them; build_time_table() switches it back on and refills them in place)
        self.time_table_on = False
        t = timestep if torch.is_tensor(timestep) else torch.tensor([float(timestep)])
        self.t_in.copy_(t.to(self.dev, torch.float32).reshape(-1).expand(self.B))
        self.text_in.copy_(text.to(self.dev, F16))

    def features(self, slot: int = 0) -> Dict[str, torch.Tensor]:  # fp16 views
        """The 16 harvested [B, HW, C] features of slot r, as views into the context buffers."""
        if self.ctx_short:
            raise ValueError("features(): this engine's context rows have different slot counts (ctx_short)")
        out = {}
        for k, (n, _) in feature_shapes(self.arch, self.H, self.W).items():
            out[k] = self.ctx[k][:, slot * n:(slot + 1) * n, :]
        return out
