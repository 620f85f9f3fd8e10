"""Stage-2 training step (BASELINE config 4, /root/reference/train_StorySalon_stage2.py:291-327) on the HIP kernels:
loss and the 80 attn3 gradients of one step.

STATUS: validated on MI355X in round 2 (tests/test_backward_gpu.py::test_training_step_vs_oracle: loss and all 80 gradients
against oracle.storygen_oracle.train_step, <= 1e-2, with the loss scaling of UNetTrainer.backward_main); BASELINE config 4
measures 10.3 it/s at bs 4 (profiles/r02a_train_step_bench.json).  A correctness-first assembly of
storygen_amd/train_blocks.py that follows oracle/storygen_backward.py (unet_forward_saving / unet_backward) record by
record: per-call allocations, torch.cat for the skip concatenations, no hipGraph yet.  The optimizer step, GradScaler and DDP
all-reduce of the 49.6 M attn3 parameters (train_StorySalon_stage2.py:187-222,327-332) are the caller's stock PyTorch.

Structure
  * reference passes (no gradient reaches a trainable parameter through them: attn3 is not evaluated there): the inference
    UNetEngine, one call per prior frame used, harvesting straight into the context buffers;
  * main pass forward keeping each module's input, then the tape walked backwards: dgrad everywhere from conv_out down to
    the first transformer block, weight gradients for the 16 attn3 modules only.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .arch import UNetArch, XfSpec
from .engine import EngineWeights, UNetEngine
from .repack import conv1x1_nk, conv3x3_krsc, conv_in_kn
from .scheduler import DDIMSchedule
from .train_blocks import F16, F32, ResnetBlockTrain, TransformerBlockTrain, _cast16, _e, _t


def _nonfinite_flag(tensors: List[torch.Tensor]) -> torch.Tensor:
    """fp32 [1] on the tensors' device: > 0 iff any entry of any tensor is inf / nan.  One multi-tensor launch (the kernel torch's
    GradScaler uses, with inverse scale 1: the tensors are left as they are), a few reductions without it."""
    dev = tensors[0].device
    fn = getattr(torch, "_amp_foreach_non_finite_check_and_unscale_", None)
    if fn is not None and dev.type == "cuda" and all(t.dtype == F32 for t in tensors):
        found = torch.zeros(1, dtype=F32, device=dev)
        fn(tensors, found, torch.ones(1, dtype=F32, device=dev))
        return found
    return torch.stack([(~torch.isfinite(v)).any() for v in tensors]).sum().to(F32).reshape(1)


SPLITK_WORKSPACE = True      # development switch (bench.py --train-no-splitk-workspace): the training classes without split-K scratch, as in rounds 2 - 5


class Transformer2DTrain:
    """Transformer2DModel (model/attention.py:26-128): GroupNorm(eps 1e-6) -> 1x1 proj_in -> block -> 1x1 proj_out -> + x."""

    def __init__(self, sd: Dict[str, torch.Tensor], spec: XfSpec, groups: int, device, trainable: str = "attn3"):
        self.dev, self.groups, self.spec, p = torch.device(device), groups, spec, spec.prefix
        g = lambda k: sd[f"{p}.{k}"].detach().to(self.dev, F16).contiguous()     # noqa: E731
        self.ng, self.nb = g("norm.weight"), g("norm.bias")
        self.w_in, self.b_in = conv1x1_nk(g("proj_in.weight")), g("proj_in.bias")
        self.w_out, self.b_out = conv1x1_nk(g("proj_out.weight")), g("proj_out.bias")
        self.w_in_t, self.w_out_t = _t(self.w_in), _t(self.w_out)
        self.blk = TransformerBlockTrain(sd, f"{p}.transformer_blocks.0", spec.heads, device, trainable)
        self.saved = None

    def forward(self, x: torch.Tensor, text16: torch.Tensor, ctx16: Optional[torch.Tensor], B: int) -> torch.Tensor:
        M, C, dev = x.shape[0], x.shape[1], self.dev
        hw = M // B
        ws = _e(ops.groupnorm_workspace_bytes(B, self.groups), dev=dev, dtype=torch.uint8)
        gn = _e(M, C, dev=dev)
        ops.groupnorm(x.view(B, hw, C), self.ng, self.nb, gn.view(B, hw, C), self.groups, 1e-6, False, ws)
        h0 = _e(M, C, dev=dev, dtype=F32)
        ops.gemm(gn, self.w_in, h0, bias=self.b_in)
        hb = self.blk.forward(h0, text16, ctx16, B)
        out = _e(M, C, dev=dev, dtype=F32)
        ops.gemm(_cast16(hb), self.w_out, out, bias=self.b_out, res1=x)
        self.saved = dict(x=x, B=B)
        return out

    def backward(self, dout: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        x, B = self.saved["x"], self.saved["B"]
        M, C, dev = x.shape[0], x.shape[1], self.dev
        hw = M // B
        dhb = _e(M, C, dev=dev, dtype=F32)
        ops.gemm(_cast16(dout), self.w_out_t, dhb)
        dh0, g = self.blk.backward(dhb)
        dgn = _e(M, C, dev=dev)
        ops.gemm(_cast16(dh0), self.w_in_t, dgn)
        ws = _e(ops.groupnorm_bwd_workspace_bytes(B, self.groups), dev=dev, dtype=torch.uint8)
        dx = _e(M, C, dev=dev, dtype=F32)
        ops.groupnorm_bwd(x.view(B, hw, C), dgn.view(B, hw, C), self.ng, self.nb, dx.view(B, hw, C), self.groups, 1e-6, False, ws,
                          res=dout.view(B, hw, C))
        p = f"{self.spec.prefix}.transformer_blocks.0.{self.blk.trainable}"
        return dx, {f"{p}.{k}": v for k, v in g.items()}


class UNetTrainer:
    def __init__(self, arch: UNetArch, state_dict: Dict[str, torch.Tensor], device, batch: int, height: int, width: int,
                 n_ref: int = 3, seq_len: int = 77, ref_engine=None, weights: Optional[EngineWeights] = None, trainable: str = "attn3"):
        """trainable: "attn3" (stage 2 / COCO: train_step with 1-3 reference frames) or "attn1" (stage 1: train_step(use_refs=()),
        no reference pass, no image context — train_StorySalon_stage1.py:175-179,288).
        ref_engine: the object that runs the reference passes (set_inputs / forward(harvest_slot=) / .ctx); default = an
        inference UNetEngine on the same weights (tests inject a CPU stand-in to exercise the host logic without a GPU)."""
        self.arch, self.dev, self.cfg = arch, torch.device(device), arch.config
        self.B, self.H, self.W, self.R = batch, height, width, n_ref
        sd = state_dict
        self.wts = weights if weights is not None else EngineWeights(arch, sd, device)
        if ref_engine is None and n_ref == 0:
            ref_engine = object()                                                     # stage 1: no reference pass ever runs
        # Default (round 5): the reference passes of a step are ONE batched UNet call per number of frames used — batch B * n_used,
        # samples ordered like the context buffer (row b, frame slot j) so that the features are written in place, and stopped after
        # the last harvest point (the epsilon a reference pass would go on to produce is discarded, :309-314) — the engine the
        # sampler's reference pass uses.  An injected ref_engine keeps the one-call-per-frame protocol (set_inputs / forward(harvest_slot=)).
        self.ref = ref_engine
        self._ref_batched: Dict[int, tuple] = {}                                      # n_used -> (engine, context buffers, harvest plan)
        self.seq_len = seq_len
        self.groups, self.eps = self.cfg["norm_num_groups"], self.cfg["norm_eps"]
        g16 = lambda k: sd[k].detach().to(self.dev, F16).contiguous()             # noqa: E731
        self.resnets = {r.prefix: ResnetBlockTrain(sd, r.prefix, self.groups, self.eps, device) for r in arch.resnets}
        self.temb_proj = {r.prefix: (g16(f"{r.prefix}.time_emb_proj.weight"), g16(f"{r.prefix}.time_emb_proj.bias")) for r in arch.resnets}
        self.trainable = trainable
        self.ref_levels = "stage2"      # noise level of reference frame i: ref_t * (3 - i) (stage 2, :311) or ref_t for all ("coco", train_COCO.py:303)
        self.xfs = {a.prefix: Transformer2DTrain(sd, a, self.groups, device, trainable)
                    for blk in arch.down + [arch.mid] + arch.up for a in blk.attns if a is not None}
        self.samplers = {}
        for blk in arch.down + arch.up:
            if blk.sampler_prefix:
                w = sd[f"{blk.sampler_prefix}.weight"].to(self.dev, F16)
                self.samplers[blk.sampler_prefix] = (conv3x3_krsc(w), g16(f"{blk.sampler_prefix}.bias"),
                                                     conv3x3_krsc(w.flip(2, 3).transpose(0, 1).contiguous()))
        self.gn_out = (g16("conv_norm_out.weight"), g16("conv_norm_out.bias"))
        w_out = sd["conv_out.weight"].to(self.dev, F16)
        self.w_conv_out, self.b_conv_out = conv3x3_krsc(w_out), g16("conv_out.bias")
        self.w_conv_out_d = conv_in_kn(w_out.flip(2, 3).transpose(0, 1).contiguous())      # dgrad of conv_out = a 4 -> C conv_in
        self.zero_bias = torch.zeros(w_out.shape[1], dtype=F16, device=self.dev)
        self.schedule = DDIMSchedule()
        # Loss scaling for the fp16 gradient operands (the reference trains under accelerate's fp16 GradScaler,
        # train_StorySalon_stage2.py:138-141,328): the MSE gradient of a mean over B*4*H*W elements is ~1e-3 and the
        # attention-score gradients dS = P (dP - delta) reach 1e-7..1e-8 — below fp16's subnormal range — so the backward
        # walk runs on s * d_pred and the fp32 weight gradients are divided by s.  "auto": s = the power of two that brings
        # max|d_pred| to 2^8 (one host read per step), retried 16x smaller if a gradient comes out non-finite.
        self.grad_scale = "auto"
        self.last_grad_scale = 1.0
        self._graphs: Dict[tuple, dict] = {}       # train_step_graph: use_refs -> captured hipGraph + static inputs / outputs
        self.check_finite = True                   # train_step_graph: one host read of the gradients' finiteness per step
        # GradScaler semantics for the graph path (train_StorySalon_stage2.py:138-141,328 trains under accelerate's fp16 GradScaler): a
        # step whose gradients stay non-finite after the scale has been lowered is SKIPPED (last_step_skipped; the caller must not
        # apply it) instead of aborting the run; every `scale_growth_interval` good steps the scale is doubled again
        self.last_step_skipped = False
        self.scale_growth_interval = 2000
        self._good_steps = 0
        self._alphas_dev = self.schedule.alphas_cumprod.to(self.dev, F32)
        # split-K scratch shared by every gemm / conv3x3 of the step (one stream: ops.default_workspace).  Largest user: 16 slices of
        # a [B*H*W/16, 1280] fp32 tile image at the 16x16 level; the engine's reference passes own theirs.
        self.ws_split = ops.new_workspace(64 << 20, self.dev) if self.dev.type == "cuda" and SPLITK_WORKSPACE else None

    def _batched_ref_engine(self, n_used: int):
        """(engine, context buffers, harvest plan) of the batched reference pass over n_used frames: built on first use, kept (a
        captured hipGraph holds the buffers' addresses).
        Memory (advisor r5): one engine per DISTINCT n_used the loop draws — train_StorySalon_stage2.py:306-313 draws 1, 2 or 3 — i.e. up
        to B (1 + 2 + 3) samples of reference-pass activations stay resident (SD-1.5 at a 64x64 latent: ~0.2 GB per sample, 4.8 GB at
        bs 4; the weights are shared).  That is the price of writing the features in place at every n_used; a caller short of memory
        can drop the engines (and the graphs captured on them) it no longer needs with release_ref_engines()."""
        st = self._ref_batched.get(n_used)
        if st is None:
            from .arch import feature_shapes
            from .engine import HarvestPlan
            B = self.B
            eng = UNetEngine(self.arch, None, self.dev, B * n_used, self.H, self.W, 0, self.seq_len, weights=self.wts)
            ctx = {k: torch.empty(B, n_used * n, c, dtype=F16, device=self.dev) for k, (n, c) in feature_shapes(self.arch, self.H, self.W).items()}
            plan = HarvestPlan(ctx, [(b * n_used, 1, b, 0, n_used) for b in range(B)], slots_per_row=n_used, direct=True)
            assert plan.is_direct(B * n_used, n_used)
            st = self._ref_batched[n_used] = (eng, ctx, plan)
        return st

    def release_ref_engines(self, keep=()):
        """Free the batched reference engines of every n_used not in `keep`, together with the captured training-step graphs, which hold
        their buffers' addresses (they are rebuilt / re-captured on next use)."""
        for n in [n for n in self._ref_batched if n not in keep]:
            del self._ref_batched[n]
        for attr in ("_graphs", "_step_graphs"):
            g = getattr(self, attr, None)
            if isinstance(g, dict):
                g.clear()

    # ------------------------------------------------------------------------------------------------ pieces
    def _add_noise(self, x: torch.Tensor, noise: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        """DDPMScheduler.add_noise with per-sample timesteps (train_StorySalon_stage2.py:303,311); elementwise plumbing."""
        a = self._alphas_dev[t.long()].view(-1, 1, 1, 1)
        return a.sqrt() * x + (1 - a).sqrt() * noise

    def _down(self, prefix: str, h: torch.Tensor, hh: int, ww: int) -> torch.Tensor:
        w, b, _ = self.samplers[prefix]
        C = h.shape[1]
        pad = torch.zeros(self.B, hh + 2, ww + 2, C, dtype=F16, device=self.dev)
        ops.pad_cast(h.view(self.B, hh, ww, C), pad)
        out = _e(self.B * (hh // 2) * (ww // 2), C, dev=self.dev, dtype=F32)
        ops.conv3x3(pad, w, out.view(self.B, hh // 2, ww // 2, C), stride=2, bias=b, x_padded=True)
        return out

    def _down_bwd(self, prefix: str, dout: torch.Tensor, ho: int, wo: int) -> torch.Tensor:
        C = dout.shape[1]
        pad = torch.zeros(self.B, 2 * ho + 2, 2 * wo + 2, C, dtype=F16, device=self.dev)
        ops.zero_stuff(dout.view(self.B, ho, wo, C), pad)
        dx = _e(self.B * 4 * ho * wo, C, dev=self.dev, dtype=F32)
        ops.conv3x3(pad, self.samplers[prefix][2], dx.view(self.B, 2 * ho, 2 * wo, C), x_padded=True)
        return dx

    def _up(self, prefix: str, h: torch.Tensor, hh: int, ww: int) -> torch.Tensor:
        w, b, _ = self.samplers[prefix]
        C = h.shape[1]
        pad = torch.zeros(self.B, hh + 2, ww + 2, C, dtype=F16, device=self.dev)
        ops.pad_cast(h.view(self.B, hh, ww, C), pad)
        out = _e(self.B * 4 * hh * ww, C, dev=self.dev, dtype=F32)
        ops.conv3x3(pad, w, out.view(self.B, 2 * hh, 2 * ww, C), upsample2x=True, bias=b, x_padded=True)
        return out

    def _up_bwd(self, prefix: str, dout: torch.Tensor, hh: int, ww: int) -> torch.Tensor:
        """dout at the upsampled resolution [B*2hh*2ww, C] -> gradient at [B*hh*ww, C]."""
        C = dout.shape[1]
        pad = torch.zeros(self.B, 2 * hh + 2, 2 * ww + 2, C, dtype=F16, device=self.dev)
        ops.pad_cast(dout.view(self.B, 2 * hh, 2 * ww, C), pad)
        du = _e(self.B * 4 * hh * ww, C, dev=self.dev, dtype=F32)
        ops.conv3x3(pad, self.samplers[prefix][2], du.view(self.B, 2 * hh, 2 * ww, C), x_padded=True)
        dx = _e(self.B * hh * ww, C, dev=self.dev, dtype=F32)
        ops.sum2x2(du.view(self.B, 2 * hh, 2 * ww, C), dx.view(self.B, hh, ww, C))
        return dx

    # ------------------------------------------------------------------------------------------------ the step
    def _stage_inputs(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The batch on the device in the dtypes the step consumes (host -> device copies happen only here)."""
        dev = self.dev
        f = lambda k: batch[k].to(dev, F32).contiguous()                             # noqa: E731
        out = dict(latents=f("latents"), noise=f("noise"), mask=f("mask"), timesteps=batch["timesteps"].to(dev).long(),
                   text=batch["text"].to(dev, F16).contiguous())
        if "ref_latents" in batch:                                                   # absent in stage-1 batches (no prior frames)
            out.update(ref_latents=f("ref_latents"), ref_noise=f("ref_noise"), prev_text=batch["prev_text"].to(dev, F16).contiguous())
        return out

    def _step_device(self, inp: Dict[str, torch.Tensor], use_refs) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """The whole step on device-resident inputs (no host <-> device traffic unless grad_scale is "auto"): reference passes,
        main pass, masked MSE, backward walk.  This is the function train_step_graph captures."""
        dev, B = self.dev, self.B
        t = inp["timesteps"]
        ref_t = torch.div(t, 10, rounding_mode="floor")                              # (timesteps / 10).long(), :297-300
        # ---- reference passes: features of the frames used, harvested into context slots 0..len(use_refs)-1
        n_used = len(use_refs)
        ctx16 = {} if n_used else None                                               # no frame: image_hidden_states=None (stage 1)
        levels = [ref_t * (3 - i) if self.ref_levels == "stage2" else ref_t for i in use_refs]                    # :309-314
        if self.ref is None and n_used:
            eng, ctx, plan = self._batched_ref_engine(n_used)
            xs = [self._add_noise(inp["ref_latents"][i], inp["ref_noise"], ti) for i, ti in zip(use_refs, levels)]
            # sample u = b * n_used + slot: the order of the context buffer's frame slots
            eng.x_in.copy_(torch.stack(xs, dim=1).flatten(0, 1))
            eng.t_in.copy_(torch.stack([ti.float() for ti in levels], dim=1).flatten())
            eng.text_in.copy_(torch.stack([inp["prev_text"][i] for i in use_refs], dim=1).flatten(0, 1))
            eng.forward(harvest=plan, harvest_only=True)
            for key, buf in ctx.items():                                             # [B, n_used*hw_k, C], written in place
                ctx16[key] = buf.view(-1, buf.shape[2])
        elif n_used:
            for slot, (i, ti) in enumerate(zip(use_refs, levels)):
                self.ref.set_inputs(self._add_noise(inp["ref_latents"][i], inp["ref_noise"], ti), ti.float(), inp["prev_text"][i])
                self.ref.forward(harvest_slot=slot)
            for key, buf in self.ref.ctx.items():                                    # [B, R*hw_k, C] -> the used slots, flattened
                n = buf.shape[1] // self.R
                ctx16[key] = buf[:, : n_used * n].reshape(B * n_used * n, buf.shape[2]).contiguous()
        text16 = inp["text"].reshape(B * inp["text"].shape[1], -1)
        noisy = self._add_noise(inp["latents"], inp["noise"], t).contiguous()        # :303
        # (the reference passes above run on the inference engine, which owns its split-K scratch; the training classes below borrow ours)
        with ops.default_workspace(self.ws_split):
            pred = self.forward_main(noisy, t, text16, ctx16)
            # ---- loss (:325) and its gradient
            d_pred, loss = torch.empty_like(pred), _e(1, dev=dev, dtype=F32)
            ops.mse_grad(pred, inp["noise"], inp["mask"], d_pred, loss)
            return loss, self.backward_main(d_pred)

    def train_step(self, batch: Dict[str, torch.Tensor], use_refs=(0, 1, 2)) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """`batch` as storygen_amd.synth.synthetic_train_batch / oracle.storygen_oracle.train_step.  Returns (loss [1] fp32 on
        the device, {parameter name: fp32 gradient}).  Eager: every kernel is launched from the host."""
        return self._step_device(self._stage_inputs(batch), tuple(use_refs))

    def train_step_graph(self, batch: Dict[str, torch.Tensor], use_refs=(0, 1, 2), _retry: int = 0) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """The same step replayed as ONE hipGraph (~3 000 kernel nodes): the first call with a given `use_refs` stages the batch
        into static device tensors, fixes the gradient scale from one eager step ("auto" needs a host read, a graph cannot), warms
        up and captures `_step_device` (its per-call allocations come from the graph's private pool, so replays reuse the same
        addresses); later calls copy the new batch into the static inputs and replay.  The returned loss / gradients are the
        graph's static outputs: consume them (optimizer step, all-reduce) before the next call.  A non-finite gradient drops ALL
        captured graphs (every `use_refs` variant must run at the same scale), lowers the scale 16x and re-captures — at most four
        times; if the gradients are still not finite the step is marked skipped (`last_step_skipped`, as torch's GradScaler skips
        optimizer.step) and the scale stays lowered.  After `scale_growth_interval` consecutive good steps the scale doubles."""
        use_refs = tuple(use_refs)
        if _retry == 0:
            self.last_step_skipped = False
        staged = self._stage_inputs(batch)
        st = self._graphs.get(use_refs)
        if st is None:
            static = {k: v.clone() for k, v in staged.items()}
            if self.grad_scale == "auto":                                             # settle the scale eagerly, then freeze it
                self._step_device(static, use_refs)
                self.grad_scale = float(self.last_grad_scale)
            torch.cuda.synchronize(self.dev)
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                self._step_device(static, use_refs)                                  # warm-up on the capture stream's allocator
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loss, grads = self._step_device(static, use_refs)
                bad = _nonfinite_flag(list(grads.values()))        # part of the graph (round 6): no per-step eager launches
            st = self._graphs[use_refs] = dict(graph=g, inputs=static, loss=loss, grads=grads, bad=bad)
        for k, v in staged.items():                  # (capture only records: the capturing call replays like every other)
            st["inputs"][k].copy_(v, non_blocking=True)
        st["graph"].replay()
        if self.check_finite:
            # ONE host read per step: "some gradient has a non-finite entry", computed by the replayed graph itself (rounds 2 - 5 ran
            # ~400 small eager launches per step here: isfinite / any per gradient tensor, stack, sum)
            if float(st["bad"]) > 0:
                # fp16 overflow of the scaled backward: drop the graph and retry with a 16x smaller scale — a BOUNDED number of
                # times (a NaN that comes from the batch or the weights never goes away: the reference's GradScaler would skip such
                # a step; here the caller gets an error instead of an endless re-capture)
                s_now = float(self.grad_scale) if self.grad_scale != "auto" else float(self.last_grad_scale)
                self._good_steps = 0
                if not bool(torch.isfinite(st["loss"]).all()):
                    # the LOSS is not finite: the batch (or the weights) is, not the loss scale — skip the step, keep the scale
                    self.last_step_skipped = True
                    return st["loss"], st["grads"]
                if _retry >= 4 or s_now / 16.0 < 2.0 ** -8:
                    # a NaN that comes from the batch or the weights never goes away: skip the step, as GradScaler does
                    self.last_step_skipped = True
                    return st["loss"], st["grads"]
                self._graphs.clear()                 # every variant was captured with the old scale baked in
                self.grad_scale = s_now / 16.0
                return self.train_step_graph(batch, use_refs, _retry=_retry + 1)
            self._good_steps += 1
            if self._good_steps >= self.scale_growth_interval and self.grad_scale != "auto":
                self._good_steps = 0
                self.grad_scale = float(self.grad_scale) * 2.0      # takes effect at the next call (re-capture)
                self._graphs.clear()
                return st["loss"].clone(), {k: v.clone() for k, v in st["grads"].items()}     # (the pool of the dropped graph may be reused)
        return st["loss"], st["grads"]

    def set_trainable_parameters(self, named_params: Dict[str, torch.Tensor]) -> None:
        """Refresh the device copies of the trainable parameters (full state-dict names, `...attn3.to_q.weight` etc.), in place."""
        for prefix, xf in self.xfs.items():
            p = f"{prefix}.transformer_blocks.0.{self.trainable}."
            sub = {k[len(p):]: v for k, v in named_params.items() if k.startswith(p)}
            if sub:
                xf.blk.set_trainable(sub)

    set_attn3_parameters = set_trainable_parameters          # round-1 name

    def forward_main(self, noisy: torch.Tensor, t: torch.Tensor, text16: torch.Tensor, ctx16: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Main pass (unet_2d_condition.py:338-485 in consume mode) keeping the tape for backward_main.  noisy fp32 NCHW
        [B,4,H,W]; t [B]; text16 fp16 [B*S, 768]; ctx16: feature key -> fp16 [B*R'*hw_k, C_k].  Returns epsilon fp32 NCHW."""
        dev, B, H, W, arch = self.dev, self.B, self.H, self.W, self.arch
        wts = self.wts
        boc0 = self.cfg["block_out_channels"][0]
        temb0, temb1, temb2 = (_e(B, n, dev=dev, dtype=F32) for n in (boc0, arch.temb_dim, arch.temb_dim))
        ops.timestep_embed(t.to(dev, F32).contiguous(), wts.freqs, temb0, self.cfg["flip_sin_to_cos"])
        ops.linear_rows(temb0, wts.w_t1, wts.b_t1, temb1, act_out=True)
        ops.linear_rows(temb1, wts.w_t2, wts.b_t2, temb2, act_out=True)     # emb is only consumed as silu(emb) (resnet.py time_emb_proj)
        tape: List[tuple] = []
        hh, ww = H, W

        def resnet(prefix, x):
            w, b = self.temb_proj[prefix]
            tp = ops.linear_rows(temb2, w, b, _e(B, w.shape[0], dev=dev, dtype=F32))
            tape.append(("resnet", prefix))
            return self.resnets[prefix].forward(x, tp, B, hh, ww)

        def xf(spec, x):
            tape.append(("xf", spec.prefix))
            return self.xfs[spec.prefix].forward(x, text16, None if ctx16 is None else ctx16[spec.feature_key], B)

        h = _e(B * H * W, boc0, dev=dev, dtype=F32)
        ops.conv_in(noisy, wts.w_conv_in, wts.b_conv_in, h.view(B, H, W, boc0))
        tape.append(("conv_in",))
        skips = [h]
        for blk in arch.down:
            for j, r in enumerate(blk.resnets):
                h = resnet(r.prefix, h)
                if blk.attns[j] is not None:
                    h = xf(blk.attns[j], h)
                skips.append(h)
                tape.append(("skip_push",))
            if blk.sampler_prefix:
                tape.append(("down", blk.sampler_prefix, hh // 2, ww // 2))
                h = self._down(blk.sampler_prefix, h, hh, ww)
                hh, ww = hh // 2, ww // 2
                skips.append(h)
                tape.append(("skip_push",))
        m0, m1 = arch.mid.resnets
        h = resnet(m0.prefix, h)
        h = xf(arch.mid.attns[0], h)
        h = resnet(m1.prefix, h)
        for blk in arch.up:
            for j, r in enumerate(blk.resnets):
                s = skips.pop()
                tape.append(("cat", h.shape[1]))
                h = torch.cat([h, s], dim=1)                                         # unet_2d_blocks.py:609,626,716
                h = resnet(r.prefix, h)
                if blk.attns[j] is not None:
                    h = xf(blk.attns[j], h)
            if blk.sampler_prefix:
                tape.append(("up", blk.sampler_prefix, hh, ww))
                h = self._up(blk.sampler_prefix, h, hh, ww)
                hh, ww = 2 * hh, 2 * ww
        x_out = h
        hw = H * W
        ws = _e(ops.groupnorm_workspace_bytes(B, self.groups), dev=dev, dtype=torch.uint8)
        gn = _e(B * hw, boc0, dev=dev)
        ops.groupnorm(x_out.view(B, hw, boc0), *self.gn_out, gn.view(B, hw, boc0), self.groups, self.eps, True, ws)
        pred = _e(B, self.cfg["out_channels"], H, W, dev=dev, dtype=F32)
        ops.conv_out(gn.view(B, H, W, boc0), self.w_conv_out, self.b_conv_out, pred)
        self._tape, self._x_out = tape, x_out
        return pred

    def backward_main(self, d_pred: torch.Tensor) -> Dict[str, torch.Tensor]:
        """d_pred fp32 NCHW [B,4,H,W] -> {attn3 parameter name: fp32 gradient} (the walk of oracle unet_backward), run on a
        power-of-two multiple of d_pred (self.grad_scale) so that fp16 gradient operands keep their precision."""
        import math
        if self.grad_scale == "auto":
            amax = float(d_pred.abs().max())
            s = 2.0 ** math.floor(math.log2(256.0 / amax)) if 0.0 < amax < float("inf") else 1.0
        else:
            s = float(self.grad_scale)
        while True:
            grads = self._backward_scaled(d_pred * s if s != 1.0 else d_pred)
            flat_ok = all(bool(torch.isfinite(g).all()) for g in grads.values()) if self.grad_scale == "auto" else True
            if flat_ok or s <= 2.0 ** -8:
                break
            s /= 16.0
        self.last_grad_scale = s
        if s != 1.0:
            torch._foreach_mul_(list(grads.values()), 1.0 / s)          # (one multi-tensor launch instead of one per gradient)
        return grads

    def _backward_scaled(self, d_pred: torch.Tensor) -> Dict[str, torch.Tensor]:
        dev, B, H, W = self.dev, self.B, self.H, self.W
        tape, x_out = self._tape, self._x_out
        boc0 = self.cfg["block_out_channels"][0]
        hw = H * W
        # ---- backward (oracle.storygen_backward.unet_backward)
        dn = _e(B * hw, boc0, dev=dev)
        ops.conv_in(d_pred, self.w_conv_out_d, self.zero_bias, dn.view(B, H, W, boc0))
        ws = _e(ops.groupnorm_bwd_workspace_bytes(B, self.groups), dev=dev, dtype=torch.uint8)
        dh = _e(B * hw, boc0, dev=dev, dtype=F32)
        ops.groupnorm_bwd(x_out.view(B, hw, boc0), dn.view(B, hw, boc0), *self.gn_out, dh.view(B, hw, boc0), self.groups, self.eps,
                          True, ws)
        grads: Dict[str, torch.Tensor] = {}
        pending: List[torch.Tensor] = []
        n_xf = sum(1 for rec in tape if rec[0] == "xf")
        for rec in reversed(tape):
            kind = rec[0]
            if kind == "up":
                dh = self._up_bwd(rec[1], dh, rec[2], rec[3])
            elif kind == "cat":
                pending.append(dh[:, rec[1]:].contiguous())
                dh = dh[:, : rec[1]].contiguous()
            elif kind == "resnet":
                dh = self.resnets[rec[1]].backward(dh)
            elif kind == "xf":
                dh, g = self.xfs[rec[1]].backward(dh)
                grads.update(g)
                if len(grads) == 5 * n_xf:
                    break                                      # nothing trainable below the first transformer block
            elif kind == "skip_push":
                dh = dh + pending.pop()
            elif kind == "down":
                dh = self._down_bwd(rec[1], dh, rec[2], rec[3])
            elif kind == "conv_in":
                break
        return grads


def allreduce_gradients(grads: Dict[str, torch.Tensor], average: bool = True) -> Dict[str, torch.Tensor]:
    """Data-parallel training (the reference wraps the UNet in DDP through accelerate, train_StorySalon_stage2.py:222): sum
    (or average) the attn3 gradients over the ranks with ONE all-reduce of one flat fp32 bucket — 80 tensors, 49.6 M
    parameters = 198 MB for SD-1.5: on xGMI's point-to-point ring a single large collective is the per-link-bandwidth
    optimum, and there is nothing to overlap it with (the gradients exist only after the backward walk has reached the
    first transformer block).  In place; identity when torch.distributed is not initialised.  With an initialised group of ONE rank
    the collective still runs (a no-op reduction over RCCL): that is how the single-GPU box exercises this exact code path
    (tests/test_optim_gpu.py::test_gradient_allreduce_over_rccl_with_one_rank)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return grads
    names = sorted(grads)
    flat = torch.cat([grads[n].reshape(-1).to(torch.float32) for n in names])
    dist.all_reduce(flat)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for n in names:
        k = grads[n].numel()
        grads[n].copy_(flat[off:off + k].view_as(grads[n]))
        off += k
    return grads


def any_rank(flag: bool, device=None) -> bool:
    """True on EVERY rank when `flag` is true on ANY rank (one 1-element MAX all-reduce; identity without a process group).  The
    GradScaler decision of a data-parallel step must be collective: under DDP a non-finite gradient on one rank reaches every rank
    through the gradient all-reduce and all of them skip the optimizer step in lockstep (train_StorySalon_stage2.py:222,328 through
    accelerate).  Here the finite check of train_step_graph is local, so the ranks agree on it explicitly — every rank calls this at
    every point where an optimizer step could happen, so the number of collectives per rank stays identical."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    dev = device if device is not None and dist.get_backend() != "gloo" else "cpu"
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0.0)


class MainPassFunction(torch.autograd.Function):
    """epsilon = UNet(sample, t, text, features) as an autograd node whose only differentiable inputs are the parameters of the
    trainer's trainable module (attn3: train_StorySalon_stage2.py:170-177; attn1 with no features: stage 1): what
    `accelerator.backward(loss)` needs from the drop-in model.

    apply(trainer, names, ctx_keys, noisy, t, text16, *features, *params): `features` are the len(ctx_keys) context
    tensors (fp16 [B*R*hw_k, C_k]), `params` the len(names) attn3 parameters under their state-dict names.  The sample,
    the text embeddings and the harvested features get no gradient (no trainable parameter sits upstream of them)."""

    @staticmethod
    def forward(ctx, trainer, names, ctx_keys, noisy, t, text16, *tensors):
        nk = len(ctx_keys)
        feats, params = tensors[:nk], tensors[nk:]
        trainer.set_trainable_parameters(dict(zip(names, params)))
        pred = trainer.forward_main(noisy, t, text16, dict(zip(ctx_keys, feats)) if nk else None)      # no features: stage 1
        ctx.trainer, ctx.names, ctx.n_feat = trainer, names, nk
        ctx.meta = [(p.dtype, p.device) for p in params]
        return pred

    @staticmethod
    def backward(ctx, d_pred):
        grads = ctx.trainer.backward_main(d_pred.to(torch.float32).contiguous())
        out = tuple(grads[n].to(dtype=dt, device=dv) for n, (dt, dv) in zip(ctx.names, ctx.meta))
        return (None,) * (6 + ctx.n_feat) + out
