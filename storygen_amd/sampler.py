"""The denoising hot loop of StoryGen on the HIP engine.

One `step()` is one iteration of /root/reference/model/pipeline.py:412-461 with classifier-free guidance: the
reference passes that harvest the 16 diffusion features of every prior frame (:418-438), one main pass (batch 3N:
latents x3 with [uncond, uncond, text]) whose attn3 cross-attends to them (:440-453), the 3-way guidance combine
(:457-458) and the DDIM update (:461).

MI355X-first structure
  * The R reference passes are ONE batched UNet call.  As written, pass i runs the batch [zero_i, img_i, img_i] with
    [uncond, text_i, text_i] (:429-430): its 2nd and 3rd samples are the same tensors, and in `multi-image-condition`
    mode (same timestep for every frame, :425-427) the zero-image sample is the same in every pass.  With `dedup`
    (default) each distinct sample is computed once — N(1+R) samples per step instead of 3NR (2NR in
    `auto-regressive` mode, where each frame has its own noise level) — and its features are written to every
    context slot the loop would have put them in; the two image-conditioned CFG branches of the main pass then share
    one copy of the context and of its K/V projection.  The result is the same arithmetic on the same values
    (SURVEY F7); `dedup=False` runs all 3NR samples (still as one batched call).
  * A reference pass ends at its last harvest point (the epsilon it would go on to produce is discarded by the loop,
    :433-435), and the text K/V projections, which do not depend on the timestep, are computed once per prompt.
  * The whole step is ONE hipGraph replayed per step; everything that changes between steps — timesteps, add_noise
    and DDIM coefficients — lives in a small device buffer refreshed by one async H2D copy from a pinned table.
    Latents stay fp32 across steps.
  * The reference pass does not depend on the latents (its inputs are the prior frames noised to ref_t,
    pipeline.py:414-427), only the main pass does.  With `overlap` (default, graph mode) the captured graph of step k
    therefore has two concurrent branches on two HIP streams: the main pass of step k, reading context set k%2, and
    the reference pass of step k+1, writing context set (k+1)%2.  Most kernels of either pass leave CUs idle at batch
    1 (grids of 60-250 workgroups, latency-bound 5-20-slab pipelines), so the two branches fill each other's gaps;
    the arithmetic and its order inside each pass are unchanged (the eager path runs the same kernels back to back).

  * `ref_ahead = G > 1` goes one step further down the same road: since no reference pass depends on the latents, the
    reference samples of G consecutive steps — they differ only in their noise level — run as ONE batched UNet call of
    G x as many samples, producing the G context sets of the NEXT group of G steps while this group's G main passes
    consume theirs.  Per sample the arithmetic is unchanged; what changes is the shape of the work: G x larger M for
    every GEMM / convolution of the reference half of the step (better tile quantisation at the 64x64 level, weights
    streamed once per G steps at the weight-bound 16x16 / 8x8 levels) and G x fewer launches.  2G context sets are kept
    (set = step mod 2G).  Round 5: the whole GROUP is ONE hipGraph — the batched reference pass of group j+1 forked
    beside the G main passes of group j, each main pass reading its own row of a [G, n] parameter block uploaded once
    per group — so the two halves interleave inside the graph like the G = 1 schedule's do (separately launched graphs
    do not overlap at all on this runtime, HISTORY.md 5.2b; that older form is kept behind `split_graphs=True`).  The
    reference batch is ordered like the G context sets laid end to end in one buffer per feature key, so its features
    (and their attn3 K / V^T projections, one GEMM pair per feature for the whole group) are written in place.  One
    `step()` call per step still works: the group is enqueued by its first step, the others return at once, and
    `latents` is the state after the group's LAST step (`lat_trace[g]` = after step g of the group; `run(trace=...)`
    reads those).  The number of UNet evaluations must be a multiple of G.

Data parallelism: one process per GPU, each running its own samples with no per-step communication;
`gather_latents` is the single RCCL all-gather of the final [N,4,h,w] latents.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from .arch import UNetArch
from .engine import EngineWeights, HarvestPlan, UNetEngine
from .scheduler import DDIMSchedule

STAGES = ("multi-image-condition", "auto-regressive", "no")
# "no" (pipeline.py:425-438,444-445; what train_StorySalon_stage1.py validates with): no reference passes, the main pass runs
# without image context (attn3 is not evaluated) on [uncond, uncond, text] — a plain text-conditioned CFG loop.


class StoryGenSampler:
    def __init__(self, arch: UNetArch, state_dict: Optional[Dict[str, torch.Tensor]], device, n_samples: int = 1,
                 height: int = 64, width: int = 64, n_ref: int = 3, seq_len: int = 77,
                 schedule: Optional[DDIMSchedule] = None, use_graph: bool = True, dedup: bool = True,
                 weights: Optional[EngineWeights] = None, overlap: bool = True, ref_ahead: int = 1,
                 split_graphs: bool = False, stream_priority: bool = False, fp8_attention: bool = False,
                 side_streams: str = "auto", short_rows: bool = True, time_tables: bool = True, shared_head: bool = True,
                 ref_cus: int = 0, ref_cu_layout: str = "first", ref_eager: bool = False, ref_fp16_stream: Optional[tuple] = (True, True)):
        if n_ref < 1:
            raise ValueError("StoryGen's loop needs at least one prior frame")
        if ref_ahead < 1 or (ref_ahead > 1 and not overlap):
            raise ValueError("ref_ahead > 1 runs the batched reference pass of the NEXT group of steps beside this group's main "
                             "passes: it needs overlap=True")
        self.G = int(ref_ahead)
        # split: reference pass and main pass are separate hipGraphs overlapped at replay time by launching them on two
        # streams.  Only then can the two halves run at different queue priorities (stream_priority: main pass high — it is
        # the step's critical path — reference pass as background filler).  Measured: separately launched graphs do not
        # overlap on this runtime, so it is an A/B switch only.
        # EXPERIMENT (measured, not adopted: HISTORY.md 5.2e).  ref_cus = n > 0 (implies the split schedule): the batched reference pass
        # runs, one group ahead, on a stream whose hardware queue may only use n of the chip's CUs (hipExtStreamCreateWithCUMask), so
        # that it fills the CUs the latency-bound main passes leave idle instead of competing with them for all of them; the G main
        # passes of a group stay hipGraphs on the caller's stream.  ref_eager: the reference pass's kernels are launched one by one
        # from the host instead of as a graph replayed on that stream.  ref_cu_layout: which bits of the mask ("first" n, or "spread").
        # Round 6: (block, resnet) — which parts of the REFERENCE engine's residual stream are stored as fp16 (UNetEngine fp16_stream).
        # What a reference pass hands on are fp16 features (attention.py:263 via the context buffers), and at batch 20 its transformer GEMMs
        # and GroupNorms are HBM-bound on an fp32 stream.  Measured on all 50 steps of config 2 against the reference's own latents
        # (tools/exp_fp16_stream.py ref, profiles/r06m_*): 5.75e-4 with the fp32 stream, 5.75e-4 with both parts in fp16, -0.10 ms per
        # step.  The MAIN engine keeps the fp32 stream (round 2: fp16 there costs 1e-4 of the 1e-3 budget).  None = fp32 (A/B: bench.py
        # --ref-fp32-stream).
        self.ref_fp16_stream = ref_fp16_stream
        self.ref_cus = int(ref_cus)
        self.ref_eager = bool(ref_eager)
        if ref_cu_layout not in ("first", "spread"):
            raise ValueError("ref_cu_layout must be 'first' or 'spread'")
        self.ref_cu_layout = ref_cu_layout
        self.split = bool(split_graphs) or self.ref_cus > 0 or self.ref_eager
        if self.split and not (use_graph and overlap):
            raise ValueError("split_graphs needs use_graph=True and overlap=True")
        # group: ref_ahead = G > 1 as ONE graph per group of G steps (batched reference pass of the next group forked beside
        # this group's G main passes).  Without use_graph the same schedule runs eagerly (tests, instrumentation).
        self.group = self.G > 1 and not self.split
        if stream_priority and not self.split:
            raise ValueError("stream_priority only applies to separately launched graphs (split_graphs=True or ref_ahead > 1)")
        self.stream_priority = bool(stream_priority)
        if side_streams not in ("auto", "none", "main", "both"):
            raise ValueError("side_streams must be auto, none, main or both")
        self.side_streams = side_streams               # intra-pass side-stream forks (engine.forward(side=...)); measured neutral
        self.fp8_attention = bool(fp8_attention)       # BASELINE config 5: D = 40 image / self attention on the e4m3 MFMA path
        # shared zero-image rows keep ONE frame slot (softmax over R copies of the same keys = softmax over one copy); False = R
        # copies as written (A/B switch: bench.py --no-short-rows)
        self.short_rows = bool(short_rows)
        # time-embedding chain tabulated per distinct timestep at prepare() (UNetEngine.build_time_table); False = recomputed by
        # every UNet call as written (A/B switch: bench.py --no-time-tables)
        self.time_tables = bool(time_tables)
        # the three CFG samples of the main pass are the same latent at the same timestep (pipeline.py:448-453): everything up to the
        # first cross-attention is computed once (UNetEngine cfg_shared_head; one story frame per GPU, dedup on).  False = batch 3
        # throughout, as written (A/B switch: bench.py --no-shared-head)
        self.shared_head = bool(shared_head)
        self.arch, self.dev = arch, torch.device(device)
        self.N, self.R, self.h, self.w, self.S = n_samples, n_ref, height, width, seq_len
        self.B = 3 * n_samples
        self.weights = weights if weights is not None else EngineWeights(arch, state_dict, device)
        self.schedule = schedule or DDIMSchedule()
        self.use_graph, self.dedup = use_graph, dedup
        self.overlap = overlap and use_graph
        # ahead: the reference pass runs one step (one group) AHEAD of the main pass that consumes it — the table rows carry the
        # NEXT step's / group's reference scalars, 2G context sets exist and prepare() primes the first one
        self.ahead = self.overlap or self.group
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.graphs: List[torch.cuda.CUDAGraph] = []
        self.tails: Dict[int, torch.cuda.CUDAGraph] = {}          # parity -> graph of the loop's last group (no look-ahead pass)
        self.main: Optional[UNetEngine] = None
        self.ref: Optional[UNetEngine] = None
        self.layout = None
        self.no_ctx = False
        f32 = dict(dtype=torch.float32, device=self.dev)
        lat_shape = (n_samples, arch.config["in_channels"], height, width)
        self.latents = torch.zeros(lat_shape, **f32)
        self.latents3 = torch.zeros((self.B,) + lat_shape[1:], **f32)
        self.noise = torch.zeros(lat_shape, **f32)
        if self.schedule.kind == "plms":          # PNDM: ring of the last 4 guided epsilons + the sample kept by the first call
            self.eps_history = torch.zeros((4,) + lat_shape, **f32)
            self.kept_sample = torch.zeros(lat_shape, **f32)
        self.table: Optional[torch.Tensor] = None
        self.num_steps = 0
        self.k = 0

    @property
    def engine(self) -> UNetEngine:   # the main-pass engine
        return self.main

    # ------------------------------------------------------------------------------------------------ layout
    def _plan(self, stage: str, share_zero: bool):
        """Which reference samples are computed and where their features go.

        Returns (kind, frame, sample) per reference-pass sample u, the harvest ops (src, src_step, ctx row, first slot, count),
        the number of context rows, the main pass's attn3 groups and the number of SHORT context rows.  kind 0 = zero-image
        latent with the uncond embedding, 1 = prior frame `frame` with its prompt.  Main-pass sample order is
        [uncond/zero-image x N, uncond/frames x N, text/frames x N] (pipeline.py:448-450 with the context rows of :440-443).

        The samples are ordered so that sample u's features belong in context slot u (rows in order, frame slots inside a row): the
        reference pass then writes its fp16 feature copies straight into the context buffer (engine: HarvestPlan.direct) and no copy
        kernel runs.  With `share_zero` the zero-image rows are SHORT: as written their R frame slots hold R copies of the same
        feature map, and softmax over R copies of the same keys equals softmax over one copy, so those rows keep ONE slot and the
        image cross-attention of the zero-image CFG branch runs over HW keys instead of R * HW (sg_attn_desc.k2)."""
        N, R = self.N, self.R
        units, hops = [], []
        short = 0
        if self.dedup:
            rows = 2 * N                      # context rows: [zero-image features x N | frame features x N]
            groups = [(0, 2 * N, 0), (2 * N, N, N)]
            if share_zero and getattr(self, "short_rows", True):
                short = N
                for n in range(N):
                    units.append((0, 0, n))
                    hops.append((n, 0, n, 0, 1))                     # one sample -> the single slot of short row n
            elif share_zero:
                for n in range(N):
                    units.append((0, 0, n))
                    hops.append((n, 0, n, 0, R))                     # one sample -> all R slots of row n (copies, as written)
            else:
                for n in range(N):
                    for i in range(R):
                        units.append((0, i, n))
                    hops.append((n * R, 1, n, 0, R))                 # samples n R .. n R + R - 1 -> slots 0..R-1 of row n
            base = len(units)
            for n in range(N):
                for i in range(R):
                    units.append((1, i, n))
                hops.append((base + n * R, 1, N + n, 0, R))
        else:
            rows = 3 * N
            groups = [(0, 3 * N, 0)]
            for j in range(3):                 # all 3 N R samples of the loop as written (pass i = [zero_i | img_i | img_i]), row-major
                for n in range(N):
                    for i in range(R):
                        units.append((0 if j == 0 else 1, i, n))
                    hops.append(((j * N + n) * R, 1, j * N + n, 0, R))
        return units, hops, rows, groups, short

    def _build(self, stage: str, share_zero: bool):
        key = (stage if not share_zero else "shared-zero", self.dedup) if self.dedup else ("as-written", False)
        if stage == "no":
            key = ("no", False)
        if self.layout == key:
            return
        self.no_ctx = stage == "no"
        if self.no_ctx:
            if self.G > 1 or self.split:
                raise ValueError("stage 'no' has no reference pass to batch or split off")
            self.group = False
            self.units, self.U, self.U0 = [], 0, 0
            self.main = UNetEngine(self.arch, None, self.dev, self.B, self.h, self.w, 0, self.S, weights=self.weights,
                                   fp8_attention=self.fp8_attention)
            self.ref, self.ctx_sets, self.kv_sets, self.plans = None, [], [], []
            self.side_main = torch.cuda.Stream(device=self.dev) if self.use_graph else None
            self.side_ref = None
            self.n_par = self.B + 2 + self.schedule.row_len
            self.params = torch.zeros(1, self.n_par, dtype=torch.float32, device=self.dev)
            self.lat_trace, self.group_direct = None, False
            self.layout, self.graph, self.graphs, self.g_ref, self.g_main, self.tails = key, None, [], [], [], {}
            return
        units, hops, rows, groups, short = self._plan(stage, share_zero)
        G = self.G
        self.U0 = len(units)                  # reference samples of ONE step; the reference engine batches G steps of them
        units = units * G                     # unit g*U0 + u = sample u of the g-th step of a group
        self.units, self.U = units, len(units)
        kw = dict(weights=self.weights, fp8_attention=self.fp8_attention)
        first = self.arch.down[0]
        head = (self.shared_head and self.dedup and self.N == 1 and not self.fp8_attention
                and bool(first.attns) and first.attns[0] is not None)
        self.main = UNetEngine(self.arch, None, self.dev, self.B, self.h, self.w, self.R, self.S, ctx_rows=rows,
                               attn3_groups=groups, ctx_short=short, cfg_shared_head=head, **kw)
        self.ref = UNetEngine(self.arch, None, self.dev, self.U, self.h, self.w, 0, self.S, fp16_stream=self.ref_fp16_stream, **kw)
        # sample u of one step's reference batch IS context slot u (_plan) when the plan alone is a bijection onto the slots
        one_step_direct = HarvestPlan(self.main.ctx, hops, short=short, slots_per_row=self.R, direct=True).is_direct(self.U0, self.R)
        # context sets: the main pass of step k reads set k%2 (only one set when the reference pass is not run ahead); G > 1: set =
        # step mod 2G.  Group schedule: the G sets of a group parity are consecutive slices of ONE buffer per feature key (and of
        # one K / V^T buffer), in the order of the batched reference pass's samples — that pass then writes them in place.
        self.group_direct = self.G > 1 and self.ahead and one_step_direct
        n_sets = 2 * G if self.ahead else 1
        if self.group_direct:
            self.ctx_base, self.kv_base, self.ctx_sets, self.kv_sets = [], [], [], []
            for _ in range(2):
                cb = {k: torch.empty((G * v.numel() // v.shape[-1], v.shape[-1]), dtype=v.dtype, device=self.dev)
                      for k, v in self.main.ctx.items()}
                kb = {k: (torch.empty_like(v), torch.empty(v.shape[1], v.shape[0], dtype=v.dtype, device=self.dev)) for k, v in cb.items()}
                self.ctx_base.append(cb)
                self.kv_base.append(kb)
                for g in range(G):
                    n = {k: v.numel() // v.shape[-1] for k, v in self.main.ctx.items()}       # token rows of one set
                    self.ctx_sets.append({k: cb[k][g * n[k]:(g + 1) * n[k]].view(v.shape) for k, v in self.main.ctx.items()})
                    self.kv_sets.append({k: (kb[k][0][g * n[k]:(g + 1) * n[k]], kb[k][1][:, g * n[k]:(g + 1) * n[k]]) for k in cb})
            self.main.ctx = self.ctx_sets[0]
            # one plan for the whole batch: sample g*U0 + u -> slot u of set g = row (g*U0 + u) of the base buffer
            self.plans = [HarvestPlan(cb, [], kb, identity=True) for cb, kb in zip(self.ctx_base, self.kv_base)]
        else:
            self.ctx_sets = [self.main.ctx]
            for _ in range(n_sets - 1):
                self.ctx_sets.append({k: torch.empty_like(v) for k, v in self.main.ctx.items()})
            # attn3 K / V^T per context set: computed by the reference pass right after each feature is harvested
            self.kv_sets = [{k: (torch.empty(v.numel() // v.shape[-1], v.shape[-1], dtype=v.dtype, device=self.dev),
                                 torch.empty(v.shape[-1], v.numel() // v.shape[-1], dtype=v.dtype, device=self.dev))
                             for k, v in c.items()} for c in self.ctx_sets]
            # G = 1: the feature copies are written in place (direct); G > 1 (split graphs, or a layout that is not a bijection):
            # one plan per step's slice of the batch, strided copies
            self.plans = [HarvestPlan(c, hops, kv, src_offset=(i % G) * self.U0, short=short, slots_per_row=self.R, direct=(G == 1))
                          for i, (c, kv) in enumerate(zip(self.ctx_sets, self.kv_sets))]
        self.plan = self.plans[0]
        # side streams for the independent branches inside a pass (engine.forward(side=...)).  In overlap mode the
        # reference pass already runs on a forked stream; a second-level fork from it crashed hipGraph capture on
        # ROCm 7.2 (segfault in the runtime), so only the main pass, which runs on the capture stream, forks.
        knob = self.side_streams
        want_main = self.use_graph and knob != "none"
        want_ref = self.use_graph and (knob == "both" or (knob == "auto" and not self.overlap))
        self.side_main = torch.cuda.Stream(device=self.dev) if want_main else None
        self.side_ref = torch.cuda.Stream(device=self.dev) if want_ref else None
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.ref_src = torch.zeros((self.U,) + tuple(self.latents.shape[1:]), **f32)
        # the noise of every reference sample (pipeline.py:409: ONE draw per story frame n, shared by its prior frames and its zero-image
        # sample): expanded per reference unit, because the units are ordered like the context buffer (row-major in n), not n-minor
        self.ref_noise = torch.zeros((self.U,) + tuple(self.latents.shape[1:]), **f32)
        # per-step parameters, one row per step of a group (G rows; only the group schedule reads rows > 0):
        # [U] ref timesteps | [B] main timestep | [U,2] add_noise coefs | guidance + update-rule coefs
        self.n_par = 3 * self.U + self.B + 2 + self.schedule.row_len
        self.params = torch.zeros(G if self.group else 1, self.n_par, **f32)
        self.lat_trace = torch.zeros((G,) + tuple(self.latents.shape), **f32) if self.group else None
        self.layout, self.graph, self.graphs, self.tails = key, None, [], {}
        self.g_ref, self.g_main, self.g_prime = [], [], None       # split graphs: one graph per group parity / per context set; the primer's graph
        self.main_stream = None
        if self.split:
            self.ref_stream = (cu_masked_stream(self.dev, self.ref_cus, self.ref_cu_layout) if self.ref_cus
                               else torch.cuda.Stream(device=self.dev))
            if self.stream_priority:
                self.main_stream = torch.cuda.Stream(device=self.dev, priority=-1)
            self.ev_ref = [torch.cuda.Event(), torch.cuda.Event()]      # "reference pass of a group of this parity is done"

    def _par_views(self, row: int = 0):
        """(reference timesteps [U], main timesteps [B], add_noise coefficients [U, 2], guidance + update-rule scalars) of parameter
        row `row` (the group schedule keeps one row per step of the group; everything else uses row 0)."""
        U, B = self.U, self.B
        p = self.params[row]
        return (p[:U], p[U:U + B], p[U + B:3 * U + B].view(U, 2), p[3 * U + B:])

    # ------------------------------------------------------------------------------------------------ setup
    def prepare(self, inputs: Dict[str, torch.Tensor], num_inference_steps: int = 50,
                stage: str = "multi-image-condition", guidance_scale: float = 7.5, image_guidance_scale: float = 3.5):
        """`inputs` as produced by storygen_amd.synth.synthetic_inputs / the pipeline's CLIP+VAE plumbing
        (pipeline.py:359-409): latents, image_prompts [R,N,..], zero_prompt, noise, text, uncond, prev_text,
        prev_uncond."""
        if stage not in STAGES:
            raise ValueError(f"stage must be one of {STAGES}")
        if guidance_scale <= 1.0:
            raise ValueError("only the classifier-free-guidance path of the reference loop works (SURVEY F6g)")
        dev, N, R = self.dev, self.N, self.R
        pu = inputs["prev_uncond"]
        share_zero = (self.dedup and stage == "multi-image-condition"
                      and all(torch.equal(pu[i], pu[0]) for i in range(1, R)))
        self._build(stage, share_zero)
        # ---- the per-step table: host arithmetic only
        ts = self.schedule.timesteps(num_inference_steps)
        rows, row0 = step_table(self.schedule, ts, num_inference_steps, self.units[:self.U0], R, stage, self.B, self.G,
                                self.ahead and not self.no_ctx, image_guidance_scale, guidance_scale)
        if self.group and len(rows) % self.G:
            raise ValueError(f"ref_ahead = {self.G} runs the loop in groups of {self.G} UNet evaluations: {len(rows)} evaluations "
                             f"({num_inference_steps} inference steps) is not a multiple")
        self.row0_ref = self._pinned("row0", torch.tensor(row0, dtype=torch.float32))      # pinned: the per-step upload is an async H2D copy
        self.table = self._pinned("table", torch.tensor(rows, dtype=torch.float32))
        self.timesteps = ts
        self.num_steps = len(ts)                  # PNDM: n + 1 UNet evaluations for n inference steps
        self.k = 0
        # ---- uploads.  The inputs are pageable host tensors, i.e. every .to(device) blocks the host until the stream reaches it: all of
        # them go first (few, batched — one stacked copy per kind instead of one per reference sample), nothing long is queued yet
        self.latents.copy_(inputs["latents"].to(dev, torch.float32) * self.schedule.init_noise_sigma)
        self.noise.copy_(inputs["noise"].to(dev, torch.float32))
        h = torch.float16
        unc, txt = inputs["uncond"].to(dev, h), inputs["text"].to(dev, h)
        self.main.text_in.copy_(torch.cat([unc, unc, txt]))                               # pipeline.py:448
        if self.units:                                                                    # :420-430 (none for stage "no")
            zero, imgs = inputs["zero_prompt"].to(dev, torch.float32), inputs["image_prompts"].to(dev, torch.float32)
            src = torch.cat([zero, imgs.flatten(0, 1)])                                   # [N + R N, ...]: zero-image latents, then frame i of sample n at N + i N + n
            pick = torch.tensor([n if kind == 0 else N + i * N + n for kind, i, n in self.units], device=dev)
            self.ref_src.copy_(src[pick])
            self.ref_noise.copy_(self.noise[torch.tensor([n for _, _, n in self.units], device=dev)])
            self.ref.text_in.copy_(torch.stack([(pu[i][n] if kind == 0 else inputs["prev_text"][i][n]) for kind, i, n in self.units]).to(dev, h))
        self.latents3.copy_(torch.cat([self.latents] * 3))                                # :450
        # ---- the reference engine's side first: with the graphs of an earlier prepare() still valid, the primer pass (nothing overlaps
        # with it) is enqueued BEFORE the main engine's tables and text projections are made, so that their host time runs under it
        U, B = self.U, self.B
        stale = False
        primed = False
        if self.ref is not None:
            if self.time_tables:                  # every timestep of the loop is known now: tabulate the time-embedding chain once
                stale |= self.ref.build_time_table([t for r in rows + [row0] for t in r[:U]])
            self.ref.cache_text_kv()
            if self.ahead and not self.no_ctx and not stale and (self.graphs or self.g_main):
                self._prime()
                primed = True
        if self.time_tables:
            stale |= self.main.build_time_table([r[U] for r in rows] + [row0[U]])
        self.main.cache_text_kv()
        if stale:                                 # first table (or one that outgrew its buffers): graphs captured earlier read the old ones
            self.graph, self.graphs, self.g_ref, self.g_main, self.tails = None, [], [], [], {}
        if self.use_graph and self.graph is None and not self.graphs and not self.g_main:
            self._capture()                       # (its warm-up step overwrites context set 0: the primer follows)
            primed = False
        last = self._last_lookahead_at()
        if last is not None and self.graphs:      # the last group's graph has no look-ahead pass: captured here (a capture synchronises once, the first time)
            self._tail_graph((last // self.G) % 2)
        if self.ahead and not self.no_ctx and not primed:
            self._prime()

    def _pinned(self, name: str, t: torch.Tensor) -> torch.Tensor:
        """t in a pinned host buffer kept across prepare() calls (a fresh pin_memory() allocation costs milliseconds per call)."""
        if self.dev.type != "cuda":
            return t
        bufs = self.__dict__.setdefault("_pin_bufs", {})
        b = bufs.get(name)
        if b is None or b.shape != t.shape:
            b = bufs[name] = torch.empty_like(t).pin_memory()
        b.copy_(t)
        return b

    # ------------------------------------------------------------------------------------------------ the step
    def _step_body(self):
        """One whole step, sequentially (eager mode, graph warm-up, bench instrumentation): reference passes :418-438,
        then the main pass.  NB when the reference pass runs ahead `self.params` holds the reference-pass scalars of the NEXT step.
        With ref_ahead = G > 1 this is one whole GROUP: the batched reference pass of G steps and G main passes (as a warm-up all
        with the scalars of one table row: good for kernel timing, not a valid piece of a trajectory)."""
        if not self.no_ctx:
            self._ref_pass(0)
        for g in range(self.G):
            self._main_pass(g, row=g if self.group else 0)

    def _ref_pass(self, ctx_set: int):
        """The reference samples of step(s) -> context set `ctx_set` (ref_ahead = G > 1: of G steps -> sets ctx_set ..
        ctx_set + G - 1: one harvest plan per step's slice of the batch, or ONE plan over the group's contiguous sets)."""
        t_ref, _, an, _ = self._par_views()
        ops.add_noise(self.ref_src, self.ref_noise, an, self.ref.x_in)                    # :419-429 (noise of unit u = noise[n(u)])
        self.ref.t_in.copy_(t_ref)
        if self.group_direct:
            plan = self.plans[ctx_set // self.G]
        else:
            plan = self.plans[ctx_set] if self.G == 1 else self.plans[ctx_set:ctx_set + self.G]
        self.ref.forward(harvest=plan, harvest_only=True, text_cache=True, side=self.side_ref)

    def _main_pass(self, ctx_set: int, row: int = 0):
        _, t_main, _, cd = self._par_views(row)
        main = self.main
        if not self.no_ctx:
            main.ctx, main.kv_ext = self.ctx_sets[ctx_set], self.kv_sets[ctx_set]
        main.x_in.copy_(self.latents3)                                                    # :448-453
        main.t_in.copy_(t_main)
        eps3 = main.forward(consume=not self.no_ctx, text_cache=True, side=self.side_main)
        if self.schedule.kind == "plms":                                                  # :457-461
            ops.cfg_plms_step(eps3, self.latents, self.latents3, self.eps_history, self.kept_sample, cd)
        else:
            ops.cfg_ddim_step(eps3, self.latents, self.latents3, cd)

    def _group_body(self, parity: int, side: Optional["torch.cuda.Stream"], with_ref: bool = True):
        """Group schedule: the G main passes of a group of this parity (context sets parity*G ..) and, beside them, the batched
        reference pass of the NEXT group (into the other parity's sets).  side = the stream of the forked branch (None: in order,
        reference pass first — the eager form).  with_ref=False: the LAST group of a loop — there is no next group, so no look-ahead
        pass (the reference's loop runs no reference pass after its last step either, pipeline.py:412-469)."""
        G, dev = self.G, self.dev
        if not with_ref:
            side = None
        elif side is not None:
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)                                                         # fork
            with torch.cuda.stream(side):
                self._ref_pass((1 - parity) * G)
        else:
            self._ref_pass((1 - parity) * G)
        for g in range(G):
            self._main_pass(parity * G + g, row=g)
            self.lat_trace[g].copy_(self.latents)
        if side is not None:
            cur.wait_stream(side)                                                         # join

    def _prime(self):
        """The reference pass of step 0 (ref_ahead > 1: of the first group) has no main pass to hide behind.  Replayed from its own
        graph when the schedule is captured (round 6: launched kernel by kernel it cost the host ~10 ms of every prepare())."""
        self.params.copy_(self.row0_ref, non_blocking=True)
        if getattr(self, "g_prime", None) is not None:
            self.g_prime.replay()
        else:
            self._ref_pass(0)
        if self.split:
            self.ev_ref[0].record(torch.cuda.current_stream(self.dev))

    def _last_lookahead_at(self) -> Optional[int]:
        """The step whose graph would run a look-ahead reference pass that nothing consumes (the first step of the last group; G = 1:
        the last step), or None when the schedule has no look-ahead to skip (no overlap, stage "no", separately launched graphs)."""
        if not (self.ahead and not self.no_ctx and not self.split and self.num_steps):
            return None
        return (self.num_steps - 1) // self.G * self.G

    def _tail_graph(self, parity: int):
        """The graph of the loop's LAST group (G = 1: last step): its main pass(es) WITHOUT the forked look-ahead reference pass (round 6:
        the one-graph form replayed a whole batched reference pass after the last step — ~ 16 ms of a 50-step image at G = 5 whose
        features nobody reads).  Captured by prepare() for the parity this loop's last group has; replays of it leave every buffer
        the other graphs use untouched."""
        g = self.tails.get(parity)
        if g is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if self.group:
                    self._group_body(parity, None, with_ref=False)
                else:
                    self._main_pass(parity)
            self.tails[parity] = g
            self.main.ctx, self.main.kv_ext = self.ctx_sets[0], self.kv_sets[0]
        return g

    def _capture(self):
        dev = self.dev
        self.tails = {}
        self.g_prime = None
        self.params.copy_(self.table[0], non_blocking=True)
        saved = self.latents.clone()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            self._step_body()                                                             # warm-up (also validates args)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        if not self.overlap or self.no_ctx:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_body()
            self.graph = g
        elif self.split:
            # separate graphs, overlapped at replay time by launching them on two streams (step()): the batched reference
            # pass of a group into context sets [p*G, (p+1)*G), and the main pass reading context set s
            self.g_ref, self.g_main = [], []
            for p in ((0, 1) if not self.ref_eager else ()):        # (ref_eager: the reference pass is launched kernel by kernel, never captured)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._ref_pass(first_ctx_set_of_group(p, self.G))
                self.g_ref.append(g)
            for s_ in range(2 * self.G):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._main_pass(s_)
                self.g_main.append(g)
        else:
            side = torch.cuda.Stream(device=dev)
            self.graphs = []
            for parity in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if self.group:                                                        # G main passes of a group || reference pass of the next
                        self._group_body(parity, side)
                    else:
                        cur = torch.cuda.current_stream(dev)
                        side.wait_stream(cur)                                             # fork
                        with torch.cuda.stream(side):
                            self._ref_pass(1 - parity)                                    # reference pass of step k+1
                        self._main_pass(parity)                                           # main pass of step k
                        cur.wait_stream(side)                                             # join
                self.graphs.append(g)
            self.g_prime = torch.cuda.CUDAGraph()          # the primer: the reference pass into context set(s) 0 .., alone
            with torch.cuda.graph(self.g_prime):
                self._ref_pass(0)
        if not self.no_ctx:
            self.main.ctx, self.main.kv_ext = self.ctx_sets[0], self.kv_sets[0]
        self.latents.copy_(saved)
        self.latents3.copy_(torch.cat([saved] * 3))
        # the warm-up ran main passes on table rows that are not a piece of any trajectory (G > 1: G passes on ONE row): a LayerNorm-fold
        # guard bit set there says nothing about the run that follows — only flags of the real trajectory may raise (check_guards)
        for eng in (self.main, self.ref):
            if eng is not None and getattr(eng, "ln_guard", None) is not None:
                eng.ln_guard.zero_()
        torch.cuda.synchronize(dev)

    def step(self, k: Optional[int] = None):
        """Run denoising step k (default: the next one).  Asynchronous on the current stream.  Group schedule (ref_ahead > 1): the
        first step of a group enqueues the whole group, the others return at once; `latents` is then the state after the group's
        last step and `lat_trace[g]` the state after its g-th."""
        k = self.k if k is None else k
        if self.table is None or k >= self.table.shape[0]:
            raise RuntimeError("prepare() first / no steps left")
        if self.ahead and not self.no_ctx and k != self.k:
            raise RuntimeError("look-ahead sampling runs the steps in order (the graph of step k also runs the reference "
                               "pass of step k+1)")
        if self.g_main:
            self._step_ahead(k)
            self.k = k + 1
            return
        last = k == self._last_lookahead_at()       # the loop ends with this group / step: no look-ahead reference pass
        if self.group:
            G = self.G
            if k % G == 0:
                self.params.copy_(self.table[k:k + G], non_blocking=True)
                if self.graphs:
                    (self._tail_graph((k // G) % 2) if last else self.graphs[(k // G) % 2]).replay()
                else:
                    self._group_body((k // G) % 2, None, with_ref=not last)
            self.k = k + 1
            return
        self.params.copy_(self.table[k], non_blocking=True)
        if self.graphs:
            (self._tail_graph(k % 2) if last else self.graphs[k % 2]).replay()
        elif self.graph is not None:
            self.graph.replay()
        else:
            self._step_body()
        self.k = k + 1

    def _step_ahead(self, k: int):
        """Split-graph schedule (split_graphs=True, any G).  At the first step of group j: the batched reference pass of group j+1 goes to the second
        stream (it overwrites the context sets group j-1 read, whose main passes are already enqueued on this stream:
        the fork orders it behind them), and this stream waits for group j's own reference pass, launched one group ago
        (or by _prime).  Then the main pass of step k, on its context set k mod 2G.
        The parameter buffer is shared: its reference-pass part is only read by the reference graph (uploaded here, on
        the second stream, behind the previous reference graph), its main part only by the main graphs."""
        G, dev = self.G, self.dev
        j, g = divmod(k, G)
        caller = torch.cuda.current_stream(dev)
        cur = self.main_stream or caller            # stream_priority: the main passes run on their own high-priority stream
        U, B = self.U, self.B
        row = self.table[k]
        par = self.params[0]
        if cur is not caller:
            cur.wait_stream(caller)
        if g == 0:
            self.ref_stream.wait_stream(cur)
            with torch.cuda.stream(self.ref_stream):
                par[:U].copy_(row[:U], non_blocking=True)                                 # reference timesteps
                par[U + B:3 * U + B].copy_(row[U + B:3 * U + B], non_blocking=True)       # add_noise coefficients
                if self.ref_eager:
                    self._ref_pass(first_ctx_set_of_group(j + 1, G))                      # kernel by kernel, on the (CU-masked) stream
                else:
                    self.g_ref[first_ctx_set_of_group(j + 1, G) // G].replay()
                self.ev_ref[(j + 1) % 2].record(self.ref_stream)
            cur.wait_event(self.ev_ref[j % 2])
        with torch.cuda.stream(cur):
            par[U:U + B].copy_(row[U:U + B], non_blocking=True)                           # main timestep
            par[3 * U + B:].copy_(row[3 * U + B:], non_blocking=True)                     # guidance + DDIM coefficients
            self.g_main[ctx_set_of_step(k, G)].replay()
        if cur is not caller:
            caller.wait_stream(cur)

    def run(self, max_steps: Optional[int] = None, trace: Optional[list] = None) -> torch.Tensor:
        n = self.num_steps if max_steps is None else min(self.num_steps, max_steps)
        if self.group and n % self.G:
            raise ValueError(f"ref_ahead = {self.G}: run() works in whole groups, max_steps = {n} is not a multiple")
        for k in range(self.k, n):
            self.step(k)
            if trace is not None and not self.group:
                trace.append(self.latents.clone())
            elif trace is not None and k % self.G == self.G - 1:
                trace.extend(self.lat_trace[g].clone() for g in range(self.G))
        return self.latents

    def check_guards(self):
        """Raise if a folded LayerNorm of either engine left its range since the last check (UNetEngine.check_ln_guard; synchronises).
        The pipeline calls it once after the loop, bench.py after the timed region."""
        for eng in (self.main, self.ref):
            if eng is not None:
                eng.check_ln_guard()

    def executed_sample_forwards(self):
        """(reference-pass samples, main-pass samples) actually computed per step — for FLOP accounting (with ref_ahead
        = G the reference engine runs G steps' samples once per G steps)."""
        return self.U0, self.B


def cu_masked_stream(device, n_cus: int, layout: str = "first"):
    """A HIP stream whose hardware queue may only use n_cus compute units (hipExtStreamCreateWithCUMask: bit i of the mask = CU i),
    wrapped as a torch stream.  layout "first": CUs 0 .. n-1; "spread": n CUs spaced evenly over the chip's CU numbering."""
    import ctypes
    dev = torch.device(device)
    total = torch.cuda.get_device_properties(dev).multi_processor_count
    n = max(1, min(int(n_cus), total))
    bits = range(n) if layout == "first" else sorted({(i * total) // n for i in range(n)})
    words = [0] * ((total + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    hip = ctypes.CDLL("libamdhip64.so")
    handle = ctypes.c_void_p()
    with torch.cuda.device(dev):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(len(words)), (ctypes.c_uint32 * len(words))(*words))
    if rc != 0 or not handle.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(handle.value, device=dev)


def ctx_set_of_step(k: int, G: int) -> int:
    """Context set the main pass of step k reads (overlap mode: 2G sets; G = 1 is the classic k % 2)."""
    return k % (2 * G)


def first_ctx_set_of_group(j: int, G: int) -> int:
    """First of the G consecutive context sets the batched reference pass of group j (steps jG .. jG+G-1) writes."""
    return (j % 2) * G


def step_table(schedule: DDIMSchedule, ts, num_inference_steps: int, units0, R: int, stage: str, B: int, G: int, overlap: bool,
               image_guidance_scale: float, guidance_scale: float):
    """The scalars every denoising step needs, as rows of the pinned table the sampler uploads from (pure host logic).

    Row k = [U reference timesteps | B main timesteps | U x 2 add_noise coefficients | 2 guidance scales + the update
    rule's scalars (schedule.step_row: 4 for DDIM, 13 for PNDM/PLMS)], U = G * len(units0).  units0 = the (kind, frame, sample) reference samples of ONE step.
    Which reference scalars row k carries depends on the schedule of the passes:
      no overlap          : those of step k itself (reference pass, then main pass);
      overlap, G = 1      : those of step k+1 (graph k runs main pass k beside reference pass k+1);
      overlap, G > 1      : those of the G steps of group k // G + 1, step-major (uploaded only when k % G == 0: the
                            batched reference pass of the next group starts with this group's first main pass).
    Steps past the end repeat the last timestep (their features are never consumed).  Also returns row0: the reference
    scalars of the very first pass / group (the one nothing overlaps with), main part zero."""
    T = len(ts)

    def ref_part(t):
        ref_t = int(t) // 10                                                              # pipeline.py:414-415
        tis = [ref_t * (R - i) if stage == "auto-regressive" else ref_t for i in range(R)]   # :419-427
        tt = [float(tis[i]) for _, i, _ in units0]
        cc: List[float] = []
        for _, i, _ in units0:
            cc += list(schedule.add_noise_coef(tis[i]))
        return tt, cc

    def group_ref(j):
        tt, cc = [], []
        for g in range(G):
            a, b = ref_part(ts[min(j * G + g, T - 1)])
            tt, cc = tt + a, cc + b
        return tt, cc

    rows = []
    for k, t in enumerate(ts):
        if G > 1:
            tt, cc = group_ref(k // G + 1)
        else:
            tt, cc = ref_part(ts[min(k + 1, T - 1)] if overlap else t)
        row = tt + [float(t)] * B + cc
        row += [image_guidance_scale, guidance_scale, *schedule.step_row(k, ts, num_inference_steps)]
        rows.append(row)
    tt, cc = group_ref(0) if G > 1 else ref_part(ts[0])
    return rows, tt + [float(ts[0])] * B + cc + [0.0] * (2 + schedule.row_len)


def gather_latents(latents: torch.Tensor) -> torch.Tensor:
    """The single collective of the data-parallel loop: all-gather of the final latents (32 KiB per rank at
    512x512) over RCCL/xGMI.  Returns [world*N, 4, h, w]; identity when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return latents
    out = [torch.empty_like(latents) for _ in range(dist.get_world_size())]
    dist.all_gather(out, latents.contiguous())
    return torch.cat(out, dim=0)


def shard_for_rank(n_total: int, rank: int, world: int) -> range:
    """Sample indices owned by `rank` when n_total independent story-frame samples are spread over `world` GPUs
    (contiguous blocks, remainder to the low ranks)."""
    q, r = divmod(n_total, world)
    start = rank * q + min(rank, r)
    return range(start, start + q + (1 if rank < r else 0))
