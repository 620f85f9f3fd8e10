"""The denoising hot loop of StoryGen on the HIP engine.

One `step()` is exactly one iteration of /root/reference/model/pipeline.py:412-461 with classifier-free guidance:
R reference UNet passes (batch 3N: [zero, img, img] latents + [uncond, prev_text_i, prev_text_i]) that harvest the 16
diffusion features per prior frame, one main pass (batch 3N: latents x3 + [uncond, uncond, text]) whose attn3
cross-attends to them, the 3-way guidance combine (:457-458) and the DDIM update (:461).

MI355X-first structure: the whole step is ONE hipGraph (R+1 UNet passes ~ 4000 kernel nodes) replayed per step;
everything that changes between steps — the R+1 timestep vectors, the add_noise coefficients and the DDIM
coefficients — lives in a small device buffer refreshed by one async H2D copy from a pinned per-run table, so
the host does no per-kernel work at all.  Latents stay fp32 across steps (the UNet consumes them as fp16).

Data parallelism: one process per GPU, each running its own samples with no per-step communication;
`gather_latents` is the single RCCL all-gather of the final [N,4,h,w] latents.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import ops
from .arch import UNetArch
from .engine import UNetEngine
from .scheduler import DDIMSchedule

STAGES = ("multi-image-condition", "auto-regressive")


class StoryGenSampler:
    def __init__(self, arch: UNetArch, state_dict: Dict[str, torch.Tensor], device, n_samples: int = 1, height: int = 64,
                 width: int = 64, n_ref: int = 3, seq_len: int = 77, schedule: Optional[DDIMSchedule] = None,
                 use_graph: bool = True):
        if n_ref < 1:
            raise ValueError("StoryGen's loop needs at least one prior frame")
        self.arch, self.dev = arch, torch.device(device)
        self.N, self.R, self.h, self.w = n_samples, n_ref, height, width
        self.B = 3 * n_samples
        self.engine = UNetEngine(arch, state_dict, device, self.B, height, width, n_ref, seq_len)
        self.schedule = schedule or DDIMSchedule()
        self.use_graph = use_graph
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        e = self.engine
        f32 = dict(dtype=torch.float32, device=self.dev)
        lat_shape = (n_samples, arch.config["in_channels"], height, width)
        self.latents = torch.zeros(lat_shape, **f32)
        self.latents3 = torch.zeros((self.B,) + lat_shape[1:], **f32)
        self.zero = torch.zeros(lat_shape, **f32)
        self.imgs = torch.zeros((n_ref,) + lat_shape, **f32)
        self.noise = torch.zeros(lat_shape, **f32)
        self.text_main = torch.zeros(self.B, seq_len, e.cad, dtype=torch.float16, device=self.dev)
        self.text_ref = torch.zeros(n_ref, self.B, seq_len, e.cad, dtype=torch.float16, device=self.dev)
        # per-step parameters: [R+1, B] timesteps | [R, 2] add_noise coefs | [6] guidance + DDIM coefs
        self.n_par = (n_ref + 1) * self.B + 2 * n_ref + 6
        self.params = torch.zeros(self.n_par, **f32)
        self.table: Optional[torch.Tensor] = None
        self.num_steps = 0
        self.k = 0

    # ------------------------------------------------------------------------------------------------ setup
    def _par_views(self):
        R, B = self.R, self.B
        o = (R + 1) * B
        return (self.params[:o].view(R + 1, B), self.params[o:o + 2 * R].view(R, 2), self.params[o + 2 * R:])

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
        self.latents.copy_(inputs["latents"].to(dev, torch.float32) * self.schedule.init_noise_sigma)
        self.zero.copy_(inputs["zero_prompt"].to(dev, torch.float32))
        self.imgs.copy_(inputs["image_prompts"].to(dev, torch.float32))
        self.noise.copy_(inputs["noise"].to(dev, torch.float32))
        h = torch.float16
        unc, txt = inputs["uncond"].to(dev, h), inputs["text"].to(dev, h)
        self.text_main.copy_(torch.cat([unc, unc, txt]))                                  # pipeline.py:448
        for i in range(R):
            pu, pt = inputs["prev_uncond"][i].to(dev, h), inputs["prev_text"][i].to(dev, h)
            self.text_ref[i].copy_(torch.cat([pu, pt, pt]))                               # :430
        self.latents3.copy_(torch.cat([self.latents] * 3))                                # :450
        # per-step table
        ts = self.schedule.timesteps(num_inference_steps)
        rows = []
        for t in ts:
            ref_t = int(t) // 10                                                          # :414-415
            row: List[float] = []
            tis = [ref_t * (R - i) if stage == "auto-regressive" else ref_t for i in range(R)]   # :419-424
            for ti in tis:
                row += [float(ti)] * self.B
            row += [float(t)] * self.B
            for ti in tis:
                row += list(self.schedule.add_noise_coef(ti))
            row += [image_guidance_scale, guidance_scale, *self.schedule.step_coef(int(t), num_inference_steps)]
            rows.append(row)
        self.table = torch.tensor(rows, dtype=torch.float32).pin_memory()
        self.timesteps = ts
        self.num_steps = num_inference_steps
        self.k = 0
        if self.use_graph and self.graph is None:
            self._capture()

    # ------------------------------------------------------------------------------------------------ the step
    def _step_body(self):
        e, R = self.engine, self.R
        t_all, an, cd = self._par_views()
        for i in range(R):                                                                # reference passes :418-438
            ops.ref_inputs(self.zero, self.imgs[i], self.noise, an[i], e.x_in)
            e.t_in.copy_(t_all[i])
            e.text_in.copy_(self.text_ref[i])
            e.forward(harvest_slot=i)
        e.x_in.copy_(self.latents3)                                                       # main pass :448-453
        e.t_in.copy_(t_all[R])
        e.text_in.copy_(self.text_main)
        eps3 = e.forward(consume=True)
        ops.cfg_ddim_step(eps3, self.latents, self.latents3, cd)                          # :457-461

    def _capture(self):
        self.params.copy_(self.table[0], non_blocking=True)
        saved = self.latents.clone()
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            self._step_body()                                                             # warm-up (also validates args)
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_body()
        self.graph = g
        self.latents.copy_(saved)
        self.latents3.copy_(torch.cat([saved] * 3))
        torch.cuda.synchronize(self.dev)

    def step(self, k: Optional[int] = None):
        """Run denoising step k (default: the next one).  Asynchronous on the current stream."""
        k = self.k if k is None else k
        if self.table is None or k >= self.table.shape[0]:
            raise RuntimeError("prepare() first / no steps left")
        self.params.copy_(self.table[k], non_blocking=True)
        if self.graph is not None:
            self.graph.replay()
        else:
            self._step_body()
        self.k = k + 1

    def run(self, max_steps: Optional[int] = None, trace: Optional[list] = None) -> torch.Tensor:
        n = self.num_steps if max_steps is None else min(self.num_steps, max_steps)
        for k in range(self.k, n):
            self.step(k)
            if trace is not None:
                trace.append(self.latents.clone())
        return self.latents


def gather_latents(latents: torch.Tensor) -> torch.Tensor:
    """The single collective of the data-parallel loop: all-gather of the final latents (32 KiB per rank at
    512x512) over RCCL/xGMI.  Returns [world*N, 4, h, w]; identity when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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
