"""Host-side logic that needs no GPU: the DDIM tables the sampler uploads, the static architecture description
(state-dict contract of SURVEY §8b), the weight repacks the kernels consume, the per-step parameter table, and the
data-parallel sharding + the single all-gather (gloo, world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ scheduler
def test_ddim_schedule_matches_oracle_and_config():
    from oracle import storygen_oracle as O
    from storygen_amd.scheduler import DDIMSchedule
    s, o = DDIMSchedule(), O.DDIM()
    assert torch.equal(s.alphas_cumprod, o.alphas_cumprod)
    for n in (1, 4, 40, 50):
        assert s.timesteps(n) == o.timesteps(n)
    x, eps = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    for t in (981, 501, 21, 1):
        ca, cb = s.add_noise_coef(t // 10)
        assert torch.allclose(ca * x + cb * eps, o.add_noise(x, eps, t // 10), atol=1e-6)
        sa, sb, pa, pb = s.step_coef(t, 50)
        mine = pa * ((x - sb * eps) / sa) + pb * eps
        assert torch.allclose(mine, o.step(eps, t, x, 50), atol=1e-5, rtol=1e-5)
    # last step uses final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one = false in scheduler_config.json)
    assert s.step_coef(1, 50)[2] == pytest.approx(float(s.alphas_cumprod[0] ** 0.5))


def test_ddim_schedule_from_shipped_json(tmp_path):
    import json
    from storygen_amd.scheduler import DDIMSchedule
    cfg = {"_class_name": "PNDMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
           "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True, "steps_offset": 1,
           "trained_betas": None, "clip_sample": False}        # ckpt/stable-diffusion-v1-5/scheduler/scheduler_config.json
    os.makedirs(tmp_path / "scheduler")
    (tmp_path / "scheduler" / "scheduler_config.json").write_text(json.dumps(cfg))
    s = DDIMSchedule.from_pretrained(str(tmp_path))
    assert s.timesteps(50)[0] == 981 and torch.equal(s.alphas_cumprod, DDIMSchedule().alphas_cumprod)


def test_saved_ddim_config_says_clip_sample_false(tmp_path):
    """diffusers' DDIMScheduler defaults clip_sample to True, so the config this framework writes (pipeline / trainer
    checkpoints) must carry the key explicitly — else the reference's inference.py:48 would clip x0 on a checkpoint saved here.
    PNDMScheduler has no such key."""
    import json
    from storygen_amd.scheduler import DDIMSchedule, PNDMSchedule
    DDIMSchedule().save_pretrained(str(tmp_path / "d"))
    cfg = json.loads((tmp_path / "d" / "scheduler_config.json").read_text())
    assert cfg["_class_name"] == "DDIMScheduler" and cfg["clip_sample"] is False
    assert DDIMSchedule.from_pretrained(str(tmp_path / "d"), subfolder=None).key() == DDIMSchedule().key()
    PNDMSchedule(skip_prk_steps=True).save_pretrained(str(tmp_path / "p"))
    assert "clip_sample" not in json.loads((tmp_path / "p" / "scheduler_config.json").read_text())


def test_scheduler_configs_are_honoured_or_rejected():
    """trained_betas must be used, non-epsilon prediction / unknown keys / unknown classes must raise, and the
    config may be a dict or an attribute object (the shim's DDIMScheduler.config)."""
    from types import SimpleNamespace
    from storygen_amd.scheduler import DDIMSchedule, PNDMSchedule, schedule_from_config
    betas = torch.linspace(1e-4, 2e-2, 1000).tolist()
    s = DDIMSchedule(trained_betas=betas)
    assert torch.allclose(s.alphas_cumprod, torch.cumprod(1 - torch.tensor(betas), 0))
    assert not torch.equal(s.alphas_cumprod, DDIMSchedule().alphas_cumprod)
    with pytest.raises(NotImplementedError, match="prediction_type"):
        DDIMSchedule(prediction_type="v_prediction")
    with pytest.raises(NotImplementedError, match="unsupported scheduler config keys"):
        DDIMSchedule(thresholding=True)
    cfg = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
               set_alpha_to_one=False, skip_prk_steps=True, trained_betas=None)
    assert isinstance(schedule_from_config(dict(cfg, _class_name="PNDMScheduler")), PNDMSchedule)
    assert type(schedule_from_config(SimpleNamespace(**cfg), "DDIMScheduler")) is DDIMSchedule      # attribute-style config
    with pytest.raises(NotImplementedError, match="EulerDiscreteScheduler"):
        schedule_from_config(cfg, "EulerDiscreteScheduler")
    with pytest.raises(NotImplementedError, match="skip_prk_steps"):
        PNDMSchedule(skip_prk_steps=False)
    assert DDIMSchedule().key() == DDIMSchedule().key() != PNDMSchedule(skip_prk_steps=True).key()


def test_plms_table_reproduces_the_stateful_pndm_rule():
    """The device applies PNDM/PLMS from a per-call table (PNDMSchedule.step_row: weights over a 4-slot ring of past
    epsilons); here that table, executed in plain torch exactly as sg_cfg_plms_step_f32 does, is compared with the stateful
    restatement of diffusers' PNDMScheduler.step_plms in the oracle."""
    from oracle import storygen_oracle as O
    from storygen_amd.scheduler import PNDMSchedule
    g = torch.Generator().manual_seed(0)
    for n in (1, 2, 3, 5, 12, 50):
        ps, po = PNDMSchedule(skip_prk_steps=True), O.PNDM()
        ts = ps.timesteps(n)
        assert ts == po.timesteps(n) and len(ts) == (n + 1 if n > 1 else 1)
        x = torch.randn(4, 8, generator=g, dtype=torch.float64)
        xo, hist, kept = x.clone(), torch.zeros(4, 4, 8, dtype=torch.float64), torch.zeros(4, 8, dtype=torch.float64)
        for k, t in enumerate(ts):
            e = torch.randn(4, 8, generator=g, dtype=torch.float64)
            A, Bc, w0, w1, w2, w3, cur, s1, s2, s3, push, use_kept, keep = ps.step_row(k, ts, n)
            ep = w0 * e + w1 * hist[int(s1)] + w2 * hist[int(s2)] + w3 * hist[int(s3)]
            if push:
                hist[int(cur)] = e
            xs = kept.clone() if use_kept else x
            if keep:
                kept = x.clone()
            x = A * xs - Bc * ep
            xo = po.step(e, t, xo, n)
            assert torch.allclose(x, xo.double(), rtol=1e-5, atol=1e-6), (n, k)


# ------------------------------------------------------------------------------------------------ architecture
def test_arch_sd15_parameter_census():
    from storygen_amd.arch import SD15_CONFIG, build_arch, feature_shapes, param_shapes
    arch = build_arch(SD15_CONFIG)
    shapes = param_shapes(arch)
    n = sum(int(torch.Size(s).numel()) for s in shapes.values())
    assert 905e6 < n < 912e6                                   # SD-1.5's 859.5 M + 49.6 M of attn3/norm4
    attn3 = sum(int(torch.Size(s).numel()) for k, s in shapes.items() if ".attn3." in k or ".norm4." in k)
    assert 49e6 < attn3 < 50.5e6
    assert len(arch.resnets) == 22 and len(arch.feature_keys) == 16
    assert arch.feature_keys == ["down_1_1", "down_1_2", "down_2_1", "down_2_2", "down_3_1", "down_3_2", "mid", "up_1_1",
                                 "up_1_2", "up_1_3", "up_2_1", "up_2_2", "up_2_3", "up_3_1", "up_3_2", "up_3_3"]
    fs = feature_shapes(arch, 64, 64)
    assert fs["down_1_1"] == (4096, 320) and fs["mid"] == (64, 1280) and fs["up_1_1"] == (256, 1280) and fs["up_3_3"] == (4096, 320)
    # resnet input widths of the up path (concat with the popped skips), SURVEY §2b
    ups = [r.cin for b in arch.up for r in b.resnets]
    assert ups == [2560, 2560, 2560, 2560, 2560, 1920, 1920, 1280, 960, 960, 640, 640]
    assert shapes["up_blocks.1.resnets.2.conv_shortcut.weight"] == (1280, 1920, 1, 1)
    assert shapes["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    assert shapes["down_blocks.0.attentions.0.transformer_blocks.0.attn3.to_k.weight"] == (320, 320)
    assert "down_blocks.3.downsamplers.0.conv.weight" not in shapes and "up_blocks.3.upsamplers.0.conv.weight" not in shapes


def test_arch_rejects_configs_off_the_storygen_path():
    from storygen_amd.arch import SD15_CONFIG, build_arch
    for bad in (dict(use_linear_projection=True), dict(class_embed_type="timestep"), dict(time_embedding_type="fourier"),
                dict(down_block_types=("AttnDownBlock2D",) * 4), dict(mid_block_type="UNetMidBlock2DSimpleCrossAttn")):
        with pytest.raises(ValueError):
            build_arch(dict(SD15_CONFIG, **bad))


def test_synthetic_tensors_are_order_and_device_independent():
    from storygen_amd.arch import build_arch
    from storygen_amd.synth import synthetic_inputs, synthetic_state_dict
    cfg = dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
               up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48)
    a = synthetic_state_dict(build_arch(cfg), 3)
    b = synthetic_state_dict(build_arch(cfg), 3)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert all(torch.equal(v, v.half().float()) for v in a.values())          # fp16-exact by construction
    i1, i2 = synthetic_inputs(1, 2, 8, 8, 5, 48), synthetic_inputs(1, 2, 8, 8, 5, 48)
    assert all(torch.equal(i1[k], i2[k]) for k in i1)
    assert not torch.equal(synthetic_inputs(1, 2, 8, 8, 6, 48)["latents"], i1["latents"])


# ------------------------------------------------------------------------------------------------ repacks
def test_conv_repacks_are_the_same_linear_map():
    from storygen_amd.repack import conv1x1_nk, conv3x3_krsc, conv_in_kn
    w = torch.randn(16, 8, 3, 3)
    x = torch.randn(2, 8, 6, 6)
    want = F.conv2d(x, w, padding=1)
    cols = F.unfold(x, 3, padding=1).view(2, 8, 9, 36).permute(0, 3, 2, 1).reshape(2, 36, 72)   # k = tap*Cin + ci
    got = (cols @ conv3x3_krsc(w).reshape(16, 72).t()).permute(0, 2, 1).reshape(2, 16, 6, 6)
    assert torch.allclose(got, want, atol=1e-4)
    got_in = (cols @ conv_in_kn(w)).permute(0, 2, 1).reshape(2, 16, 6, 6)
    assert torch.allclose(got_in, want, atol=1e-4)
    w1 = torch.randn(16, 8, 1, 1)
    assert torch.allclose(F.conv2d(x, w1), torch.einsum("bchw,oc->bohw", x, conv1x1_nk(w1)), atol=1e-5)


def test_geglu_interleave_roundtrip():
    from storygen_amd.repack import interleave_geglu
    inner, k = 128, 16
    w, b = torch.randn(2 * inner, k), torch.randn(2 * inner)
    wi, bi = interleave_geglu(w, b)
    x = torch.randn(5, k)
    val, gate = (x @ w.t() + b).chunk(2, dim=-1)                 # attention.py:391-392
    want = val * F.gelu(gate)
    y = x @ wi.t() + bi
    y = y.view(5, inner // 32, 2, 32)                            # [64u, 64u+32) value | [64u+32, 64u+64) gate
    got = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(5, inner)
    assert torch.allclose(got, want, atol=1e-5)


# ------------------------------------------------------------------------------------------------ data parallel
def test_shard_for_rank_partitions_the_samples():
    from storygen_amd.sampler import shard_for_rank
    for n, world in ((8, 8), (8, 4), (10, 4), (3, 8), (1, 1)):
        parts = [list(shard_for_rank(n, r, world)) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from storygen_amd.sampler import gather_latents, shard_for_rank
    mine = shard_for_rank(4, rank, world)
    # each rank "denoises" its own samples: the payload encodes the global sample index
    lat = torch.stack([torch.full((4, 8, 8), float(i)) for i in mine])
    allv = gather_latents(lat)
    torch.save(allv, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_final_allgather_world2_gloo(tmp_path):
    """The one collective of the DP path: every rank ends with all latents in global sample order."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a, b) and a.shape == (4, 4, 8, 8)
    assert [float(a[i, 0, 0, 0]) for i in range(4)] == [0.0, 1.0, 2.0, 3.0]


def _ddp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from storygen_amd.train import allreduce_gradients
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    grads = {"b.attn3.to_q.weight": torch.full((4, 4), float(rank + 1)), "a.attn3.to_out.0.bias": torch.arange(4.0) * (rank + 1)}
    allreduce_gradients(grads)
    torch.save(grads, os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo(tmp_path):
    """Data-parallel training: one bucketed all-reduce averages the attn3 gradients over the ranks."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    for k in a:
        assert torch.equal(a[k], b[k])
    assert torch.equal(a["b.attn3.to_q.weight"], torch.full((4, 4), 1.5))
    assert torch.equal(a["a.attn3.to_out.0.bias"], torch.arange(4.0) * 1.5)


def test_traffic_on_file_was_measured_on_this_build():
    """roofline.traffic comes from profiles/traffic.json (rocprofv3 PMC passes of bench.py): it must carry the hash of the
    kernel sources it was measured on, and that hash must be the current one — a kernel edit without a re-measurement fails here
    (the number on file was several kernel changes old)."""
    import json
    from storygen_amd.build import source_hash
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    with open(path) as f:
        t = json.load(f)
    assert t.get("kernel_source_hash") == source_hash(), "re-run the two PMC passes on the GPU box (tools/calls/r4_call21.sh) and copy traffic.json to profiles/"
    assert t["kernels"]["mma_pipe_body (gemm + conv3x3: mma_pipe_kernel / mma_lat_kernel)"]["hbm_bytes_per_launch"] > 0


def test_producers_emit_groupnorm_statistics_for_the_wider_concats_of_the_up_path():
    """UNetEngine._stats_wanted: a producer writes GroupNorm partials when ANY GroupNorm that can read its output takes them — its own
    width or a channel concat [h | skip] of the up blocks (unet_2d_blocks.py:600-601).  SD-1.5: at 16x16 a 1280-channel GroupNorm is
    the one-launch kernel (no partials), but the 2560- / 1920-channel concats are the wide pair; at 8x8 nothing takes them."""
    from storygen_amd.arch import SD15_CONFIG, build_arch
    from storygen_amd.engine import UNetEngine
    eng = object.__new__(UNetEngine)
    eng.arch, eng.groups, eng._stats_want = build_arch(SD15_CONFIG), 32, {}
    eng.hw = [64 * 64, 32 * 32, 16 * 16, 8 * 8]
    assert eng._stats_wanted(0, 320) and eng._stats_wanted(1, 640) and eng._stats_wanted(1, 320)
    assert eng._stats_wanted(2, 1280) and eng._stats_wanted(2, 640)        # via the 2560 / 1920 concats
    assert not eng._stats_wanted(3, 1280)
    from storygen_amd import ops
    assert not ops.groupnorm_uses_pstats(16 * 16, 1280, 32) and ops.groupnorm_uses_pstats(16 * 16, 2560, 32)
    assert eng._stats_want[(2, 1280)] is True                              # cached per (level, width)


def test_float_reciprocal_quotients_used_by_the_groupnorm_kernels_are_exact():
    """gn_apply_wide_kernel (storygen_amd/csrc/norm.hip) replaces its integer divisions by (int)((e + 0.5f) * (1.0f / d)) — entry ->
    (channel, tile), channel -> group, pixel -> image row.  Exact for every operand the kernels can see (e < 2^20, any divisor)."""
    import numpy as np
    e = np.arange(0, 1 << 20, dtype=np.int64)
    ef = e.astype(np.float32) + np.float32(0.5)
    for d in list(range(1, 130)) + [144, 192, 256, 320, 576, 768, 1024, 2304, 2560, 4096, 9216]:
        q = (ef * (np.float32(1.0) / np.float32(d))).astype(np.int64)
        assert np.array_equal(q, e // d), d


def test_gather_is_identity_without_process_group():
    from storygen_amd.sampler import gather_latents
    x = torch.randn(1, 4, 8, 8)
    assert gather_latents(x) is x


# ------------------------------------------------------------------------------------------------ sampler layout
@pytest.mark.parametrize("N,R", [(1, 1), (1, 3), (2, 2), (3, 5)])
@pytest.mark.parametrize("mode", ["shared-zero", "dedup", "as-written"])
def test_reference_batch_layout_reproduces_the_as_written_context(N, R, mode):
    """StoryGenSampler._plan (which reference samples are computed, where their features are scattered, which context
    row each main-pass sample reads) against the loop as written: main sample (j, n) must see, in frame slot i, the
    feature of sample j*N + n of reference pass i = [zero_i | img_i | img_i] (pipeline.py:429-430,440-443)."""
    from storygen_amd.sampler import StoryGenSampler
    smp = object.__new__(StoryGenSampler)
    smp.N, smp.R, smp.dedup = N, R, mode != "as-written"
    units, hops, rows, groups, short = smp._plan("multi-image-condition", mode == "shared-zero")
    assert len(units) == {"shared-zero": N * (1 + R), "dedup": 2 * N * R, "as-written": 3 * N * R}[mode]
    assert short == (N if mode == "shared-zero" else 0)
    # short rows (the zero-image rows of shared-zero mode) hold ONE slot: as written their R slots are R copies of one feature map
    # and softmax over R copies of the same keys is softmax over one copy
    ctx = [[None] * (1 if row < short else R) for row in range(rows)]
    slot_of_unit = {}
    for src, step, row, slot, cnt in hops:
        for j in range(cnt):
            assert ctx[row][slot + j] is None, "a context slot is written twice"
            ctx[row][slot + j] = units[src + j * step]
            slot_of_unit[src + j * step] = (row if row < short else short + (row - short) * R + slot + j)
    assert all(c is not None for r in ctx for c in r), "a context slot is never written"
    # the reference batch is ordered like the context buffer: sample u belongs in flat slot u (direct harvest, no copy kernel)
    assert slot_of_unit == {u: u for u in range(len(units))}
    for row in range(short):
        ctx[row] = ctx[row] * R
    row_of = {}
    for q0, n, c0 in groups:
        for k in range(n):
            assert q0 + k not in row_of
            row_of[q0 + k] = c0 + k
    assert sorted(row_of) == list(range(3 * N))
    for j in range(3):
        for n in range(N):
            for i in range(R):
                kind, frame, sample = ctx[row_of[j * N + n]][i]
                assert kind == (0 if j == 0 else 1) and sample == n
                if kind == 1 or mode != "shared-zero":
                    assert frame == i


def test_tuned_tile_table_is_well_formed():
    """storygen_amd/tuning/mi355x_tiles.json: every override names a tile the kernels implement."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "storygen_amd", "tuning", "mi355x_tiles.json")
    with open(path) as f:
        t = json.load(f)
    tiles = {(256, 128), (128, 128), (256, 64), (128, 64), (64, 128), (64, 64)}
    assert t["tiles"], "empty tuning table"
    for sig, e in t["tiles"].items():
        assert sig[0] in "gc" and (e[0], e[1]) in tiles and 0 <= e[2] <= 64, (sig, e)
        if len(e) > 3:
            assert (e[0], e[1], e[3]) in {(256, 128, 4), (256, 128, 8), (128, 128, 2), (128, 128, 4)}


@pytest.mark.parametrize("ups", [False, True])
def test_running_counters_of_the_conv_mainloop_visit_the_same_operand_bases_as_the_division(ups):
    """gemm_conv.hip, mma_pipe_body: slab kt of the implicit-GEMM convolution reads tap (ky, kx) = divmod(kt // cpt, 3), channel block
    cc = kt % cpt, from element offset (ky * wp + kx) * lda + cc * 64 (nearest-2x upsampling: cc * 64, the tap enters per piece).  Since
    round 4 the kernel advances running counters instead of dividing per slab: this is the same recurrence in Python, checked against
    the closed form for every start slab (split-K slices start anywhere) — the derivation the device code relies on."""
    BK = 64
    for cpt, wp, lda in ((5, 66, 320), (10, 34, 640), (20, 18, 1280), (40, 10, 2560), (1, 66, 64)):
        nkt = 9 * cpt

        def closed(kt):
            tap, cc = divmod(kt, cpt)
            ky, kx = divmod(tap, 3)
            return (cc * BK if ups else (ky * wp + kx) * lda + cc * BK), ky, kx

        dx = (0 if ups else lda) - cpt * BK
        dy = 0 if ups else (wp - 3) * lda
        for kt0 in range(nkt):
            off, ky, kx = closed(kt0)
            cc = kt0 % cpt
            for kt in range(kt0, nkt):
                assert (off, ky, kx) == closed(kt), (cpt, kt0, kt)
                off += BK
                cc += 1
                if cc == cpt:
                    cc = 0
                    off += dx
                    kx += 1
                    if kx == 3:
                        kx, ky, off = 0, ky + 1, off + dy


@pytest.mark.parametrize("outlier", ["none", "first-channel", "first-tile", "one-row"])
def test_groupnorm_merge_arithmetic_in_fp32_against_fp64(outlier):
    """The arithmetic of gn_apply_wide_kernel's merge of the producers' per-(row tile, channel) partials (norm.hip, round 4), restated in
    numpy with every intermediate rounded to fp32: a thread owns one channel and every slots-th row tile of it and runs Chan's update about
    the mean of ITS first tile; a group sums its cpg x slots thread partials as mean first, then non-negative M2 terms.  Against the fp64
    statistics of the same data, with the outliers ADVICE r3 asked about (a channel / a tile / one pixel off by ~1e3 standard
    deviations): the variance keeps >= 5 digits.  (The kernel itself is compared with torch on the GPU: tests/test_kernels_gpu.py.)"""
    f32 = np.float32
    rng = np.random.default_rng(5)
    HW, C, G, rows = 1024, 320, 32, 64
    cpg, tiles, slots = C // G, HW // rows, 1024 // C
    x = rng.standard_normal((HW, C)).astype(f32) * f32(0.7) + rng.standard_normal(C).astype(f32)
    if outlier == "first-channel":
        x[:, 0::cpg] += f32(1000.0)
    elif outlier == "first-tile":
        x[:rows] += f32(1000.0)
    elif outlier == "one-row":
        x[5, 0::cpg] = f32(30000.0)
    xt = x.reshape(tiles, rows, C)
    s1 = xt.sum(1, dtype=f32)                       # what the producers' epilogues write (fp32 sums of fp32 values)
    s2 = (xt * xt).sum(1, dtype=f32)
    inv_rows, frows = f32(1.0 / rows), f32(rows)
    part = {}
    for slot in range(slots):
        for c in range(C):
            a_s = a_q = a_n = f32(0)
            K = None
            for t in range(slot, tiles, slots):
                me = f32(s1[t, c] * inv_rows)
                if K is None:
                    K = me
                dm = f32(me - K)
                a_s = f32(a_s + s1[t, c])
                a_n = f32(a_n + frows)
                a_q = f32(a_q + f32(max(f32(s2[t, c] - f32(s1[t, c] * me)), f32(0)) + f32(frows * f32(dm * dm))))
            if a_n > 0:
                dk = f32(f32(a_s / a_n) - K)
                part[slot, c] = (a_s, f32(max(f32(a_q - f32(a_n * f32(dk * dk))), f32(0))), a_n)
            else:
                part[slot, c] = (f32(0), f32(0), f32(0))
    ntot = f32(HW * cpg)
    for g in range(G):
        ent = [part[sl, g * cpg + k] for sl in range(slots) for k in range(cpg)]
        sm = f32(0)
        for a_s, _, _ in ent:
            sm = f32(sm + a_s)
        mean = f32(sm / ntot)
        m2 = f32(0)
        for a_s, q, n in ent:
            d = f32(f32(a_s / n) - mean) if n > 0 else f32(0)
            m2 = f32(m2 + f32(q + f32(n * f32(d * d))))
        var = float(m2 / ntot)
        ref = x[:, g * cpg:(g + 1) * cpg].astype(np.float64)
        assert abs(float(mean) - ref.mean()) <= 1e-5 * max(1.0, abs(ref.mean())), (g, outlier)
        assert abs(var - ref.var()) <= 2e-5 * ref.var(), (g, outlier, var, ref.var())


def test_tuned_split_k_is_taken_only_with_a_workspace_that_holds_it(monkeypatch):
    """The tile table is keyed by shape; its split-K counts (tools/tune_tiles.py measures them with a 256 MB workspace) must not reach a
    caller of the same shape that passes no split-K scratch — the training blocks' attn1.to_out at the mid level did exactly that with
    the table of round 4's closing run (sg_gemm_f16: "split_k=3 needs 3932160 workspace bytes, got 0")."""
    from types import SimpleNamespace
    from storygen_amd import ops
    monkeypatch.setitem(ops.TILE_TABLE, "g:test", (128, 64, 3))
    M, N = 1024, 1280
    for ws_bytes, want in ((0, 0), (3 * M * N * 4 - 1, 0), (3 * M * N * 4, 3)):
        d = SimpleNamespace(tile_m=0, tile_n=0, split_k=0, tile_waves=0, workspace_bytes=ws_bytes)
        ops._apply_tile(d, None, 0, "g:test", M * N)
        assert (d.tile_m, d.tile_n, d.split_k) == (128, 64, want), ws_bytes
    d = SimpleNamespace(tile_m=0, tile_n=0, split_k=4, tile_waves=0, workspace_bytes=0)
    ops._apply_tile(d, None, 4, "g:test", M * N)            # an explicit split_k from the caller bypasses the table
    assert (d.tile_m, d.tile_n, d.split_k) == (0, 0, 4)


def test_committed_bench_line_honours_the_contract():
    """profiles/r04s_bench.json (round 4's closing line; r03f_bench.json, r01g_final_bench.json: rounds 3 and 1) are verbatim `python bench.py` lines from the GPU box: guard the
    keys the driver and the judge read (a format regression in bench.py shows up here the next time the line is refreshed)."""
    import json
    for name in ("r04s_bench.json", "r03f_bench.json", "r01g_final_bench.json"):
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)
        line = [ln for ln in open(path) if ln.startswith("{")][-1]
        _check_bench_line(json.loads(line))


def _check_bench_line(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


@pytest.mark.parametrize("G", [1, 2, 4])
@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
def test_step_table_reference_scalars_follow_the_pass_schedule(G, stage):
    """sampler.step_table: row k must carry the reference-pass scalars of the step(s) whose reference pass is launched
    with main pass k (pipeline.py:414-427 for the values): step k (no overlap), step k+1 (overlap), or the G steps of the
    next group (ref_ahead = G).  Also the context-set arithmetic of the look-ahead schedule."""
    from storygen_amd.sampler import ctx_set_of_step, first_ctx_set_of_group, step_table
    from storygen_amd.scheduler import DDIMSchedule
    sch = DDIMSchedule()
    T, R, B = 10, 3, 3
    ts = sch.timesteps(T)
    units0 = [(0, 0, 0)] + [(1, i, 0) for i in range(R)]           # zero-image sample + the R prior frames (N = 1)
    U0 = len(units0)

    def want_ref(step):
        step = min(step, T - 1)
        ref_t = int(ts[step]) // 10
        tis = [ref_t * (R - i) if stage == "auto-regressive" else ref_t for i in range(R)]
        tt = [float(tis[i]) for _, i, _ in units0]
        cc = [c for _, i, _ in units0 for c in sch.add_noise_coef(tis[i])]
        return tt, cc

    for overlap in ((True,) if G > 1 else (False, True)):
        rows, row0 = step_table(sch, ts, T, units0, R, stage, B, G, overlap, 3.5, 7.5)
        U = G * U0
        assert len(rows) == T and all(len(r) == 3 * U + B + 6 for r in rows) and len(row0) == 3 * U + B + 6
        for k, row in enumerate(rows):
            first = (k // G + 1) * G if G > 1 else (k + 1 if overlap else k)
            for g in range(G):
                tt, cc = want_ref(first + g)
                assert row[g * U0:(g + 1) * U0] == tt
                assert row[U + B + 2 * g * U0:U + B + 2 * (g + 1) * U0] == pytest.approx(cc)
            assert row[U:U + B] == [float(ts[k])] * B
            assert row[3 * U + B:3 * U + B + 2] == [3.5, 7.5]
            assert row[3 * U + B + 2:] == pytest.approx(list(sch.step_coef(int(ts[k]), T)))
        for g in range(G):                                          # the very first pass / group
            tt, cc = want_ref(g)
            assert row0[g * U0:(g + 1) * U0] == tt and row0[U + B + 2 * g * U0:U + B + 2 * (g + 1) * U0] == pytest.approx(cc)
    # context sets: the G main passes of group j read exactly the sets its reference pass wrote, and the reference pass of
    # group j+1 (running beside them) writes the other half
    for k in range(6 * G):
        j = k // G
        mine = range(first_ctx_set_of_group(j, G), first_ctx_set_of_group(j, G) + G)
        nxt = range(first_ctx_set_of_group(j + 1, G), first_ctx_set_of_group(j + 1, G) + G)
        assert ctx_set_of_step(k, G) == mine[k % G]
        assert not set(mine) & set(nxt) and set(mine) | set(nxt) == set(range(2 * G))


def test_ref_ahead_needs_overlap_and_split_graphs_needs_a_graph():
    from storygen_amd.arch import build_arch
    from storygen_amd.sampler import StoryGenSampler
    from test_oracle_golden import _load
    arch = build_arch(_load("tiny")["config"])
    for kw in (dict(overlap=False), dict(ref_ahead=0), dict(split_graphs=True, use_graph=False), dict(stream_priority=True)):
        with pytest.raises(ValueError):
            StoryGenSampler(arch, None, "cpu", ref_ahead=kw.pop("ref_ahead", 2), weights=object(), **kw)
    smp = StoryGenSampler(arch, None, "cpu", ref_ahead=2, weights=object(), use_graph=False)     # the group schedule also runs eagerly
    assert smp.group and smp.ahead and not smp.split and not smp.overlap


def _stub_inputs(N, R, hw=4, S=5, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *sh: torch.randn(*sh, generator=g)                                         # noqa: E731
    return dict(latents=r(N, 4, hw, hw), noise=r(N, 4, hw, hw), image_prompts=r(R, N, 4, hw, hw), zero_prompt=r(N, 4, hw, hw),
                text=r(N, S, 8), uncond=r(N, S, 8), prev_text=r(R, N, S, 8), prev_uncond=r(1, N, S, 8).expand(R, N, S, 8).clone())


@pytest.mark.parametrize("stage", ["multi-image-condition", "auto-regressive"])
@pytest.mark.parametrize("N,R", [(1, 3), (2, 2)])
@pytest.mark.parametrize("kw", [dict(ref_ahead=2), dict(ref_ahead=5), dict(ref_ahead=2, short_rows=False), dict(ref_ahead=2, dedup=False)])
def test_group_schedule_is_the_step_by_step_trajectory(monkeypatch, stage, N, R, kw):
    """ref_ahead = G as ONE schedule unit per group (sampler._group_body: batched reference pass of the NEXT group, then this group's G
    main passes on parameter rows 0..G-1 and context sets parity*G + g) against the plain loop (reference pass, then main pass, per
    step), on a stand-in engine whose arithmetic depends on the context set, the parameter row, the harvest layout and the K / V^T
    pairing (tests/stub_engine.py): the latents after EVERY step must be bit-identical.  Covers the in-place group layout (one plan
    over the G context sets laid end to end) and the strided-copy fallback (short_rows=False / dedup=False)."""
    import stub_engine
    S = stub_engine.install(monkeypatch)
    from storygen_amd.arch import build_arch
    from test_oracle_golden import _load
    arch = build_arch(_load("tiny")["config"])
    inp = _stub_inputs(N, R)
    T = 10

    def run(**skw):
        smp = S.StoryGenSampler(arch, None, "cpu", N, 4, 4, R, 5, use_graph=False, weights=object(), time_tables=False, **skw)
        smp.prepare(inp, T, stage, 7.5, 3.5)
        tr = []
        smp.run(trace=tr)
        return smp, tr

    base_kw = {k: v for k, v in kw.items() if k != "ref_ahead"}
    ref, want = run(**base_kw)
    smp, got = run(**kw)
    G = kw["ref_ahead"]
    assert smp.group and len(got) == len(want) == T
    assert smp.group_direct == ("short_rows" not in kw or stage == "auto-regressive")
    for k in range(T):
        assert torch.equal(got[k], want[k]), f"step {k}"
    # the work: one batched reference call per group, each G x the per-step batch — the primer + one look-ahead per group BUT the
    # last (round 6: the last group's graph runs no look-ahead pass; the reference's loop runs none after its last step either)
    refs = [b for kind, b in smp.ref.calls if kind == "ref"]
    assert refs == [G * ref.U0] * (T // G) and smp.U == G * ref.U0
    assert smp._last_lookahead_at() == T - G
    assert [b for kind, b in smp.main.calls if kind == "main"] == [3 * N] * T
    with pytest.raises(ValueError):                                                       # groups of G: the evaluation count must divide
        smp.prepare(inp, T + 1, stage, 7.5, 3.5)


def test_bench_launcher_path_world2_gloo_dry_run():
    """bench.py exactly as the driver launches it for N = 2 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps 20 --warmup 5`) with --dry-run: gloo instead of RCCL and a
    stand-in engine instead of the GPU, everything else — ranks from the environment, the default group schedule (ref_ahead 5),
    barrier + max-over-ranks timing (bench.timed_steps), the one all-gather and the per-rank-distinct check (bench.gather_and_check),
    ONE JSON line from rank 0 — is the code the real run executes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and "DRY RUN" in d["metric"]          # can never be read as a measurement
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["latents_gathered"] == 2 and d["latents_distinct_per_rank"] and d["latents_finite"]
    assert d["config"]["ref_ahead"] == 5 and d["config"]["warmup_run"] == 5


@pytest.mark.parametrize("steps,warmup,G", [(3, 1, 3), (7, 2, 1), (8, 3, 4), (10, 3, 5)])
def test_bench_default_group_size_follows_the_step_count(steps, warmup, G):
    """`bench.py --steps K --warmup W` for K the driver may pick: the default group size is the largest G <= 5 dividing K, the warm-up is
    rounded up to whole groups and the schedule to a multiple of G (K = 3 crashed on a 50-step table before the dry run existed)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run", "--steps", str(steps), "--warmup", str(warmup)],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["ref_ahead"] == G and d["config"]["warmup_run"] == -(-warmup // G) * G and d["steps"] == steps


def test_bench_argument_parser_builds():
    """`python bench.py --help` must exit 0: a duplicated add_argument (it happened) would break every driver run before any GPU work."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    assert "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout


def _all_tensors(obj, prefix=""):
    """Every tensor held by an EngineWeights (recursively through dicts / tuples / the per-layer records), with a path name."""
    import torch as _t
    from storygen_amd.engine import _Resnet, _Xf
    if _t.is_tensor(obj):
        yield prefix, obj
    elif isinstance(obj, dict):
        for k, v in obj.items():
            yield from _all_tensors(v, f"{prefix}.{k}")
    elif isinstance(obj, (tuple, list)):
        for i, v in enumerate(obj):
            yield from _all_tensors(v, f"{prefix}[{i}]")
    elif isinstance(obj, (_Resnet, _Xf)):
        for k in obj.__slots__:
            if k != "spec" and getattr(obj, k, None) is not None:
                yield from _all_tensors(getattr(obj, k), f"{prefix}.{k}")


@pytest.mark.parametrize("module", ["attn1", "attn3"])
def test_in_place_refresh_of_the_trainable_module_equals_a_fresh_repack(module):
    """EngineWeights.refresh_attn{1,3}_ (what the drop-in UNet calls after an optimizer step of stage 1 / stage 2): after changing every
    parameter of that module, the refreshed weights equal a fresh re-pack of the whole checkpoint, tensor for tensor, and no tensor was
    re-allocated (captured hipGraphs hold their addresses).  Pure packing logic: runs on the CPU device."""
    from storygen_amd.arch import build_arch, load_config
    from storygen_amd.engine import EngineWeights
    from storygen_amd.synth import synthetic_state_dict
    cfg = load_config(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                           up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, attention_head_dim=2,
                           norm_num_groups=8, sample_size=64))
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, 3)
    w = EngineWeights(arch, sd, "cpu")
    ptrs = {n: t.data_ptr() for n, t in _all_tensors(w.__dict__)}
    g = torch.Generator().manual_seed(1)
    sd2 = {k: (v + 0.05 * torch.randn(v.shape, generator=g) if f".{module}." in k else v) for k, v in sd.items()}
    assert sum(f".{module}." in k for k in sd2) == 5 * len(arch.feature_keys)
    getattr(w, f"refresh_{module}_")(sd2)
    fresh = dict(_all_tensors(EngineWeights(arch, sd2, "cpu").__dict__))
    mine = dict(_all_tensors(w.__dict__))
    assert set(mine) == set(fresh)
    for n, t in mine.items():
        assert torch.equal(t, fresh[n]), n
        assert t.data_ptr() == ptrs[n], n


def test_dropin_unet_tag_routes_attn1_changes_to_the_in_place_refresh():
    from storygen_amd.model import UNet2DConditionModel
    cfg = dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
               up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, attention_head_dim=2, norm_num_groups=8, sample_size=64)
    unet = UNet2DConditionModel.from_config(cfg)
    w = unet._engine_weights()
    calls = []
    w.refresh_attn1_ = lambda sd, prefixes=None: calls.append(("attn1", sorted(prefixes)))
    w.reload_ = lambda sd: calls.append(("reload", None))
    with torch.no_grad():
        for n, p in unet.named_parameters():
            if ".attn1." in n and n.startswith("mid_block"):
                p.add_(1.0)
    assert unet._engine_weights() is w and calls == [("attn1", ["mid_block.attentions.0"])]
    with torch.no_grad():
        next(p for n, p in unet.named_parameters() if "conv_in" in n).add_(1.0)
    unet._engine_weights()
    assert calls[-1] == ("reload", None)


def test_layernorm_fold_is_the_same_linear_map():
    """repack.fold_layernorm (preparation for folding LayerNorm into the following GEMM's epilogue): the folded form on fp16 operands stays
    within ~1.5x of the current two-step fp16 path's distance from fp32, also when the row mean is several standard deviations."""
    from storygen_amd.repack import fold_layernorm
    g = torch.Generator().manual_seed(0)
    M, K, N = 256, 320, 96
    W, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    gamma, beta = 1.0 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    for offset in (0.0, 3.0):
        x = torch.randn(M, K, generator=g) + offset
        ref = F.layer_norm(x, (K,), gamma, beta, 1e-5) @ W.t() + b
        cur = F.layer_norm(x, (K,), gamma, beta, 1e-5).half().float() @ W.half().float().t() + b
        wf, c, d = fold_layernorm(W, b, gamma, beta)
        mu, rstd = x.mean(1, keepdim=True), (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        fold = rstd * (x.half().float() @ wf.float().t()) - rstd * mu * c[None] + d[None]
        e_cur = float((cur - ref).norm() / ref.norm())
        e_fold = float((fold - ref).norm() / ref.norm())
        assert e_fold < 1e-3 * (1.0 + offset) and e_fold < 2.0 * (1.0 + offset) * e_cur, (offset, e_cur, e_fold)


def test_default_workspace_is_lent_and_returned():
    """ops.default_workspace (round 6): gemm / conv3x3 calls that pass no `workspace=` borrow the current default — the training classes'
    split-K scratch — and an explicit argument still wins; nested scopes restore the outer buffer; nothing is left set afterwards."""
    from storygen_amd import ops
    from storygen_amd._lib import GemmDesc
    a, b = torch.empty(1024, dtype=torch.uint8), torch.empty(4096, dtype=torch.uint8)

    def ws_of(explicit):
        d = GemmDesc()
        ops._ws(explicit, d)
        return d.workspace or 0, d.workspace_bytes

    assert ops.DEFAULT_WORKSPACE is None and ws_of(None) == (0, 0)
    with ops.default_workspace(a):
        assert ws_of(None) == (a.data_ptr(), 1024)
        assert ws_of(b) == (b.data_ptr(), 4096)
        with ops.default_workspace(b):
            assert ws_of(None) == (b.data_ptr(), 4096)
        assert ws_of(None) == (a.data_ptr(), 1024)
        with ops.default_workspace(None):                      # a caller that must not borrow (e.g. concurrent streams)
            assert ws_of(None) == (0, 0)
    assert ops.DEFAULT_WORKSPACE is None and ws_of(None) == (0, 0)


def test_nonfinite_flag_of_a_gradient_list():
    """train._nonfinite_flag (round 6: the per-step finiteness check is part of the captured training graph): > 0 iff any entry of any
    tensor is inf / nan; the tensors are left as they are."""
    from storygen_amd.train import _nonfinite_flag
    g = [torch.randn(7, 5), torch.randn(3), torch.zeros(2, 2)]
    keep = [t.clone() for t in g]
    assert float(_nonfinite_flag(g)) == 0.0
    g[1][2] = float("inf")
    assert float(_nonfinite_flag(g)) > 0.0
    g[1][2] = 0.5
    g[2][1, 1] = float("nan")
    assert float(_nonfinite_flag(g)) > 0.0
    g[2][1, 1] = 0.0
    keep[1][2] = 0.5
    assert all(torch.equal(x, y) for x, y in zip(g, keep))


def test_library_is_built_without_packed_fp32_arithmetic():
    """the one run-to-run difference this project saw was one half of a `v_pk_add_f32 ... op_sel` losing a term; the kernels are
    built with -fno-slp-vectorize (for every translation unit, not per file)."""
    from storygen_amd import build
    assert "-fno-slp-vectorize" in build.FLAGS
    assert all("-fslp-vectorize" not in f for flags in build.EXTRA_FLAGS.values() for f in flags)
