"""SURVEY §8 f3 on the GPU: the CLIP text encoder and the AutoencoderKL on the HIP kernels, against oracle/encoders_oracle.py
(CLIP pinned to transformers through tests/golden/clip_text_tiny.pt; the VAE restated from diffusers 0.13.1, parity unpinned), and
the drop-in pipeline end to end with both HIP encoders in place (/root/reference/model/pipeline.py:137,183,198-205,392,401)."""
import os
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_l2
from oracle import encoders_oracle as eo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F16, F32 = torch.float16, torch.float32


def _h(sd):
    return {k: v.half().float() for k, v in sd.items()}


# ------------------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("M,N", [(100, 64), (37, 60), (512, 4096), (8, 5000)])
def test_softmax_rows(gpu, M, N):
    from storygen_amd import ops
    torch.manual_seed(N)
    N8 = (N + 7) & ~7
    s_full = (torch.randn(M, N8, device=gpu) * 6).contiguous()
    p = torch.full((M, N8), 7.0, dtype=F16, device=gpu)
    ops.softmax_rows(s_full[:, :N], p, 0.37)
    want = torch.softmax(s_full[:, :N].double() * 0.37, -1)
    assert rel_l2(p[:, :N].cpu(), want.cpu()) < 1e-3
    assert float(p[:, N:].abs().max()) == 0.0 if N8 > N else True
    assert abs(float(p[:, :N].float().sum(-1).mean()) - 1.0) < 1e-3


@pytest.mark.parametrize("B,T,H,D,causal,bias", [(2, 77, 12, 64, True, False), (1, 24, 4, 32, True, True), (3, 128, 2, 64, False, True),
                                                  (1, 1, 1, 8, True, False)])
def test_attention_small(gpu, B, T, H, D, causal, bias):
    from storygen_amd import ops
    torch.manual_seed(T + D)
    qkv = torch.randn(B, T, 3 * H * D, device=gpu).half()
    q, k, v = qkv[..., :H * D], qkv[..., H * D:2 * H * D], qkv[..., 2 * H * D:]
    kb = None
    if bias:
        kb = torch.zeros(B, T, device=gpu)
        kb[:, T - max(1, T // 4):] = torch.finfo(F32).min
        kb[:, 0] = 0.0
    out = torch.empty(B, T, H * D, dtype=F16, device=gpu)
    scale = D ** -0.5
    ops.attention_small(q, k, v, out, H, scale, causal, kb)
    qh, kh, vh = (t.double().view(B, T, H, D).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        s = s + torch.full((T, T), float("-inf"), device=gpu, dtype=torch.float64).triu(1)
    if kb is not None:
        s = s + kb.double()[:, None, None, :]
    want = (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, T, H * D)
    assert rel_l2(out.cpu(), want.cpu()) < 1.5e-3


def test_act_embed_and_gaussian_kernels(gpu):
    from storygen_amd import ops
    torch.manual_seed(0)
    x = (torch.randn(77, 3072, device=gpu) * 3).half()
    want = x.double() * torch.sigmoid(1.702 * x.double())
    y = x.clone()
    ops.act_rows(y, ops.ACT_QUICK_GELU)
    assert rel_l2(y.cpu(), want.cpu()) < 1e-3
    buf = torch.zeros(10, 96, dtype=F16, device=gpu)
    view = buf[:, 8:72]
    view.copy_(x[:10, :64])
    ops.act_rows(view, ops.ACT_GELU)
    assert rel_l2(view.cpu(), torch.nn.functional.gelu(x[:10, :64].float()).cpu()) < 1e-3
    assert float(buf[:, :8].abs().max()) == 0.0 and float(buf[:, 72:].abs().max()) == 0.0
    tok, pos = torch.randn(1000, 128, device=gpu), torch.randn(77, 128, device=gpu)
    ids = torch.randint(0, 1000, (3 * 77,), device=gpu)
    out = torch.empty(3 * 77, 128, device=gpu)
    ops.embed_tokens(ids, tok, pos, out, 77)
    assert torch.equal(out, tok[ids] + pos[torch.arange(3 * 77, device=gpu) % 77])
    mean, logvar, noise = torch.randn(2, 4, 8, 8, device=gpu), torch.randn(2, 4, 8, 8, device=gpu) * 20, torch.randn(2, 4, 8, 8, device=gpu)
    z = torch.empty_like(mean)
    ops.gaussian_sample(mean, logvar, noise, z, 0.18215)
    want = (mean + torch.exp(0.5 * logvar.clamp(-30, 20)) * noise) * 0.18215
    assert rel_l2(z.cpu(), want.cpu()) < 1e-5
    ops.gaussian_sample(mean, None, None, z, 2.0)
    assert torch.equal(z, mean * 2.0)


# --------------------------------------------------------------------------------------------------------------- CLIP
def test_clip_text_engine_vs_transformers_golden(gpu):
    from storygen_amd.encoders import ClipTextEngine
    gold = torch.load(os.path.join(GOLDEN, "clip_text_tiny.pt"), weights_only=True)
    eng = ClipTextEngine(gold["state_dict"], gpu, heads=gold["heads"])
    hidden, pooled = eng(gold["input_ids"])
    e1, e2 = rel_l2(hidden.cpu(), gold["last_hidden_state"]), rel_l2(pooled.cpu(), gold["pooled"])
    print(f"CLIP tiny vs transformers: hidden {e1:.2e} pooled {e2:.2e}")
    assert e1 < 3e-3 and e2 < 3e-3
    with pytest.raises(IndexError):
        eng(torch.full((1, 77), 1000))


def test_clip_text_engine_sd15_size_vs_oracle(gpu):
    """The text encoder the reference loads (clip-vit-large-patch14 text tower: 12 layers x 768, 12 heads, 49408 tokens), random
    weights, the reference's batch of [uncond, prompt, 3 previous prompts]; with and without a padding mask."""
    from storygen_amd.encoders import ClipTextEngine
    sd = _h(eo.clip_text_random_state(seed=5))
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1, 49406, (5, 77), generator=g)
    ids[:, 0] = 49406
    ids[:, 30:] = 49407
    eng = ClipTextEngine(sd, gpu, heads=12)
    hidden, pooled = eng(ids)
    want, wpool = eo.clip_text_forward(sd, ids, heads=12)
    e1, e2 = rel_l2(hidden.cpu(), want), rel_l2(pooled.cpu(), wpool)
    print(f"CLIP-L text tower vs oracle: hidden {e1:.2e} pooled {e2:.2e}")
    assert e1 < 5e-3 and e2 < 5e-3
    mask = torch.ones(5, 77)
    mask[:, 31:] = 0
    hm, _ = eng(ids, attention_mask=mask)
    wm, _ = eo.clip_text_forward(sd, ids, heads=12, attention_mask=mask)
    assert rel_l2(hm.cpu(), wm) < 5e-3


# ---------------------------------------------------------------------------------------------------------------- VAE
@pytest.mark.parametrize("hw", [(16, 16), (12, 20)])
def test_vae_engine_small_vs_oracle(gpu, hw):
    from storygen_amd.encoders import VaeEngine
    sd = _h(eo.vae_random_state(block_out=(64, 128), layers_per_block=1, seed=3))
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, *hw, generator=g)
    want = eo.vae_encode_moments(sd, x)
    eng = VaeEngine(sd, gpu)
    mean, logvar = eng.encode(x.to(gpu))
    e = rel_l2(torch.cat([mean, logvar], 1).cpu(), want)
    noise = torch.randn(mean.shape, generator=g)
    z = eng.sample(mean, logvar, noise, 0.18215)
    zw = eo.gaussian_sample(want, noise) * 0.18215
    img = eng.decode(zw.to(gpu) / 0.18215)
    e3 = rel_l2(img.cpu(), eo.vae_decode(sd, zw / 0.18215))
    print(f"VAE (64,128) {hw}: moments {e:.2e} sample {rel_l2(z.cpu(), zw):.2e} decode {e3:.2e}")
    assert e < 5e-3 and rel_l2(z.cpu(), zw) < 5e-3 and e3 < 1e-2


def test_vae_engine_sd15_config_vs_oracle(gpu):
    """ckpt/stable-diffusion-v1-5/vae/config.json (128/256/512/512, 2 layers per block) on 128x128 images: encode of the
    [zero image, previous frame] pair the pipeline encodes (model/pipeline.py:390-402) and decode of a 16x16 latent (:198-205)."""
    from storygen_amd.encoders import VaeEngine
    sd = _h(eo.vae_random_state(seed=11))
    g = torch.Generator().manual_seed(2)
    x = torch.cat([torch.zeros(1, 3, 128, 128), torch.rand(1, 3, 128, 128, generator=g)])
    eng = VaeEngine(sd, gpu)
    mean, logvar = eng.encode(x.to(gpu))
    want = eo.vae_encode_moments(sd, x)
    e1 = rel_l2(torch.cat([mean, logvar], 1).cpu(), want)
    z = torch.randn(1, 4, 16, 16, generator=g)
    img = eng.decode(z.to(gpu))
    e2 = rel_l2(img.cpu(), eo.vae_decode(sd, z))
    print(f"VAE SD-1.5 config 128x128: moments {e1:.2e} decode {e2:.2e}")
    assert tuple(mean.shape) == (2, 4, 16, 16) and tuple(img.shape) == (1, 3, 128, 128)
    assert e1 < 1e-2 and e2 < 1e-2


def test_vae_decode_512_runs_and_is_finite(gpu):
    """Full-size decode (64x64 latent -> 512x512 image), the call at the end of every pipeline run: finite, deterministic, timed."""
    from storygen_amd.encoders import VaeEngine
    eng = VaeEngine(_h(eo.vae_random_state(seed=11)), gpu)
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(0)).to(gpu)
    a = eng.decode(z)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    b = eng.decode(z)
    t1.record()
    torch.cuda.synchronize()
    print(f"VAE decode 64x64 latent -> 512x512: {t0.elapsed_time(t1):.1f} ms")
    assert tuple(a.shape) == (1, 3, 512, 512) and bool(torch.isfinite(a).all()) and torch.equal(a, b)


# ----------------------------------------------------------------------------------------------------------- pipeline
class _Tok:
    model_max_length = 77

    def __init__(self, prompts, vocab):
        g = torch.Generator().manual_seed(7)
        self.rows = {p: torch.randint(1, vocab - 1, (77,), generator=g) for p in prompts}

    def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None):
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        ids = torch.stack([self.rows[p] for p in prompts])
        return SimpleNamespace(input_ids=ids, attention_mask=torch.ones_like(ids))


def test_pipeline_with_hip_vae_and_clip_matches_oracle_encoders(gpu):
    """StableDiffusionPipeline.__call__ with the HIP AutoencoderKL and CLIPTextModel in place of the torch modules, against the same
    pipeline fed by the oracle's CLIP / VAE (stand-ins that return the oracle's outputs): images must agree."""
    from storygen_amd.arch import SD15_CONFIG
    from storygen_amd.model import AutoencoderKL, CLIPTextModel, StableDiffusionPipeline, UNet2DConditionModel
    from storygen_amd.scheduler import DDIMSchedule
    R, hw = 2, 32                       # the 4-level UNet needs a latent of at least 32x32 (8 tokens at the deepest attention)
    unet = UNet2DConditionModel.from_config(SD15_CONFIG).to(gpu, F16).eval()
    boc = (64, 64, 128, 128)
    vae = AutoencoderKL(block_out_channels=boc, down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4,
                        layers_per_block=1, seed=4).to(gpu, F16)
    clip = CLIPTextModel(dict(vocab_size=1000, hidden_size=768, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=12), seed=5)
    clip = clip.to(gpu, F16)
    prompts = ["", "main"] + [f"prev{i}" for i in range(R)]
    tok = _Tok(prompts, 1000)
    frames = torch.rand(1, R, 3, 8 * hw, 8 * hw, generator=torch.Generator().manual_seed(1))
    lat0 = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(2)).to(gpu, F16)

    vsd = {k: v.float().cpu() for k, v in vae.state_dict().items()}
    csd = {k: v.float().cpu() for k, v in clip.state_dict().items()}

    class OracleClip(torch.nn.Module):
        config = SimpleNamespace()

        def forward(self, input_ids, attention_mask=None):
            return (eo.clip_text_forward(csd, input_ids.cpu(), heads=12)[0].to(gpu, F16),)

    class OracleVae:
        config = SimpleNamespace(block_out_channels=boc)

        def encode(self, x):
            m = eo.vae_encode_moments(vsd, x.float().cpu())

            def sample():
                noise = torch.randn(m[:, :4].shape, device=gpu, dtype=F32)      # the same draw DiagonalGaussianDistribution.sample makes
                return eo.gaussian_sample(m, noise.cpu()).to(gpu, x.dtype)
            return SimpleNamespace(latent_dist=SimpleNamespace(sample=sample))

        def decode(self, z):
            return SimpleNamespace(sample=eo.vae_decode(vsd, z.float().cpu()).to(gpu, z.dtype))

    def run(v, c):
        pipe = StableDiffusionPipeline(vae=v, text_encoder=c, tokenizer=tok, unet=unet, scheduler=DDIMSchedule())
        pipe.set_progress_bar_config(disable=True)
        torch.manual_seed(1234)
        return pipe(stage="multi-image-condition", prompt="main", image_prompt=frames, prev_prompt=prompts[2:], height=8 * hw, width=8 * hw,
                    num_inference_steps=2, guidance_scale=7.5, image_guidance_scale=3.5, latents=lat0.clone(), output_type="np").images

    hip = run(vae, clip)
    ref = run(OracleVae(), OracleClip())
    err = rel_l2(torch.as_tensor(hip), torch.as_tensor(ref))
    print(f"pipeline images, HIP encoders vs oracle encoders: rel-L2 {err:.2e}")
    assert hip.shape == (1, 8 * hw, 8 * hw, 3) and err < 2e-2


def test_dropin_encoder_classes_round_trip(gpu, tmp_path):
    from storygen_amd.model import AutoencoderKL, CLIPTextModel
    vae = AutoencoderKL(block_out_channels=(64, 128), down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2)
    vae.save_pretrained(str(tmp_path / "vae"), safe_serialization=True)
    v2 = AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae", torch_dtype=F16).to(gpu)
    x = torch.rand(1, 3, 16, 16, device=gpu, dtype=F16)
    d = v2.encode(x).latent_dist
    z = d.sample(torch.Generator().manual_seed(0))
    assert z.dtype == F16 and tuple(z.shape) == (1, 4, 8, 8) and tuple(d.mode().shape) == (1, 4, 8, 8)
    assert tuple(v2.decode(z).sample.shape) == (1, 3, 16, 16)
    want = eo.vae_encode_moments({k: v.float().cpu() for k, v in v2.state_dict().items()}, x.float().cpu())
    assert rel_l2(d.mean.float().cpu(), want[:, :4]) < 5e-3
    clip = CLIPTextModel(dict(vocab_size=500, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2))
    clip.save_pretrained(str(tmp_path / "te"))
    c2 = CLIPTextModel.from_pretrained(str(tmp_path), subfolder="te").to(gpu)
    out = c2(torch.randint(0, 500, (2, 77)))
    assert tuple(out[0].shape) == (2, 77, 64) and tuple(out.pooler_output.shape) == (2, 64) and out[0].dtype == F32
