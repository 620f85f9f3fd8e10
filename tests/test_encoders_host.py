"""Host logic of storygen_amd/encoders.py (weight folds, layouts, buffer reuse, the shifted-view stride-2 convolution, the ragged
attention path) on CPU: the engines run over tests/ops_emulation.py's torch emulation of the kernels and are compared with the oracle.
The real kernels run the same engine code in tests/test_encoders_gpu.py."""
import os

import pytest
import torch

from oracle import encoders_oracle as eo
from tests.ops_emulation import patched_ops

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _h(sd):
    """The engine rounds weights to fp16; give the oracle the same rounded weights so that only activation rounding differs."""
    return {k: v.half().float() for k, v in sd.items()}


@pytest.mark.parametrize("hw", [(16, 16), (12, 20)])
def test_vae_engine_host_logic_matches_oracle(hw):
    from storygen_amd.encoders import VaeEngine
    sd = _h(eo.vae_random_state(block_out=(64, 128), layers_per_block=1, seed=3))
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 3, *hw, generator=g)
    want = eo.vae_encode_moments(sd, x)
    with patched_ops():
        eng = VaeEngine(sd, "cpu")
        mean, logvar = eng.encode(x)
        noise = torch.randn(mean.shape, generator=g)
        z = eng.sample(mean, logvar, noise, 0.18215)
        img = eng.decode(z / 0.18215)
        mode = eng.sample(mean, logvar, None)
    assert rel(torch.cat([mean, logvar], 1), want) < 5e-3
    zw = eo.gaussian_sample(want, noise) * 0.18215
    assert rel(z, zw) < 5e-3
    assert rel(mode, want[:, :4]) < 5e-3
    assert tuple(img.shape) == (2, 3, *hw)
    assert rel(img, eo.vae_decode(sd, zw / 0.18215)) < 1e-2


def test_vae_engine_sd15_shapes_one_resnet_deep():
    """The reference's VAE config (ckpt/stable-diffusion-v1-5/vae/config.json channel plan 128/256/512/512) on an 8x8 latent."""
    from storygen_amd.encoders import VaeEngine
    sd = _h(eo.vae_random_state(block_out=(128, 256, 512, 512), layers_per_block=1, seed=1))
    x = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    want = eo.vae_encode_moments(sd, x)
    with patched_ops():
        eng = VaeEngine(sd, "cpu")
        mean, logvar = eng.encode(x)
        img = eng.decode(mean)
    assert tuple(mean.shape) == (1, 4, 4, 4)
    assert rel(torch.cat([mean, logvar], 1), want) < 5e-3
    assert rel(img, eo.vae_decode(sd, want[:, :4])) < 1e-2


def test_vae_engine_rejects_bad_shapes():
    from storygen_amd.encoders import VaeEngine
    sd = eo.vae_random_state(block_out=(64, 128), layers_per_block=1)
    with patched_ops():
        eng = VaeEngine(sd, "cpu")
        with pytest.raises(ValueError):
            eng.encode(torch.zeros(1, 3, 15, 16))
        with pytest.raises(ValueError):
            eng.encode(torch.zeros(1, 4, 16, 16))
        with pytest.raises(ValueError):
            eng.decode(torch.zeros(1, 3, 4, 4))


def test_clip_engine_host_logic_matches_transformers_golden():
    from storygen_amd.encoders import ClipTextEngine
    gold = torch.load(os.path.join(GOLDEN, "clip_text_tiny.pt"), weights_only=True)
    with patched_ops():
        eng = ClipTextEngine(gold["state_dict"], "cpu", heads=gold["heads"])
        hidden, pooled = eng(gold["input_ids"])
        with pytest.raises(IndexError):
            eng(torch.full((1, 77), 1000))
    assert rel(hidden, gold["last_hidden_state"]) < 3e-3
    assert rel(pooled, gold["pooled"]) < 3e-3


def test_clip_engine_padding_mask():
    from storygen_amd.encoders import ClipTextEngine
    gold = torch.load(os.path.join(GOLDEN, "clip_text_tiny.pt"), weights_only=True)
    ids = gold["input_ids"][:1, :24]
    mask = torch.ones(1, 24)
    mask[:, 20:] = 0
    want, _ = eo.clip_text_forward({k: v.float() for k, v in gold["state_dict"].items()}, ids, heads=gold["heads"], attention_mask=mask)
    with patched_ops():
        eng = ClipTextEngine(gold["state_dict"], "cpu", heads=gold["heads"])
        hidden, _ = eng(ids, attention_mask=mask)
    assert rel(hidden, want) < 3e-3


def test_dropin_classes_adopt_third_party_modules_and_checkpoints(tmp_path):
    """CLIPTextModel.from_torch on a transformers module (un-prefixed names in transformers 5, `text_model.` in 4.x), the reference's
    CLIP/config.json (a full CLIPModel config), AutoencoderKL from the reference's vae/config.json, save/load round trips."""
    import json
    transformers = pytest.importorskip("transformers")
    from storygen_amd.model import AutoencoderKL, CLIPTextModel
    cfg = transformers.CLIPTextConfig(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2,
                                      max_position_embeddings=77, bos_token_id=0, eos_token_id=2)
    tm = transformers.CLIPTextModel(cfg).eval()
    mine = CLIPTextModel.from_torch(tm)
    assert mine.config.hidden_size == 64 and mine.config.num_attention_heads == 2
    for k, v in tm.state_dict().items():
        if not k.endswith("position_ids"):
            kk = k if k.startswith("text_model.") else "text_model." + k
            assert torch.equal(mine.state_dict()[kk], v)
    with patched_ops():
        from storygen_amd.encoders import ClipTextEngine
        ids = torch.randint(0, 299, (2, 77))
        hidden, _ = ClipTextEngine(mine.state_dict(), "cpu", heads=2)(ids)
    with torch.no_grad():
        assert rel(hidden, tm(ids)[0]) < 3e-3
    mine.save_pretrained(str(tmp_path / "te"), safe_serialization=True)
    again = CLIPTextModel.from_pretrained(str(tmp_path), subfolder="te", torch_dtype=torch.float16)
    assert again.dtype == torch.float16 and set(again.state_dict()) == set(mine.state_dict())
    with pytest.raises(RuntimeError):
        again(torch.zeros(1, 77, dtype=torch.long))                 # no CPU path
    with pytest.raises(NotImplementedError):
        again.requires_grad_(True)
    full = {"text_config": {"hidden_size": 768, "num_attention_heads": 12, "num_hidden_layers": 1, "intermediate_size": 3072}, "vision_config": {}}
    assert CLIPTextModel(full).config.hidden_size == 768
    vae = AutoencoderKL(block_out_channels=(64, 128), down_block_types=("DownEncoderBlock2D",) * 2, up_block_types=("UpDecoderBlock2D",) * 2)
    assert set(vae.state_dict()) == set(eo.vae_random_state(block_out=(64, 128), layers_per_block=1))
    vae.save_pretrained(str(tmp_path / "vae"))
    v2 = AutoencoderKL.from_pretrained(str(tmp_path), subfolder="vae")
    assert all(torch.equal(v2.state_dict()[k], v) for k, v in vae.state_dict().items())
    assert json.load(open(tmp_path / "vae" / "config.json"))["_class_name"] == "AutoencoderKL"
    with pytest.raises(RuntimeError):
        v2.load_state_dict({k: v for k, v in list(vae.state_dict().items())[:-1]})
    with pytest.raises(TypeError):
        AutoencoderKL(block_out=(64,))
