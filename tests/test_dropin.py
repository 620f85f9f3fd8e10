"""The drop-in boundary that needs no GPU: module-tree names, state-dict contract, config behaviour,
checkpoint round trips, SD-1.5 adoption (`load_SDM_state_dict`), error behaviour."""
import os

import pytest
import torch

TINY = dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
            up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), cross_attention_dim=48, sample_size=64)


@pytest.fixture(scope="module")
def tiny():
    from storygen_amd.model import UNet2DConditionModel
    torch.manual_seed(0)
    return UNet2DConditionModel(**TINY)


def test_state_dict_is_the_reference_checkpoint_contract(tiny):
    from storygen_amd.arch import build_arch, param_shapes
    want = param_shapes(build_arch(TINY))
    sd = tiny.state_dict()
    assert list(sd) == list(want)
    assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    # PyTorch layouts: conv [Cout, Cin, kh, kw], linear [out, in]
    assert sd["conv_in.weight"].shape == (32, 4, 3, 3)
    assert sd["down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape == (32, 48)
    assert sd["down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight"].shape == (256, 32)


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference tree only exists in the build container")
def test_state_dict_matches_the_reference_module_tree():
    """Keys and shapes against the reference's own UNet2DConditionModel (imported verbatim on the oracle's shim)."""
    from oracle.ref_runner import import_reference
    from storygen_amd.model import UNet2DConditionModel
    RefUNet, _, _ = import_reference()
    ref = RefUNet(**TINY)
    mine = UNet2DConditionModel(**TINY)
    rs, ms = ref.state_dict(), mine.state_dict()
    assert set(rs) == set(ms)
    assert all(rs[k].shape == ms[k].shape for k in rs)
    ref_attn3 = sorted(n for n, _ in ref.named_modules() if n.endswith("attn3"))
    my_attn3 = sorted(n for n, _ in mine.named_modules() if n.endswith("attn3"))
    assert ref_attn3 == my_attn3 and len(my_attn3) == 6
    # train_StorySalon_stage2.py:170-177 — select attn3 modules by name and unfreeze their parameters
    mine.requires_grad_(False)
    for name, module in mine.named_modules():
        if name.endswith("attn3"):
            for p in module.parameters():
                p.requires_grad = True
    n_train = sum(p.numel() for p in mine.parameters() if p.requires_grad)
    n_ref = sum(p.numel() for n, m in ref.named_modules() if n.endswith("attn3") for p in m.parameters())
    assert n_train == n_ref > 0


def test_config_and_attributes(tiny):
    assert tiny.config.sample_size == 64 and tiny.config["cross_attention_dim"] == 48
    assert tiny.config.block_out_channels == (32, 64)
    assert tiny.in_channels == 4 and tiny.sample_size == 64
    assert tiny.dtype == torch.float32 and tiny.device.type == "cpu"
    with pytest.raises((AttributeError, TypeError)):
        tiny.config.sample_size = 3
    assert tiny.half().dtype == torch.float16
    tiny.float()


def test_unknown_block_types_raise_like_the_reference():
    from storygen_amd.model import UNet2DConditionModel
    with pytest.raises(ValueError, match="does not exist"):
        UNet2DConditionModel(**dict(TINY, down_block_types=("FooBlock2D", "DownBlock2D")))
    with pytest.raises(ValueError, match="unknown mid_block_type"):
        UNet2DConditionModel(**dict(TINY, mid_block_type="Nope"))


def test_save_and_from_pretrained_roundtrip(tiny, tmp_path):
    from storygen_amd.model import UNet2DConditionModel
    tiny.save_pretrained(str(tmp_path / "ckpt" / "unet"))
    assert sorted(os.listdir(tmp_path / "ckpt" / "unet")) == ["config.json", "diffusion_pytorch_model.bin"]
    back = UNet2DConditionModel.from_pretrained(str(tmp_path / "ckpt"), subfolder="unet")
    assert dict(back.config) == dict(tiny.config)
    a, b = tiny.state_dict(), back.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    tiny.save_pretrained(str(tmp_path / "st"), safe_serialization=True)
    back2 = UNet2DConditionModel.from_pretrained(str(tmp_path / "st"), torch_dtype=torch.float16)
    assert back2.dtype == torch.float16
    assert torch.equal(back2.state_dict()["conv_in.weight"], a["conv_in.weight"].half())
    # from_config with the shipped SD-1.5 json keys (extra keys such as _class_name are ignored)
    m = UNet2DConditionModel.from_config(dict(TINY, _class_name="UNet2DConditionModel", _diffusers_version="0.6.0"))
    assert m.config.cross_attention_dim == 48


def test_load_sdm_state_dict_adopts_self_attention(tiny):
    """unet_2d_condition.py:487-510: attn3 <- attn1, norm4 <- norm1; unknown keys raise."""
    from storygen_amd.model import UNet2DConditionModel
    torch.manual_seed(1)
    donor = UNet2DConditionModel(**TINY)
    sdm = {k: v.clone() for k, v in donor.state_dict().items() if ".attn3." not in k and ".norm4." not in k}
    m = UNet2DConditionModel(**TINY)
    m.load_SDM_state_dict(dict(sdm))
    sd = m.state_dict()
    p = "down_blocks.0.attentions.0.transformer_blocks.0"
    assert torch.equal(sd[f"{p}.attn3.to_q.weight"], sdm[f"{p}.attn1.to_q.weight"])
    assert torch.equal(sd[f"{p}.attn3.to_out.0.bias"], sdm[f"{p}.attn1.to_out.0.bias"])
    assert torch.equal(sd[f"{p}.norm4.weight"], sdm[f"{p}.norm1.weight"])
    assert torch.equal(sd["conv_in.weight"], sdm["conv_in.weight"])
    with pytest.raises(KeyError, match="does not exist in model"):
        m.load_SDM_state_dict(dict(sdm, bogus=torch.zeros(1)))
    # a shape mismatch is dropped with a message (:494-497); the fill loop then looks the key up again and — exactly like
    # the reference, whose replace("attn3","attn1") leaves such a key unchanged (:501-505) — raises KeyError
    bad = dict(sdm)
    bad["conv_out.bias"] = torch.zeros(7)
    with pytest.raises(KeyError):
        m.load_SDM_state_dict(bad)


def test_memory_knobs_are_accepted(tiny):
    tiny.set_attention_slice("auto")
    tiny.set_attention_slice(4)
    with pytest.raises(ValueError):
        tiny.set_attention_slice(64)
    tiny.enable_xformers_memory_efficient_attention()
    tiny.set_use_memory_efficient_attention_xformers(True)
    tiny.enable_gradient_checkpointing()


def test_no_cpu_path_and_no_silent_autograd(tiny):
    x, e = torch.zeros(1, 4, 16, 16), torch.zeros(1, 77, 48)
    with pytest.raises(RuntimeError, match="no CPU path"):
        with torch.no_grad():
            tiny(x, 10, e)
    with pytest.raises(ValueError):
        tiny(x, 10, e, class_labels=torch.zeros(1))


def test_pipeline_mirror_validates_like_the_reference(tiny):
    from storygen_amd.model import StableDiffusionPipeline
    from storygen_amd.scheduler import DDIMSchedule
    pipe = StableDiffusionPipeline(vae=None, text_encoder=None, tokenizer=None, unet=tiny, scheduler=DDIMSchedule())
    assert pipe.vae_scale_factor == 8
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe.check_inputs("a", 100, 512, 1)
    with pytest.raises(ValueError, match="callback_steps"):
        pipe.check_inputs("a", 512, 512, 0)
    with pytest.raises(ValueError, match="prompt"):
        pipe.check_inputs(3, 512, 512, 1)
    lat = pipe.prepare_latents(2, 4, 64, 64, torch.float32, torch.device("cpu"), torch.Generator().manual_seed(0))
    assert lat.shape == (2, 4, 8, 8)
    with pytest.raises(ValueError, match="Unexpected latents shape"):
        pipe.prepare_latents(2, 4, 64, 64, torch.float32, torch.device("cpu"), None, torch.zeros(1, 4, 8, 8))
