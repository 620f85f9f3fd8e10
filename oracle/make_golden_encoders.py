#!/usr/bin/env python
"""TEST INFRASTRUCTURE — pins oracle/encoders_oracle.py::clip_text_forward to the installed transformers' CLIPTextModel and writes
tests/golden/clip_text_tiny.pt (random weights of a small CLIP text config, ids, last_hidden_state, pooled).

    python oracle/make_golden_encoders.py          # needs transformers (present in the build container, not required on the GPU box)

The reference builds the text encoder with `CLIPTextModel.from_pretrained(..., subfolder="text_encoder")` (inference.py:45,
train_StorySalon_stage2.py:141) and calls it at model/pipeline.py:137,183.  The AutoencoderKL has no importable implementation
here (diffusers absent), so the VAE restatement stays unpinned — see the oracle's header."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import encoders_oracle as eo  # noqa: E402


def main():
    from transformers import CLIPTextConfig, CLIPTextModel
    out = {}
    for name, kw, B in (("tiny", dict(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4), 2),
                        ("sd15_2layers", dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=12), 1)):
        cfg = CLIPTextConfig(max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=0, eos_token_id=2, **kw)
        torch.manual_seed(0)
        model = CLIPTextModel(cfg).eval()
        # default init leaves biases at zero and norms at identity: perturb so that every term is exercised
        with torch.no_grad():
            for k, p in model.named_parameters():
                if k.endswith("bias"):
                    p.normal_(0.0, 0.05)
                elif "layer_norm" in k and k.endswith("weight"):
                    p.add_(0.1 * torch.randn_like(p))
        ids = torch.randint(3, kw["vocab_size"] - 1, (B, 77))
        ids[:, 0] = 0
        ids[:, 40] = kw["vocab_size"] - 1           # the largest id marks the pooled position (EOS in the real vocabulary)
        with torch.no_grad():
            ref = model(ids)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        mine, pooled = eo.clip_text_forward(sd, ids, heads=kw["num_attention_heads"])
        e1 = ((mine - ref[0]).norm() / ref[0].norm()).item()
        e2 = ((pooled - ref[1]).norm() / ref[1].norm()).item()
        print(f"{name}: restatement vs transformers {__import__('transformers').__version__}: hidden {e1:.2e} pooled {e2:.2e}")
        assert e1 < 1e-5 and e2 < 1e-5
        if name == "tiny":
            out = dict(state_dict={k: v.half() for k, v in sd.items()}, input_ids=ids, heads=4, last_hidden_state=None, pooled=None,
                       made_by="oracle/make_golden_encoders.py", transformers=__import__("transformers").__version__)
            # the fixture stores fp16 weights (what the engine uses); the expected outputs are transformers' on those same weights
            model.load_state_dict({k: v.float() for k, v in out["state_dict"].items()})
            with torch.no_grad():
                ref = model(ids)
            out["last_hidden_state"], out["pooled"] = ref[0].clone(), ref[1].clone()
    path = os.path.join(ROOT, "tests", "golden", "clip_text_tiny.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
