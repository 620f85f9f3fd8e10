"""ORACLE tooling — golden vectors for the stage-2 TRAINING step at BASELINE config 4's OWN SIZE, from the REFERENCE ITSELF.

oracle/make_golden_train.py pins the step on a tiny UNet; this recipe runs the reference's own UNet2DConditionModel
(/root/reference/model/unet_2d_condition.py on oracle/diffusers_shim) at the SD-1.5 config, 64x64 latent, batch 4, reference
frames (0, 1, 2), fixed seeded timesteps, in the state train_StorySalon_stage2.py:166-177 puts it in, and the loss / backward of
:291-327 with torch autograd on CPU fp32.

Memory: one batch-4 backward through the five 64x64-level transformer blocks keeps ~45 GB of attention probabilities alive (more
than this container can promise), so the batch is walked ONE SAMPLE AT A TIME and the gradients are accumulated in `.grad`:
the samples of a batch are independent in this network (GroupNorm / LayerNorm / attention are per sample) and the loss is a mean
over all B*4*H*W elements, so  d loss / d W = sum_b d (sum-of-squares_b / (B*4*H*W)) / d W  exactly — what autograd itself does
when a batch is split for gradient accumulation.  `mode` in the file records this.

Stored: the loss, the per-sample loss terms, and for each of the 80 attn3 gradients its L2 norm + a fixed random index sample;
full tensors for the representative set FULL (both ends of the UNet, every projection kind).

Usage:  python oracle/make_golden_train_sd15.py        (build container only; writes tests/golden/sd15_train_bs4.pt, ~6 min on 6 cores)
"""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from storygen_amd.arch import SD15_CONFIG, build_arch, load_config  # noqa: E402
from storygen_amd.synth import seed_int, synthetic_state_dict, synthetic_train_batch  # noqa: E402
from oracle import storygen_oracle as O  # noqa: E402
from oracle.ref_runner import build_reference_unet  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
N_PROBE = 512
SEED, BATCH, HW = 11, 4, 64
USE_REFS = (0, 1, 2)
FULL = ("down_blocks.0.attentions.0.transformer_blocks.0.attn3.to_q.weight",
        "down_blocks.0.attentions.0.transformer_blocks.0.attn3.to_k.weight",
        "down_blocks.0.attentions.0.transformer_blocks.0.attn3.to_v.weight",
        "down_blocks.0.attentions.0.transformer_blocks.0.attn3.to_out.0.weight",
        "down_blocks.1.attentions.1.transformer_blocks.0.attn3.to_q.weight",
        "up_blocks.3.attentions.2.transformer_blocks.0.attn3.to_q.weight",
        "up_blocks.3.attentions.2.transformer_blocks.0.attn3.to_k.weight",
        "up_blocks.3.attentions.2.transformer_blocks.0.attn3.to_out.0.weight",
        "up_blocks.3.attentions.2.transformer_blocks.0.attn3.to_out.0.bias")


def main():
    cfg = load_config(SD15_CONFIG)
    arch = build_arch(cfg)
    sd = synthetic_state_dict(arch, SEED)
    batch = synthetic_train_batch(BATCH, HW, cfg["cross_attention_dim"], SEED)
    unet = build_reference_unet(cfg, sd)
    unet.requires_grad_(False)                                                            # train_StorySalon_stage2.py:166-168
    for name, module in unet.named_modules():                                             # :170-175
        if name.endswith("attn3"):
            for p in module.parameters():
                p.requires_grad = True
    unet.train()                                                                          # :263
    sched = O.DDIM()
    n_el = BATCH * 4 * HW * HW
    terms, t_all = [], time.time()
    for b in range(BATCH):
        t0 = time.time()
        sl = slice(b, b + 1)
        t = batch["timesteps"][sl].long()
        ref_t = (batch["timesteps"][sl] / 10).long()                                      # :297-300
        noisy = O.ddpm_add_noise(sched, batch["latents"][sl], batch["noise"][sl], t)      # :303
        feats = []
        with torch.no_grad():                                                             # no trainable parameter is evaluated in a reference pass
            for i in USE_REFS:
                ti = ref_t * (3 - i)                                                      # :311
                x = O.ddpm_add_noise(sched, batch["ref_latents"][i][sl], batch["ref_noise"][sl], ti)
                feats.append(unet(x, ti, encoder_hidden_states=batch["prev_text"][i][sl], return_dict=False)[1])
        ctx = {k: torch.cat([f[k] for f in feats], dim=1) for k in feats[0]}              # :313-316
        pred = unet(noisy, t, encoder_hidden_states=batch["text"][sl], image_hidden_states=ctx, return_dict=False)[0]
        m = 1.0 - batch["mask"][sl]
        term = ((pred.float() * m - batch["noise"][sl].float() * m) ** 2).sum() / n_el    # this sample's share of F.mse_loss(..., "mean") (:324)
        term.backward()                                                                   # accumulates into .grad
        terms.append(float(term))
        print(f"sample {b}: loss share {float(term):.6f}, {time.time() - t0:.1f}s", flush=True)
        del pred, ctx, feats, term
    loss = sum(terms)
    grads = {n: p.grad.detach().clone() for n, p in unet.named_parameters() if p.requires_grad}
    assert len(grads) == 5 * len(arch.feature_keys) and all(k.endswith(O.TRAINABLE_SUFFIXES) for k in grads)
    print(f"reference train step at SD-1.5 size: loss {loss:.6f}, {len(grads)} grads, {time.time() - t_all:.1f}s", flush=True)
    out = dict(case="sd15_train_bs4", config=cfg, seed=SEED, batch=BATCH, hw=HW, use_refs=USE_REFS, trainable="attn3",
               mode="per-sample accumulation (exact: independent samples, mean loss)", loss=loss, loss_terms=terms,
               made_by="oracle/make_golden_train_sd15.py", torch=torch.__version__, threads=torch.get_num_threads(), grads={})
    for k, g in grads.items():
        gi = torch.Generator().manual_seed(seed_int("probe." + k, 0))
        idx = torch.randint(0, g.numel(), (min(N_PROBE, g.numel()),), generator=gi)
        e = dict(shape=tuple(g.shape), l2=float(g.double().norm()), absmax=float(g.abs().max()), idx=idx, values=g.flatten()[idx].clone())
        if k in FULL:
            e["full"] = g.clone()
        out["grads"][k] = e
    assert all(k in grads for k in FULL)
    path = os.path.join(GOLDEN, "sd15_train_bs4.pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("SG_GOLDEN_THREADS", os.cpu_count() or 1)))
    main()
