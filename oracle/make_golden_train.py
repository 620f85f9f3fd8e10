"""ORACLE tooling — golden vectors for the stage-2 TRAINING step (BASELINE config 4) from the REFERENCE ITSELF.

The reference's own UNet2DConditionModel (/root/reference/model/unet_2d_condition.py on oracle/diffusers_shim) is put in the
state train_StorySalon_stage2.py:166-177 puts it in (everything frozen, parameters of modules named `*attn3` trainable,
`unet.train()`), and the loss / backward of :291-327 is run on CPU fp32 with torch autograd on seeded synthetic inputs
(the CLIP / VAE plumbing is outside the path).  Stored: the loss, and for every trainable tensor its L2 norm plus a
fixed random index sample.  As a self-check the restatement oracle.storygen_oracle.train_step must reproduce them.

Usage:  python oracle/make_golden_train.py            (build container only; writes tests/golden/tiny_train.pt)
        python oracle/make_golden_train.py stage1     (train_StorySalon_stage1.py:171-179,262-291: modules named `*attn1` trainable, no
                                                       reference pass, image_hidden_states=None; writes tests/golden/tiny_train_stage1.pt)
        python oracle/make_golden_train.py coco       (train_COCO.py:286-316: three frames, all at noise level ref_t, unmasked loss;
                                                       writes tests/golden/tiny_train_coco.pt)
"""
from __future__ import annotations

import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from storygen_amd.arch import SD15_CONFIG, build_arch, load_config  # noqa: E402
from storygen_amd.synth import seed_int, synthetic_state_dict, synthetic_train_batch  # noqa: E402
from oracle import storygen_oracle as O  # noqa: E402
from oracle.ref_runner import build_reference_unet  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
TINY_CONFIG = dict(SD15_CONFIG, block_out_channels=(32, 64, 128, 128), cross_attention_dim=48, sample_size=128)
N_PROBE = 512


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main(stage1: bool = False, coco: bool = False):
    cfg = load_config(TINY_CONFIG)
    arch = build_arch(cfg)
    seed, b, hw = 5, 2, 64            # the reference's consume path needs a 64..94 latent
    sd = synthetic_state_dict(arch, seed)
    batch = synthetic_train_batch(b, hw, cfg["cross_attention_dim"], seed)
    target = "attn1" if stage1 else "attn3"
    if coco:
        batch["mask"] = torch.zeros_like(batch["mask"])                                   # train_COCO.py:315: plain MSE
    out = dict(case="tiny_train_stage1" if stage1 else ("tiny_train_coco" if coco else "tiny_train"), config=cfg, seed=seed, batch=b, hw=hw, trainable=target,
               made_by="oracle/make_golden_train.py", cases={})
    for use_refs in (((),) if stage1 else (((0, 1, 2),) if coco else ((0, 1, 2), (2,)))):
        unet = build_reference_unet(cfg, sd)
        unet.requires_grad_(False)                                                        # :166-168
        for name, module in unet.named_modules():                                         # :170-175
            if name.endswith(target):
                for p in module.parameters():
                    p.requires_grad = True
        unet.train()                                                                      # :263
        sched = O.DDIM()
        t = batch["timesteps"].long()
        ref_t = (batch["timesteps"] / 10).long()
        t0 = time.time()
        noisy = O.ddpm_add_noise(sched, batch["latents"], batch["noise"], t)
        feats = []
        for i in use_refs:
            ti = ref_t if coco else ref_t * (3 - i)                                       # train_COCO.py:303-304 vs stage 2 :311
            x = O.ddpm_add_noise(sched, batch["ref_latents"][i], batch["ref_noise"], ti)
            feats.append(unet(x, ti, encoder_hidden_states=batch["prev_text"][i], return_dict=False)[1])
        ctx = {k: torch.cat([f[k] for f in feats], dim=1) for k in feats[0]} if feats else None
        pred = unet(noisy, t, encoder_hidden_states=batch["text"], image_hidden_states=ctx, return_dict=False)[0]
        loss = F.mse_loss(pred.float() * (1.0 - batch["mask"]), batch["noise"].float() * (1 - batch["mask"]), reduction="mean")
        loss.backward()
        grads = {n: p.grad.detach().clone() for n, p in unet.named_parameters() if p.requires_grad}
        assert all(k.endswith(O.trainable_suffixes(target)) for k in grads) and len(grads) == 5 * len(arch.feature_keys)
        print(f"reference train step refs={use_refs}: loss {float(loss):.6f}, {len(grads)} grads, {time.time() - t0:.1f}s", flush=True)
        o_loss, o_grads = O.train_step(sd, cfg, batch, use_refs, trainable=target, ref_levels="coco" if coco else "stage2")
        errs = [abs(float(o_loss) - float(loss)) / abs(float(loss))] + [rel_l2(o_grads[k], grads[k]) for k in grads]
        print(f"restatement vs reference: max rel err {max(errs):.2e}", flush=True)
        assert max(errs) < 1e-4, errs
        entry = dict(loss=float(loss), restatement_rel_err=max(errs), grads={})
        for k, g in grads.items():
            gi = torch.Generator().manual_seed(seed_int("probe." + k, 0))
            idx = torch.randint(0, g.numel(), (min(N_PROBE, g.numel()),), generator=gi)
            entry["grads"][k] = dict(shape=tuple(g.shape), l2=float(g.double().norm()), idx=idx, values=g.flatten()[idx].clone())
        out["cases"]["refs_" + "".join(map(str, use_refs))] = entry
    path = os.path.join(GOLDEN, out["case"] + ".pt")
    torch.save(out, path)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 1)
    main(stage1=len(sys.argv) > 1 and sys.argv[1] == "stage1", coco=len(sys.argv) > 1 and sys.argv[1] == "coco")
