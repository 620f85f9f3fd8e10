#!/bin/bash
# round 5, call 14: the LayerNorm-fold guard over the inputs of eight ranks (bench.py seeds its inputs by rank) on full 50-step runs, and the
# non-contract lines on the final sources
O=$GRAFT_REPO_ROOT/gpurun_out/r5n; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python - > $O/guard_seeds.txt 2>&1 <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from storygen_amd.arch import SD15_CONFIG, build_arch
from storygen_amd.engine import EngineWeights
from storygen_amd.sampler import StoryGenSampler
from storygen_amd.synth import synthetic_inputs, synthetic_state_dict
arch = build_arch(SD15_CONFIG)
wts = EngineWeights(arch, synthetic_state_dict(arch, 0), "cuda:0")
smp = StoryGenSampler(arch, None, "cuda:0", 1, 64, 64, 3, weights=wts, ref_ahead=5)
for stage in ("multi-image-condition", "auto-regressive"):
    for seed in range(8):
        inputs = synthetic_inputs(1, 3, 64, 64, seed=seed, cross_attention_dim=768)
        smp.prepare(inputs, 50, stage, 7.5, 3.5)
        lat = smp.run()
        torch.cuda.synchronize()
        flags = [int(e.ln_guard.item()) for e in (smp.main, smp.ref)]
        print(stage, "rank seed", seed, "finite", bool(torch.isfinite(lat).all()), "guard flags (main, ref)", flags, "|latents| max", float(lat.abs().max()))
        smp.check_guards()
print("all clear")
PY
tail -18 $O/guard_seeds.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --stage auto-regressive 2>>$O/bench.err | tail -n 1 > $O/bench_autoregressive.json; cut -c1-200 $O/bench_autoregressive.json
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --config5-shape 2>>$O/bench.err | tail -n 1 > $O/config5_fp16_bench.json; cut -c1-200 $O/config5_fp16_bench.json
timeout 300 python bench.py --train-step --steps 8 --warmup 2 2>>$O/bench.err | tail -n 1 > $O/train_none.json; cut -c1-260 $O/train_none.json
