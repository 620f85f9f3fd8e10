#!/bin/bash
# round 3, call 12: second-pass GroupNorm statistics, batched statistics prologue of gn_apply_wide, paired text+image attention,
# SiLU of the time embedding moved to its producer — targeted tests, then A/B bench lines and the per-shape table
O=$GRAFT_REPO_ROOT/gpurun_out/r3c12; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1 | tee $O/smoke.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --no-header -p no:cacheprovider --maxfail=10 -k "statistics or attention or groupnorm or register_epilogue or conv3x3_every" 2>&1 | tee $O/kernel_tests.log | tail -n 15
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_dropin_gpu.py -q --no-header -p no:cacheprovider --maxfail=10 -k "single_pass or denoise_steps or graph_replay or config5_fp16 or harvest_then_consume or loop_vs_oracle" 2>&1 | tee $O/unet_tests.log | tail -n 15
timeout 300 python bench.py --no-cpu-baseline > $O/bench_a.json 2>/dev/null; cut -c1-190 $O/bench_a.json
timeout 300 python bench.py --no-cpu-baseline --no-attn-pair > $O/bench_nopair.json 2>/dev/null; cut -c1-190 $O/bench_nopair.json
timeout 300 python bench.py --no-cpu-baseline > $O/bench_b.json 2>/dev/null; cut -c1-190 $O/bench_b.json
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; head -n 4 $O/per_shape.txt
