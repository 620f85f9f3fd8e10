#!/bin/bash
# round 6, call 49: non-contract lines and per-shape / chain tables on the FINAL library (no packed fp32 arithmetic)
O=$GRAFT_REPO_ROOT/gpurun_out/r6ey; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --stage auto-regressive > $O/bench_autoregressive.json 2>$O/err.txt; cut -c1-200 $O/bench_autoregressive.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --config5-shape > $O/config5_fp16.json 2>>$O/err.txt; cut -c1-220 $O/config5_fp16.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --config5-shape --fp8-attention > $O/config5_fp8.json 2>>$O/err.txt; cut -c1-220 $O/config5_fp8.json
timeout 900 python tools/profile_step.py --ref-ahead 5 > $O/per_shape.txt 2>&1; head -6 $O/per_shape.txt
timeout 600 python tools/bench_chain.py default > $O/chain.txt 2>&1; tail -6 $O/chain.txt
timeout 300 python tools/bench_attn_bwd.py > $O/attn_bwd.txt 2>&1; tail -11 $O/attn_bwd.txt
