#!/bin/bash
# round 6, call 21: closing measurements B: contract line again (traffic on file is now of this build), non-contract lines
O=$GRAFT_REPO_ROOT/gpurun_out/r6y; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/err.txt; cut -c1-200 $O/bench.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --stage auto-regressive > $O/bench_autoregressive.json 2>>$O/err.txt; cut -c1-260 $O/bench_autoregressive.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --config5-shape > $O/config5_fp16.json 2>>$O/err.txt; cut -c1-300 $O/config5_fp16.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --config5-shape --fp8-attention > $O/config5_fp8.json 2>>$O/err.txt; cut -c1-300 $O/config5_fp8.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --config5-shape > $O/config5_fp16_b.json 2>>$O/err.txt; cut -c1-300 $O/config5_fp16_b.json
timeout 900 python bench.py --train-step --steps 5 --warmup 2 > $O/train_none.json 2>>$O/err.txt; cut -c1-400 $O/train_none.json
timeout 900 python bench.py --train-step --steps 5 --warmup 2 --optimizer adamw8bit > $O/train_adamw8bit.json 2>>$O/err.txt; cut -c1-300 $O/train_adamw8bit.json
tail -5 $O/err.txt
