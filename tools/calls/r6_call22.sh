#!/bin/bash
# round 6, call 22: training-step lines (bench.py fix), config-5 fp8 line with its deviation from the fp16 path, fp16 line beside it
O=$GRAFT_REPO_ROOT/gpurun_out/r6x; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --train-step --steps 5 --warmup 2 > $O/train_none.json 2>$O/err.txt; cut -c1-500 $O/train_none.json
timeout 900 python bench.py --train-step --steps 5 --warmup 2 --optimizer adamw8bit > $O/train_adamw8bit.json 2>>$O/err.txt; cut -c1-300 $O/train_adamw8bit.json
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --config5-shape --fp8-attention > $O/config5_fp8.json 2>>$O/err.txt; python -c "
import json; d=json.load(open('$O/config5_fp8.json')); print(d['ms_per_step'], d['fp8_attention'])"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop --config5-shape > $O/config5_fp16.json 2>>$O/err.txt; cut -c1-200 $O/config5_fp16.json
tail -3 $O/err.txt
