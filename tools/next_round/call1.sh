#!/bin/bash
O=gpurun_out/r2c1; mkdir -p $O
SG_TEST_UNVALIDATED=1 timeout 400 python -m pytest tests/test_backward_gpu.py -q -m gpu -x --no-header -p no:cacheprovider 2>&1 | tee $O/backward_x.log | tail -n 40
SG_TEST_UNVALIDATED=1 timeout 400 python -m pytest tests/test_backward_gpu.py -q -m gpu --no-header -p no:cacheprovider 2>&1 > $O/backward_all.log; tail -n 60 $O/backward_all.log
SG_TEST_UNVALIDATED=1 timeout 120 python -m pytest tests/test_unet_gpu.py -q -m gpu -k "split_graphs" 2>&1 | tee $O/split.log | tail -n 5
timeout 60 python tools/probe_mfma_f8.py > $O/f8probe.log 2>&1; tail -n 30 $O/f8probe.log
B="python bench.py --steps 16 --warmup 4 --no-cpu-baseline"
run() { name=$1; shift; timeout 150 $B "$@" > $O/$name.log 2>&1; echo -n "$name: "; grep -o '"ms_per_step": [0-9.]*' $O/$name.log || tail -n 3 $O/$name.log; }
run single_graph
run split --split-graphs
run split_prio --split-graphs --stream-priority
run single_graph_again
