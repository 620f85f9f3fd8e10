#!/bin/bash
# gpurun --timeout 400 -- 'bash tools/next_round/02_schedule_ab.sh'
O=gpurun_out/next2; mkdir -p $O
B="python bench.py --steps 16 --warmup 4 --no-cpu-baseline"
run() { name=$1; shift; timeout 150 $B "$@" > $O/$name.log 2>&1; echo -n "$name: "; grep -o '"ms_per_step": [0-9.]*' $O/$name.log || tail -n 3 $O/$name.log; }
run single_graph
run split --split-graphs
run split_prio --split-graphs --stream-priority
run ahead2_prio --ref-ahead 2 --stream-priority
run ahead4_prio --ref-ahead 4 --stream-priority
run ahead4 --ref-ahead 4
SG_SIDE=both run split_prio_sideboth --split-graphs --stream-priority   # separate graphs: the reference pass can fork too
run single_graph_again
