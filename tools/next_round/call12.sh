#!/bin/bash
O=gpurun_out/r2c12; mkdir -p $O
timeout 400 python -m pytest tests/test_backward_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "training_step" 2>&1 | tail -n 15
timeout 600 python bench.py --train-step --steps 5 --warmup 1 > $O/train_graph.json 2>$O/err1.log; tail -c 700 $O/train_graph.json; tail -n 3 $O/err1.log
timeout 600 python bench.py --train-step --steps 3 --warmup 1 --no-graph > $O/train_eager.json 2>$O/err2.log; tail -c 400 $O/train_eager.json
