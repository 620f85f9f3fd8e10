#!/bin/bash
# round 3, call 3: epilogue v3 (wave-private LDS transpose, coalesced row quads, phase-batched loads): parity + anatomy + step
O=$GRAFT_REPO_ROOT/gpurun_out/r3c03; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -x -k "mfma or gemm or conv3x3 or geglu or register or tile_waves or groupnorm_statistics or split_k" 2>&1 | tee $O/kernel_tests.log | tail -n 8
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -n 1
timeout 300 python tools/anatomy.py > $O/anatomy.txt 2>&1; grep -E "gemm|conv" $O/anatomy.txt | cut -c1-60,180-260
timeout 300 python tools/profile_step.py > $O/per_shape.txt 2>&1; head -n 30 $O/per_shape.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cut -c1-200 $O/bench.json
