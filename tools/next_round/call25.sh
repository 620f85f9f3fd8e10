#!/bin/bash
# ones-row denominator patch for the fp16 attention kernel: parity + per-launch time (tree restored afterwards; not adopted this round)
O=$GRAFT_REPO_ROOT/gpurun_out/r2c25; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 60 python -m pytest tests/test_kernels_gpu.py -q -m gpu --no-header -p no:cacheprovider -k "attention and not f8 and not small" 2>&1 | tail -n 3 | tee $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o p -- python $GRAFT_REPO_ROOT/tools/pmc_kernels.py attn > $O/st.log 2>&1
grep attn_fwd $O/st/p_kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete
