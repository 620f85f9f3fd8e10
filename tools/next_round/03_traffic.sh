#!/bin/bash
# gpurun --timeout 400 -- 'bash tools/next_round/03_traffic.sh'   (two PMC passes; --pmc only with --kernel-trace, as gpurun requires)
O=$GRAFT_REPO_ROOT/gpurun_out/next3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
timeout 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/traffic_from_pmc.py $O/f/p_counter_collection.csv $O/w/p_counter_collection.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline; MI355X; $(date -u +%F)" | tee $O/traffic.json
rm -f $O/f/p_kernel_trace.csv $O/w/p_kernel_trace.csv       # keep the merge-back small
# profiles/traffic.json is rewritten on the GPU box only: copy gpurun_out/next3/traffic.json over profiles/traffic.json afterwards
