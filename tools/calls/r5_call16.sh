#!/bin/bash
# round 5, call 16: rocprofv3 kernel stats of the contract command on the FINAL sources + the contract line of the same box
O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; cut -c1-200 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/ks.log 2>&1
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; head -8 $O/kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/ks
