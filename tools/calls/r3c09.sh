#!/bin/bash
# round 3, call 9: is split-K (one extra reduce launch per GEMM) still paying at step level?
O=$GRAFT_REPO_ROOT/gpurun_out/r3c09; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
run A=1 2>&1 | tee -a $O/nosplit.txt
run SG_DEV_OPTIONS=1 SG_NO_SPLIT=1 2>&1 | tee -a $O/nosplit.txt
run A=1 2>&1 | tee -a $O/nosplit.txt
