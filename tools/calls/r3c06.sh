#!/bin/bash
# round 3, call 6: how the HIP graph runtime's knobs change the overlap of the two branches of the step graph
O=$GRAFT_REPO_ROOT/gpurun_out/r3c06; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; }
run A=1 2>&1 | tee -a $O/env_matrix.txt
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 2>&1 | tee -a $O/env_matrix.txt
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2 2>&1 | tee -a $O/env_matrix.txt
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8 2>&1 | tee -a $O/env_matrix.txt
run GPU_MAX_HW_QUEUES=8 2>&1 | tee -a $O/env_matrix.txt
run GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 2>&1 | tee -a $O/env_matrix.txt
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 2>&1 | tee -a $O/env_matrix.txt
run DEBUG_HIP_DYNAMIC_QUEUES=0 2>&1 | tee -a $O/env_matrix.txt
run DEBUG_HIP_GRAPH_BATCH_SIZE=1 2>&1 | tee -a $O/env_matrix.txt
run DEBUG_HIP_GRAPH_BATCH_SIZE=64 2>&1 | tee -a $O/env_matrix.txt
echo "== probe with packet capture off"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 python tools/probe_schedule.py 2>&1 | tail -n 4 | tee -a $O/env_matrix.txt
