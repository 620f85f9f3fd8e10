#!/bin/bash
# round 6, call 12: kernel-level stress with varying predecessors (does any form read LDS / registers it never wrote?), with pairs forced onto the latency kernel by hint;
# the whole-loop line with the primer as a graph
O=$GRAFT_REPO_ROOT/gpurun_out/r6l; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python tools/exp_kernel_determinism.py 200 pollute > $O/kernel_det_polluted.txt 2>&1; grep -v " 0 of" $O/kernel_det_polluted.txt | tail -30
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>$O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6l/bench.json"))
print({k:d[k] for k in ("ms_per_step","loop_50_steps_ms","loop")})
PY
