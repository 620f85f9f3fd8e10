#!/bin/bash
# round 6, call 4: full GPU suite on the round-6 sources so far (latency kernel, ff2 + proj_out as one GEMM, tail graph, un-gated N = 2 test,
# config-5 tests at G = 5, per-feature bar), contract line with loop_50_steps_ms, merge A/B, cold-weight chain probe
O=$GRAFT_REPO_ROOT/gpurun_out/r6d; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log; grep -E "eps\(ref\)|merged|N=2|config 5" $O/gpu_tests.log | cut -c1-400
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_$i.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ff-proj-merge --no-loop > $O/bench_nomerge_$i.json 2>$O/bench.err; cut -c1-200 $O/bench_nomerge_$i.json
done
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6d/bench_1.json"))
print({k:d[k] for k in ("ms_per_step","loop_50_steps_ms","loop","ms_per_step_by_rank")})
PY
timeout 600 python tools/bench_chain.py --cold > $O/chain_cold.txt 2>&1; cat $O/chain_cold.txt
