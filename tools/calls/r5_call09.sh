#!/bin/bash
# round 5, call 9: key-split D = 160 attention (four waves on one 32-query block, keys split across the waves) — parity, isolated timings
# against the query-split instantiation, same-box A/B of the step
O=$GRAFT_REPO_ROOT/gpurun_out/r5i; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "attention" > $O/kernel_tests.log 2>&1; echo "attention tests rc=$?" > $O/summary.txt; tail -3 $O/kernel_tests.log >> $O/summary.txt
timeout 300 python - > $O/microbench.txt 2>&1 <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from storygen_amd import ops
dev = torch.device("cuda:0")
def rnd(*s): return (torch.randn(*s, device=dev)).half()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (B, Bk, Nq, Nk) in [(3, 2, 256, 768), (3, 3, 256, 256), (3, 3, 256, 77), (20, 20, 256, 256), (20, 20, 256, 77), (3, 3, 64, 192), (3, 3, 64, 64)]:
    C = 1280
    q, k, vt = rnd(B, Nq, C), rnd(Bk, Nk, C), rnd(Bk, C, (Nk + 7) & ~7)
    o = torch.empty(B, Nq, C, dtype=torch.float16, device=dev)
    res = {}
    for v in (3, 4):
        ops.debug_set_option("attn_d160", v)
        res[v] = t(lambda: ops.attention(q, k, vt, o, 8, 160 ** -0.5, nk=Nk))
    ops.debug_set_option("attn_d160", 4)
    print(f"D160 B{B} (kv {Bk}) Nq{Nq} Nk{Nk}: query-split {res[3]:6.1f} us   key-split {res[4]:6.1f} us")
PY
cat $O/microbench.txt >> $O/summary.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s -k "full_depth or unet_passes" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt; grep "latent rel-L2 at steps" $O/unet_tests.log >> $O/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_ksplit_$i.json 2>> $O/bench.err
  SG_DEV_OPTIONS=1 SG_ATTN_D160=3 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_qsplit_$i.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", {k:(v["launches"],round(v["ms"],3),round(v["tflops"],1)) for k,v in r["families"].items() if k.startswith("attention")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt
