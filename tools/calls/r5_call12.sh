#!/bin/bash
# round 5, call 12: hidden-split fused feed-forward (two workgroups per 128 tokens for launches <= 16 k tokens; proj_out contracts [y_a | y_b]
# with [W | W]) — kernel parity, full-depth parity, isolated timing, same-box A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r5l; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "feed_forward or geglu" > $O/kernel_tests.log 2>&1; echo "ff tests rc=$?" > $O/summary.txt; tail -3 $O/kernel_tests.log >> $O/summary.txt
timeout 300 python - > $O/microbench.txt 2>&1 <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from storygen_amd import ops
from storygen_amd.repack import ff_fused_pack, fold_layernorm, interleave_geglu
dev = torch.device("cuda:0")
def rnd(*s, sc=1.0, dt=torch.float16): return (torch.randn(*s, device=dev) * sc).to(dt)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
C = 320
w1, b1 = interleave_geglu(rnd(8 * C, C, sc=C ** -0.5), rnd(8 * C))
w1f, _, d1 = fold_layernorm(w1, b1, rnd(C) + 1.0, rnd(C))
pack = ff_fused_pack(w1f.contiguous(), d1.contiguous(), rnd(C, 4 * C, sc=(4 * C) ** -0.5))
wo = rnd(C, C, sc=C ** -0.5); wo2 = torch.cat([wo, wo], 1).contiguous(); bo = rnd(C)
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for M in (4096, 12288, 16384, 81920):
    x = rnd(M, C, dt=torch.float32); y = torch.empty(M, C, dtype=torch.float16, device=dev); y2 = torch.empty(M, 2 * C, dtype=torch.float16, device=dev)
    o = torch.empty(M, C, dtype=torch.float32, device=dev)
    a = t(lambda: ops.ff_fused(x, pack, rnd(C), y)); b = t(lambda: ops.ff_fused(x, pack, rnd(C), y2, split=True))
    c = t(lambda: ops.gemm(y, wo, o, bias=bo, res1=x, workspace=ws)); d = t(lambda: ops.gemm(y2, wo2, o, bias=bo, res1=x, workspace=ws))
    print(f"M{M}: ff_fused {a:6.1f} us, hidden-split {b:6.1f} us | proj_out K=C {c:5.1f} us, K=2C {d:5.1f} us | chain {a + c:6.1f} -> {b + d:6.1f} us")
PY
cat $O/microbench.txt >> $O/summary.txt
timeout 600 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -s -k "full_depth or unet_passes" > $O/unet_tests.log 2>&1; echo "unet tests rc=$?" >> $O/summary.txt; grep "latent rel-L2 at steps" $O/unet_tests.log >> $O/summary.txt
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 > $O/bench_split_$i.json 2>> $O/bench.err
  timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 10 --no-ff-split > $O/bench_nosplit_$i.json 2>> $O/bench.err
done
for f in $O/bench_*.json; do python - "$f" <<'PY' >> $O/summary.txt
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], "ms", d["tflop_per_step_executed"], {k:(v["launches"],round(v["ms"],2),round(v["tflops"])) for k,v in r["families"].items() if k in ("gemm","ff_fused")})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat $O/summary.txt
