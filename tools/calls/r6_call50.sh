#!/bin/bash
# round 6, call 50: with the packed-add hazard gone, is there anything to gain from letting paired launches / the columns-are-tokens fold take the latency kernel (lat_mask 63)?
O=$GRAFT_REPO_ROOT/gpurun_out/r6br; mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  SG_DEV_OPTIONS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/m62_$i.json 2>$O/err.txt; echo "lat_mask 62 (shipped) $(python -c "import json;print(json.load(open('$O/m62_$i.json'))['ms_per_step'])")"
  SG_DEV_OPTIONS=1 SG_LAT_MASK=63 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-loop > $O/m63_$i.json 2>$O/err.txt; echo "lat_mask 63           $(python -c "import json;print(json.load(open('$O/m63_$i.json'))['ms_per_step'])")"
done
