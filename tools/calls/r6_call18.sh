#!/bin/bash
# round 6, call 18: every kind on the latency kernel (lat_mask 63): ONE graph per step without the two-branch overlap vs with it
O=$GRAFT_REPO_ROOT/gpurun_out/r6r; mkdir -p $O
cd $GRAFT_REPO_ROOT
SG_LAT_MASK=63 timeout 900 python tools/exp_determinism.py reps=40 only=graph-no-overlap > $O/det_graph_no_overlap.txt 2>&1; echo "graph, no overlap: $(grep -c bit-identical $O/det_graph_no_overlap.txt) of 40"
SG_LAT_MASK=63 timeout 900 python tools/exp_determinism.py reps=40 only=split > $O/det_split.txt 2>&1; echo "split graphs: $(grep -c bit-identical $O/det_split.txt) of 40"
SG_LAT_MASK=63 timeout 900 python tools/exp_determinism.py reps=40 only=eager > $O/det_eager.txt 2>&1; echo "eager: $(grep -c bit-identical $O/det_eager.txt) of 40"
