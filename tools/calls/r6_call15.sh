#!/bin/bash
# round 6, call 15: grid-barrier probe (what a layer boundary costs inside a persistent launch vs a dependent launch in a graph)
O=$GRAFT_REPO_ROOT/gpurun_out/r6o; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 storygen_amd/lib/probe_grid_barrier > $O/grid_barrier.txt 2>&1; cat $O/grid_barrier.txt
