#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05m
bash tools/ab.sh "17 1041 81 529" 3 10 "--no-secondary" | tee gpurun_out/r05m/ab_tile_major_bound.txt
