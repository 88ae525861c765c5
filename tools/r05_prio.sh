#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05l
bash tools/ab.sh "0 2 4 16 18 20" 3 10 "--no-secondary" | tee gpurun_out/r05l/ab_setprio.txt
