#!/bin/bash
# the ladder from the micro-benchmark's slab loop to the product kernel, on ONE box: loop line, no-filter kernel with L2-resident / cache-resident /
# real operands, + filter, + admissions; then where the waves wait (PMC)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05c}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 300 tools/bin/mfma_ceiling 12 t16b 2>&1 | tee $O/ladder.txt
bash tools/ab.sh "529 81 17 16 0" 2 10 "--no-secondary" | tee -a $O/ladder.txt
bash tools/pmc_waits.sh 2>&1 | tee $O/pmc_waits.txt
