#!/bin/bash
# what do the admissions cost, and which part of them: product (0), branches taken but no record stores (8), tau = +inf (16: fast path only)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
bash tools/ab.sh "0 8 16" 2 10 --no-secondary > $OUT/adm_ab.txt 2>&1
cat $OUT/adm_ab.txt
