#!/bin/bash
# A/B: rare-path address arithmetic on 24-bit multiplies (lightningdot_amd/libldot.so) vs the committed head (ab_base/)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
bash tools/ab_lib.sh "ab_base/lightningdot_amd/libldot.so lightningdot_amd/libldot.so" 3 20 > $OUT/ab_u24.txt 2>&1
cat $OUT/ab_u24.txt
