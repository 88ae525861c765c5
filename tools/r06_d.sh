#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests_d.txt 2>&1; tail -4 $OUT/gpu_tests_d.txt
bash tools/r06_mining_prof.sh > $OUT/mining_prof5.log 2>&1
for f in $OUT/mining_kernels_*top*_ids.txt $OUT/mining_kernels_*top*_exact.txt; do mv $f ${f%.txt}_v5.txt; done
grep -h -A6 "per SEARCH" $OUT/mining_kernels_t2i_top1000_ids_v5.txt $OUT/mining_kernels_i2t_top1000_ids_v5.txt | cut -c1-160
