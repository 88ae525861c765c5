#!/bin/bash
# round 6: PMC passes on the score kernel at head (separate --pmc runs with --kernel-trace only) + the wall time of the default bench run
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
( for g in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE WRITE_SIZE"; do
    echo "## --pmc $g (5 passes: 2 steps + recall / parity passes of bench.py --steps 1 --warmup 1)"; bash tools/pmc2.sh "$g"
  done ) > $O/pmc_score_filter_t16.txt 2>&1
cat $O/pmc_score_filter_t16.txt
cd $GRAFT_REPO_ROOT; ( time python bench.py > $O/bench_default_timed.json 2> $O/bench_default_timed.err ) 2> $O/bench_default_time.txt; cat $O/bench_default_time.txt
