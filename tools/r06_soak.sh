#!/bin/bash
# round-6 soak: the randomised parity sweeps at more cases / other seeds than the test suite runs (k in {500, 1000, 2048} exercise select_big.hip)
O=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $O
timeout 1500 python tools/fuzz_search.py 160 61 > $O/fuzz_search_soak.txt 2>&1; tail -2 $O/fuzz_search_soak.txt
timeout 1500 python tools/fuzz_search.py 160 62 >> $O/fuzz_search_soak.txt 2>&1; tail -2 $O/fuzz_search_soak.txt
timeout 900 python tools/fuzz_sharded.py 120 63 > $O/fuzz_sharded_soak.txt 2>&1; tail -2 $O/fuzz_sharded_soak.txt
