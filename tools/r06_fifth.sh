#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_tests.txt 2>&1
tail -4 gpurun_out/r06/gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r06/bench_default.json 2> gpurun_out/r06/bench_default.err
tail -c 600 gpurun_out/r06/bench_default.json
( for hog in 0 64; do
    for w in poll runtime; do
      echo "=== LDOT_HOST_WAIT=$w hog=$hog"
      LDOT_HOST_WAIT=$w timeout 600 python tools/host_stall_probe.py --skip-torch-only --evals 8000 --hog $hog 2>&1 | grep -v amdgpu.ids
    done
  done ) > gpurun_out/r06/host_stall_probe2.txt
cat gpurun_out/r06/host_stall_probe2.txt
bash tools/r06_mining_prof.sh > gpurun_out/r06/mining_prof4.log 2>&1
for f in gpurun_out/r06/mining_kernels_*top*_ids.txt gpurun_out/r06/mining_kernels_*top*_exact.txt; do mv $f ${f%.txt}_v4.txt; done
