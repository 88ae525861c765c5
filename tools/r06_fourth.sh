#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_tests.txt 2>&1
tail -6 gpurun_out/r06/gpu_tests.txt
timeout 600 python tools/mining_bench.py > gpurun_out/r06/mining_bench3.json 2> gpurun_out/r06/mining_bench3.err
bash tools/r06_mining_prof.sh > gpurun_out/r06/mining_prof3.log 2>&1
for f in gpurun_out/r06/mining_kernels_*top*_ids.txt gpurun_out/r06/mining_kernels_*top*_exact.txt; do mv $f ${f%.txt}_v3.txt; done
timeout 600 python tools/host_stall_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/host_stall_probe.txt
cat gpurun_out/r06/host_stall_probe.txt
bash tools/r06_counters.sh > gpurun_out/r06/counters.log 2>&1
tail -30 gpurun_out/r06/counters.log
