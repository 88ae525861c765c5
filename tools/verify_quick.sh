O=gpurun_out/ev_r05_reentry; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -3 $O/pytest_gpu.log; tail -1 $O/smoke.log; cut -c1-700 $O/bench.json
