#!/bin/bash
# round 6, GPU call 3: whole GPU suite on the split library + new kernels, mining bench / kernel stats (incl. dense-mode top-1000), stall probe
# with and without the spin policy
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_tests.txt 2>&1
tail -6 gpurun_out/r06/gpu_tests.txt
timeout 600 python tools/mining_bench.py > gpurun_out/r06/mining_bench2.json 2> gpurun_out/r06/mining_bench2.err
bash tools/r06_mining_prof.sh > gpurun_out/r06/mining_prof2.log 2>&1
for f in gpurun_out/r06/mining_kernels_*; do mv $f ${f%.txt}_v2.txt; done
( cd /tmp; MINING_SEARCH_MODE=1 timeout 300 python $GRAFT_REPO_ROOT/tools/mining_one.py i2t 1000 ids 3 ) > gpurun_out/r06/mining_i2t_top1000_dense_mode.txt 2>&1
python - <<'PY' >> gpurun_out/r06/mining_i2t_top1000_dense_mode.txt
import time, torch, sys
sys.path.insert(0, '.')
from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
img, txt = s2_embeddings(29000, 768, 5, seed=11, device='cuda')
for mode in (0, 1):
    ix = FlatIPIndex(768); ix.add(txt)
    if mode: ix.set_option(1, mode)
    for k in (1000,):
        ix.search_tensors(img, k, ids_only=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): ix.search_tensors(img, k, ids_only=True)
        torch.cuda.synchronize()
        print('i2t 29k x 145k top-%d ids-only, LDOT_OPT_MODE %d: %.2f ms' % (k, mode, (time.perf_counter() - t0) / 3 * 1e3), ix.last_regime()['path'])
PY
echo "== default host wait policy (spin)" > gpurun_out/r06/stall_probe_policy.txt
STALL_PROBE_ROUNDS=10 timeout 600 python tools/stall_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/stall_probe_policy.txt
echo "== LDOT_HOST_WAIT=runtime" >> gpurun_out/r06/stall_probe_policy.txt
LDOT_HOST_WAIT=runtime STALL_PROBE_ROUNDS=10 timeout 600 python tools/stall_probe.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/stall_probe_policy.txt
cat gpurun_out/r06/stall_probe_policy.txt
