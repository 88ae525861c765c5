#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05f}; mkdir -p $O; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_row_shuffle.py -m gpu -q 2>&1 | tail -3
timeout 300 python tools/coco_probe.py 2>&1 | tee $O/coco_probe.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kp -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/coco_probe.py > /dev/null 2>&1
head -12 $(find /tmp/kp -name "*kernel_stats.csv" | head -1) | cut -c1-70,180-330 | tee $O/coco_probe_kernel_stats.txt
bash $GRAFT_REPO_ROOT/tools/prof_s2.sh $O
