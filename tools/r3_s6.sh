#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_kernels.py -x -q -m gpu -k "c64 or conv2d" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-roofline --no-parity --no-precise-leg --fixed A"
for i in 1 2; do
AVSR_TUNE=12=1 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c64 tiled     ', d['ms_per_step'])"
timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c64 persistent', d['ms_per_step'])"
done
bash tools/gpu_timeline.sh r3e_bf16 --no-precise-leg
