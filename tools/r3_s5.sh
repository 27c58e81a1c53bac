#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_basic.py tests/test_conv_kernels.py tests/test_modules.py tests/test_attention.py -x -q -m gpu -k "split or hpf or kv_fast or bench_geometry" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-roofline --no-parity --no-precise-leg --fixed A"
for i in 1 2; do
AVSR_TUNE=10=1 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kv generic', d['ms_per_step'])"
timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kv fast   ', d['ms_per_step'])"
done
bash tools/gpu_timeline.sh r3d_hpf --mode hpf
bash tools/gpu_timeline.sh r3d_bf16
