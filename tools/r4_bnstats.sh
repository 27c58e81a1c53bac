#!/bin/bash
# round 4: BatchNorm statistics from the split-plane convolution / stem epilogues -- tests, parity at the benchmarked shapes, step time A/B
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mixed_mode.py tests/test_bench_parity.py tests/test_e2e_gpu.py -x -q -m gpu -k "leaves or e2e_small or (mixed and (A or B or AA)) or hipgraph" -s 2>&1 | grep -E "PARITY.*mixed|passed|failed|Error" | cut -c1-330
for f in 1 0; do
AVSR_FUSE_BN_STATS=$f timeout 300 python bench.py --fixed A --no-cpu-baseline --no-roofline --no-bf16-leg --no-parity --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fuse=$f fixed A', d['ms_per_step'])"
AVSR_FUSE_BN_STATS=$f timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-bf16-leg --no-parity --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fuse=$f default', d['ms_per_step'], d['value'])"
done
