#!/bin/bash
# the N > 1 code path on ONE rank: C-API world-1 tool at full size (two communicators), the GPU test, and bench.py with the
# data-parallel machinery forced on (AVSR_BENCH_FORCE_DP=1): default (auto = buckets-graph) and torch DDP eager
mkdir -p gpurun_out
timeout 500 python tools/rccl_capi_world1.py 2>&1 | grep "^{" | tail -1 > gpurun_out/r3_rccl_capi_world1_full.json; cut -c1-600 gpurun_out/r3_rccl_capi_world1_full.json
timeout 600 python -m pytest tests/test_rccl_single.py -q -m gpu -x 2>&1 | tail -2
export AVSR_BENCH_FORCE_DP=1
for mode in auto torch; do
  timeout 600 python bench.py $( [ $mode = auto ] || echo --ddp $mode ) --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/r3_dp1_$mode.json 2> gpurun_out/r3_dp1_$mode.err
  echo "$mode rc=$? $(python -c "import json; d=json.loads([l for l in open('gpurun_out/r3_dp1_$mode.json') if l.startswith('{')][0]); print(d['ms_per_step'], d['value'], d['config']['final_loss'], d['config']['workload'][-200:])" 2>&1 | tail -1)"
  grep -i "failed\|falling back" gpurun_out/r3_dp1_$mode.err | head -3
done
