#!/bin/bash
# round 6 session 8: speed-of-light table of the Conformer GEMMs; is the test_modules failure of the full-suite run reproducible in-file
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/microbench_sol.py 2>&1 | grep -v amdgpu.ids | tee $O/r6_microbench_sol.txt
for i in 1 2; do timeout 900 python -m pytest tests/test_modules.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -12; done
