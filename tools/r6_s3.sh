#!/bin/bash
# round 6 session 3: 256-row split-plane tiles on 8 waves: microbench + kernel tests on the device
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python tools/microbench_presplit.py 2>&1 | grep -v amdgpu.ids | tee $O/r6_s3_microbench_presplit.txt
timeout 900 python -m pytest tests/test_mixed_mode.py -x -q -m gpu -k "split8 or h16x2" 2>&1 | tail -4
