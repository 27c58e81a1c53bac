#!/bin/bash
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decoding.py tests/test_e2e_gpu.py -x -q -m gpu -k "decod or beam or prefix or scorers or native" 2>&1 | tail -8 > $O/r4_s10_tests.txt; cat $O/r4_s10_tests.txt
bash tools/r4_s9.sh
