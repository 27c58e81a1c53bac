#!/bin/bash
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for t in "tests/test_trainer_standin.py::test_trainer_native_step_replays_hipgraphs" "tests/test_trainer_standin.py::test_trainer_protocol_auto_and_native_train_alike"; do
  echo "=== $t"; AMD_LOG_LEVEL=0 timeout 600 python -X faulthandler -m pytest "$t" -x -q -m gpu -s 2>&1 | grep -v "Extension modules" | tail -40
done
