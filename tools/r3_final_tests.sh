#!/bin/bash
# round-3 final gate on the GPU box: the -m gpu suite, smoke(), and which shared objects the processes loaded
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 ) > gpurun_out/r3_final_gputests.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> gpurun_out/r3_final_gputests.txt
cat gpurun_out/r3_final_gputests.txt
