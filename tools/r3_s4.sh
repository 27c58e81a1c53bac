#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_basic.py tests/test_conv_kernels.py tests/test_modules.py -x -q -m gpu -k "split or hpf" 2>&1 | tail -3
bash tools/gpu_timeline.sh r3c_hpf --mode hpf
timeout 600 python tools/bench_decode.py > $O/r3_decode_throughput.json 2> $O/r3_decode.err; echo "decode rc=$?"; tail -5 $O/r3_decode.err; tail -c 300 $O/r3_decode_throughput.json
