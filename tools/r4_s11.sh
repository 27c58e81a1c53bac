#!/bin/bash
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
bash tools/r4_s10.sh
timeout 600 python tools/bench_decode.py --reps 3 > $O/r4_decode_throughput.json 2> $O/r4_decode_throughput.err; grep -h "ms_per_token" $O/r4_decode_throughput.err | cut -c1-230
