#!/bin/bash
# round 4, session 8: the one-call-per-step beam search (csrc/decode.hip) on the MI355X -- parity tests, decode throughput, kernel stats of one search
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decoding.py tests/test_e2e_gpu.py -x -q -m gpu -k "decod or beam or prefix or scorers" 2>&1 | tail -8 > $O/r4_s8_tests.txt; cat $O/r4_s8_tests.txt
timeout 600 python tools/bench_decode.py --reps 2 > $O/r4_decode_throughput.json 2> $O/r4_decode_throughput.err; tail -12 $O/r4_decode_throughput.err
