#!/bin/bash
# round 4: evaluation-path evidence on the MI355X -- decoding tests, kernel-trace summaries of one natively stepped beam search
# (T = 100 and T = 400 frames), decode throughput of the one-call-per-step search against the python-issued step.
#   /usr/local/graft/bin/gpurun --timeout 1700 -- 'bash tools/r4_decode_evidence.sh'   then copy gpurun_out/r4_decode_* to profiles/
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_decoding.py tests/test_e2e_gpu.py -x -q -m gpu -k "decod or beam or prefix or scorers or native" 2>&1 | tail -4 > $O/r4_decode_tests.txt; cat $O/r4_decode_tests.txt
for T in 100 400; do
rm -rf $O/dec_prof; timeout 300 rocprofv3 --kernel-trace -d $O/dec_prof -o r -- python tools/prof_decode.py $T > $O/r4_decode_prof_T$T.log 2>&1
db=$(find $O/dec_prof -name "*.db" | head -1); python tools/rocpd_summary.py "$db" $O/r4_decode_kernel_stats_T$T.txt > /dev/null 2>&1; find $O/dec_prof -name "*.db" -delete
grep search $O/r4_decode_prof_T$T.log; head -14 $O/r4_decode_kernel_stats_T$T.txt | cut -c1-150
done
timeout 600 python tools/bench_decode.py --reps 3 > $O/r4_decode_throughput.json 2> $O/r4_decode_throughput.err; grep -h "ms_per_token" $O/r4_decode_throughput.err | cut -c1-230
