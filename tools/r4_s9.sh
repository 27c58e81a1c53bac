#!/bin/bash
# round 4, session 9: kernel-trace of one natively stepped beam search
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for T in 100 400; do
rm -rf $O/dec_prof; timeout 300 rocprofv3 --kernel-trace -d $O/dec_prof -o r -- python tools/prof_decode.py $T > $O/r4_decode_prof_T$T.log 2>&1
db=$(find $O/dec_prof -name "*.db" | head -1); python tools/rocpd_summary.py "$db" $O/r4_decode_kernel_stats_T$T.txt > /dev/null 2>&1; find $O/dec_prof -name "*.db" -delete
grep search $O/r4_decode_prof_T$T.log; head -30 $O/r4_decode_kernel_stats_T$T.txt | cut -c1-150
done
