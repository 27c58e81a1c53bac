#!/bin/bash
# kernel-trace profile of the default bench command -> gpurun_out/<tag>_kernel_stats.txt
tag=${1:-prof}; shift
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/${tag}_prof
timeout 500 rocprofv3 --kernel-trace -d gpurun_out/${tag}_prof -o r -- python bench.py --no-cpu-baseline --no-roofline --no-parity "$@" > gpurun_out/${tag}_prof.log 2>&1
db=$(find gpurun_out/${tag}_prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" gpurun_out/${tag}_kernel_stats.txt > /dev/null 2>&1
find gpurun_out/${tag}_prof -name "*.db" -delete
tail -1 gpurun_out/${tag}_prof.log | cut -c100-260
