#!/bin/bash
# one-step kernel timeline of the graph-replayed bench (fixed batch A) -> gpurun_out/<tag>_timeline.txt
tag=${1:-tl}; shift
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/${tag}_tlprof
timeout 500 rocprofv3 --kernel-trace -d gpurun_out/${tag}_tlprof -o r -- python bench.py --no-cpu-baseline --no-roofline --no-parity --fixed A --steps 6 --warmup 3 "$@" > gpurun_out/${tag}_tl.log 2>&1
db=$(find gpurun_out/${tag}_tlprof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$db" gpurun_out/${tag}_timeline.txt 2 | head -60
rm -rf gpurun_out/${tag}_tlprof
tail -1 gpurun_out/${tag}_tl.log | cut -c100-260
