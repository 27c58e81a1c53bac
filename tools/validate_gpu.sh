#!/bin/bash
# One-shot validation on an MI355X box (what every round ends with):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/validate_gpu.sh r2'
# Writes gpurun_out/<tag>_{tests,smoke,bench}.log and gpurun_out/<tag>_kernel_stats.txt (rocprofv3 kernel-trace summary of the
# default bench command); copy the last two into profiles/ to have them judged.
tag=${1:-rX}
mkdir -p gpurun_out
timeout 700 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; tail -2 gpurun_out/${tag}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log
python bench.py > gpurun_out/${tag}_bench.log 2>&1; tail -1 gpurun_out/${tag}_bench.log | cut -c1-320
export TMPDIR=/tmp
rm -rf gpurun_out/${tag}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_prof -o r -- python bench.py --no-cpu-baseline > gpurun_out/${tag}_prof.log 2>&1
db=$(find gpurun_out/${tag}_prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" gpurun_out/${tag}_kernel_stats.txt > /dev/null 2>&1
head -12 gpurun_out/${tag}_kernel_stats.txt
find gpurun_out/${tag}_prof -name "*.db" -delete
