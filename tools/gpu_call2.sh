#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
./tools/probes/xcc_probe > gpurun_out/c2_xcc.log 2>&1; cat gpurun_out/c2_xcc.log
timeout 300 python tools/microbench_xcd.py > gpurun_out/c2_xcd.log 2>&1; tail -6 gpurun_out/c2_xcd.log
timeout 300 python -m pytest tests/test_optim.py tests/test_train_eval_loops.py tests/test_boundary.py -x -q -m gpu > gpurun_out/c2_tests.log 2>&1; tail -3 gpurun_out/c2_tests.log
for f in 0 1; do
rm -rf gpurun_out/c2_prof$f
AVSR_FUSE_STEM_POOL=$f timeout 400 rocprofv3 --kernel-trace -d gpurun_out/c2_prof$f -o r -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/c2_prof$f.log 2>&1
db=$(find gpurun_out/c2_prof$f -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" gpurun_out/c2_kernel_stats_pool$f.txt > /dev/null 2>&1
find gpurun_out/c2_prof$f -name "*.db" -delete
done
grep -i "pool\|stem\|bn_" gpurun_out/c2_kernel_stats_pool1.txt | head -20
