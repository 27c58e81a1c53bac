#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
python tools/kv_fault_probe.py > $O/r3_kv_fault_probe2.txt 2>&1; cat $O/r3_kv_fault_probe2.txt
B="python bench.py --no-cpu-baseline"
timeout 600 $B --precise --no-roofline --steps 6 --warmup 2 > $O/r3_bench_precise.json 2> $O/r3_bench_precise.err; echo "precise rc=$?"; tail -c 400 $O/r3_bench_precise.json
bash tools/gpu_timeline.sh r3_precise --precise 
