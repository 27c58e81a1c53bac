#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_attention.py tests/test_kernels_basic.py tests/test_conv_kernels.py tests/test_modules.py -x -q -m gpu -k "attention or split or stem or hpf or bench_geometry" 2>&1 | tail -8
B="python bench.py --no-cpu-baseline"
timeout 900 $B --steps 16 --warmup 4 > $O/r3b_bench_default.json 2> $O/r3b_bench_default.err; echo "default rc=$?"; tail -c 1500 $O/r3b_bench_default.json
bash tools/gpu_timeline.sh r3b_hpf --mode hpf
timeout 600 $B --mode precise --no-roofline --steps 6 --warmup 2 --no-precise-leg > $O/r3b_bench_precise.json 2> $O/r3b_bench_precise.err; echo "precise rc=$?"; tail -c 600 $O/r3b_bench_precise.json
