#!/bin/bash
# Round-3 evidence run (GPU box): every throughput figure DESIGN / README quote, as raw JSON lines under gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
B="python bench.py --no-cpu-baseline"
run() { name=$1; shift; timeout 600 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -c 300 $O/$name.json | head -c 300)"; }
run r3_bench_default $B --steps 16 --warmup 4
run r3_bench_fixedA $B --fixed A --no-roofline --no-parity
run r3_bench_fixedB $B --fixed B --no-roofline --no-parity
run r3_bench_precise $B --precise --no-roofline --steps 4 --warmup 2
run r3_bench_audio $B --modality audio --no-roofline
run r3_bench_eager $B --no-graph --no-roofline --no-parity
run r3_bench_av3200 python tools/bench_av.py
python tools/kv_fault_probe.py > $O/r3_kv_fault_probe.txt 2>&1; tail -30 $O/r3_kv_fault_probe.txt
