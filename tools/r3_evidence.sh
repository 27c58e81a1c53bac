#!/bin/bash
# Round-3 evidence run (GPU box): every throughput figure DESIGN / README quote, as raw JSON lines under gpurun_out/r3_final_*.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
B="python bench.py --no-cpu-baseline"
run() { name=$1; shift; timeout 900 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -c 200 $O/$name.json | head -c 200)"; }
run r3_final_bench_default python bench.py --steps 20 --warmup 5
run r3_final_bench_fixedA $B --fixed A --no-roofline --no-parity --no-precise-leg
run r3_final_bench_fixedB $B --fixed B --no-roofline --no-parity --no-precise-leg
run r3_final_bench_hpf $B --mode hpf --no-roofline --steps 16 --warmup 4
run r3_final_bench_hpf_fixedA $B --mode hpf --fixed A --no-roofline --no-parity
run r3_final_bench_precise $B --mode precise --no-roofline --steps 8 --warmup 2
run r3_final_bench_audio $B --modality audio --no-roofline --no-precise-leg
run r3_final_bench_eager $B --no-graph --no-roofline --no-parity --no-precise-leg
run r3_final_bench_av3200 python tools/bench_av.py
run r3_final_bench_av3200_eager python tools/bench_av.py --no-graph
run r3_final_bench_audio_babble $B --modality audio --babble --no-roofline --no-precise-leg
run r3_final_decode_throughput python tools/bench_decode.py
bash tools/gpu_timeline.sh r3_final_bf16 --no-precise-leg > /dev/null 2>&1; echo "timeline bf16 rc=$?"
bash tools/gpu_timeline.sh r3_final_hpf --mode hpf > /dev/null 2>&1; echo "timeline hpf rc=$?"
bash tools/gpu_prof.sh r3_final --no-precise-leg > /dev/null 2>&1; echo "kernel stats rc=$?"
ls -la $O | grep r3_final
