#!/bin/bash
# robustness sweep: other max-frames / shape counts / modes must run clean
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-roofline --no-parity --no-precise-leg"
run() { name=$1; shift; timeout 600 "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.loads(open('gpurun_out/$name.json').readline()); print(d['ms_per_step'], d['value'], d['config'].get('batch_shapes'))" 2>&1 | tail -1 | cut -c1-200)"; }
run s18_mf3200 $B --max-frames 3200 --steps 6
run s18_mf800 $B --max-frames 800 --steps 6
run s18_shapes16 $B --shapes 16 --steps 16 --warmup 16
run s18_audio_A $B --modality audio --fixed A
run s18_hpf_mf3200 $B --mode hpf --max-frames 3200 --steps 4 --warmup 2
run s18_noopt $B --no-optimizer --steps 6
