#!/bin/bash
# Round-4 evidence run (GPU box): every throughput figure DESIGN / README quote, as raw JSON lines under gpurun_out/r4_final_*.
# The default numerical mode is "mixed" (bench.py); --mode bf16 / hpf / precise are the comparison lines.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
B="python bench.py --no-cpu-baseline"
run() { name=$1; shift; timeout 900 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(tail -c 300 $O/$name.json | head -c 0)$(python -c "
import json,sys
try:
    d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print(d.get('ms_per_step'), d.get('value'))
except Exception as e: print('?', e)
")"; }
run r4_final_bench_default python bench.py --steps 20 --warmup 5
run r4_final_bench_fixedA $B --fixed A --no-roofline --no-parity --no-bf16-leg
run r4_final_bench_fixedB $B --fixed B --no-roofline --no-parity --no-bf16-leg
run r4_final_bench_bf16 $B --mode bf16 --no-roofline --steps 16 --warmup 4
run r4_final_bench_bf16_fixedA $B --mode bf16 --fixed A --no-roofline --no-parity
run r4_final_bench_hpf $B --mode hpf --no-roofline --steps 16 --warmup 4 --no-bf16-leg
run r4_final_bench_eager $B --no-graph --no-roofline --no-parity --no-bf16-leg
run r4_final_bench_audio $B --modality audio --no-roofline --no-bf16-leg
run r4_final_bench_audio_babble $B --modality audio --babble --no-roofline --no-bf16-leg
run r4_final_bench_av3200 python tools/bench_av.py
run r4_final_bench_av3200_bf16 python tools/bench_av.py --mode bf16
for v in "AVSR_DDP=buckets-graph" "AVSR_DDP=buckets-graph1" "AVSR_DDP=buckets-graph AVSR_GRAD_WIRE=bf16" "AVSR_DDP=torch"; do
  n=r4_final_dp1_$(echo $v | tr -c 'a-zA-Z0-9\n' '_')
  env $v AVSR_BENCH_FORCE_DP=1 timeout 300 $B --no-roofline --no-parity --no-bf16-leg --steps 16 --warmup 4 > $O/$n.json 2> $O/$n.err
  python -c "import json; d=json.loads(open('$O/$n.json').readline()); c=d['config']; print('DP1 $v', d['ms_per_step'], {k: c[k] for k in ('ddp_mode','communicators','grad_wire','rccl_ranks')})"
done
bash tools/gpu_timeline.sh r4_final_mixed --no-bf16-leg > /dev/null 2>&1; echo "timeline mixed rc=$?"
bash tools/gpu_timeline.sh r4_final_bf16 --mode bf16 > /dev/null 2>&1; echo "timeline bf16 rc=$?"
bash tools/gpu_prof.sh r4_final --no-bf16-leg > /dev/null 2>&1; echo "kernel stats rc=$?"
ls $O | grep r4_final | head -40
