#!/bin/bash
# Round-6 evidence run (GPU box): every throughput figure DESIGN / README quote, as raw JSON lines under gpurun_out/r6_final_*.
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
run r6_final_bench_default python bench.py --steps 20 --warmup 5
run r6_final_train_py python -u train.py --synthetic --synthetic-utterances 400 --steps 110 --time-last 30 --exp-dir "" --val-batches 0 --log-every 34
run r6_final_bench_fixedA $B --fixed A --no-roofline --no-parity --no-bf16-leg
run r6_final_bench_fixedB $B --fixed B --no-roofline --no-parity --no-bf16-leg
run r6_final_bench_bf16 $B --mode bf16 --no-roofline --steps 16 --warmup 4
run r6_final_bench_bf16_fixedA $B --mode bf16 --fixed A --no-roofline --no-parity
run r6_final_bench_hpf $B --mode hpf --no-roofline --steps 16 --warmup 4 --no-bf16-leg
run r6_final_bench_eager $B --no-graph --no-roofline --no-parity --no-bf16-leg
run r6_final_bench_audio $B --modality audio --no-roofline --no-bf16-leg
run r6_final_bench_audio_babble $B --modality audio --babble --no-roofline --no-bf16-leg
run r6_final_bench_av3200 python tools/bench_av.py
run r6_final_bench_av3200_bf16 python tools/bench_av.py --mode bf16
run r6_final_bench_deterministic $B --deterministic --no-roofline --no-parity --no-bf16-leg --steps 12 --warmup 3
for v in "AVSR_DDP=auto" "AVSR_DDP=buckets-graph" "AVSR_DDP=buckets-graph1 AVSR_GRAD_WIRE=bf16" "AVSR_DDP=buckets" "AVSR_DDP=torch"; do
  n=r6_final_dp1_$(echo $v | tr -c 'a-zA-Z0-9\n' '_')
  env $v AVSR_BENCH_FORCE_DP=1 timeout 300 $B --no-roofline --no-parity --no-bf16-leg --steps 16 --warmup 4 > $O/$n.json 2> $O/$n.err
  python -c "import json; d=json.loads(open('$O/$n.json').readline()); c=d['config']; print('DP1 $v', d['ms_per_step'], {k: c.get(k) for k in ('ddp_mode','communicators','grad_wire','rccl_ranks')}, (c.get('bucket_overlap') or {}).get('exposed_ms'))"
done
bash tools/gpu_timeline.sh r6_final_mixed --no-bf16-leg > /dev/null 2>&1; echo "timeline mixed rc=$?"
bash tools/gpu_timeline.sh r6_final_bf16 --mode bf16 > /dev/null 2>&1; echo "timeline bf16 rc=$?"
bash tools/gpu_prof.sh r6_final --no-bf16-leg > /dev/null 2>&1; echo "kernel stats rc=$?"
ls $O | grep r6_final | head -40
