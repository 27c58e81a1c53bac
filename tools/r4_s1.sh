#!/bin/bash
# round 4, session 1: first hardware run of the mixed mode (f16 encoder forward): parity at batch A, step time, timeline
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 300 python /root/repo/tools/mixed_smoke.py video 2>&1 | grep -v Warn | tail -4
for m in mixed hpf bf16; do
timeout 300 python bench.py --mode $m --fixed A --no-cpu-baseline --no-roofline --no-precise-leg --steps 12 --warmup 3 2>gpurun_out/s1_$m.err | tee gpurun_out/s1_bench_${m}_fixedA.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['parity']; print('$m', d['ms_per_step'], d['value'], 'logits', p['dec_logits_rel_l2'], 'ctc_logp', p['ctc_logp_rel_l2'], 'grad cos', p['grad_sample_cos_min'], 'loss err', p['loss_rel_err'])" || tail -5 gpurun_out/s1_$m.err
done
timeout 300 python bench.py --mode mixed --no-cpu-baseline --no-roofline --no-precise-leg --no-parity --steps 16 --warmup 4 2>/dev/null | tee gpurun_out/s1_bench_mixed_default.json | cut -c1-300
bash tools/gpu_timeline.sh s1_mixed --mode mixed --no-precise-leg > gpurun_out/s1_tl.out 2>&1; head -70 gpurun_out/s1_mixed_timeline.txt
