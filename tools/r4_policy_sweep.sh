#!/bin/bash
# round 4, session 2: forward-format policy sweep of the mixed mode on the MI355X (fixed batch A: step time + parity against the reference golden)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
run() {
  AVSR_MIXED_POLICY=$1 timeout 300 python bench.py --mode mixed --fixed A --no-cpu-baseline --no-roofline --no-precise-leg --steps 12 --warmup 3 2>gpurun_out/s2.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['parity']; print('$1', '| ms', d['ms_per_step'], '| logits', p['dec_logits_rel_l2'], 'ctc_logp', p['ctc_logp_rel_l2'], 'grad cos', p['grad_sample_cos_min'], 'grad relL2 med', p['grad_sample_rel_l2_median'], 'loss', p['loss_rel_err'])" || tail -5 gpurun_out/s2.err
}
run encoder=f16 | tee gpurun_out/s2_sweep.txt
run encoder=f16,trunk3=f16,trunk4=f16 | tee -a gpurun_out/s2_sweep.txt
run encoder=f16,trunk2=f16,trunk3=f16,trunk4=f16 | tee -a gpurun_out/s2_sweep.txt
run encoder=f16,trunk1=f16,trunk2=f16,trunk3=f16,trunk4=f16 | tee -a gpurun_out/s2_sweep.txt
run encoder=f16,decoder=f16,dec_out=f16 | tee -a gpurun_out/s2_sweep.txt
run encoder=f16,trunk2=f16,trunk3=f16,trunk4=f16,decoder=f16,dec_out=f16 | tee -a gpurun_out/s2_sweep.txt
run encoder=f16,trunk1=f16,trunk2=f16,trunk3=f16,trunk4=f16,decoder=f16,dec_out=f16 | tee -a gpurun_out/s2_sweep.txt
run encoder=f16,trunk2=f16,trunk3=f16,trunk4=f16,decoder=f16 | tee -a gpurun_out/s2_sweep.txt
timeout 600 python -m pytest tests/test_modules.py -q -m gpu -x -k "hpf or e2e_small" 2>&1 | tail -2
