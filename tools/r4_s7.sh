#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_convmod_kernels.py tests/test_modules.py tests/test_bench_parity.py tests/test_e2e_gpu.py -q -m gpu -x -k "pool or stem or (mixed and (A or B)) or (bf16 and (A or B)) or golden" 2>&1 | tail -3
for m in mixed bf16; do
timeout 300 python bench.py --mode $m --fixed A --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); p=d['parity']; print('$m fixed A', d['ms_per_step'], 'logits', p['dec_logits_rel_l2'], 'grad cos', p['grad_sample_cos_min'], p['grad_sample_rel_l2_median'])"
done
bash tools/gpu_timeline.sh s7_bf16 --mode bf16 > /dev/null 2>&1; grep -E "bn_act_pool3|bn_pool3|one step" gpurun_out/s7_bf16_timeline.txt | head -4
