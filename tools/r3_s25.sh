#!/bin/bash
# final hpf evidence after the twin producers: default line (bf16 + hpf leg), hpf, hpf fixed A, GPU tests of the touched suites
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_modules.py tests/test_convmod_kernels.py tests/test_attention.py tests/test_e2e_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r3_final_bench_default.json 2> $O/r3_final_bench_default.err; echo "default rc=$?"
timeout 600 python bench.py --no-cpu-baseline --mode hpf --no-roofline --steps 16 --warmup 4 > $O/r3_final_bench_hpf.json 2>/dev/null; echo "hpf rc=$?"
timeout 600 python bench.py --no-cpu-baseline --mode hpf --fixed A --no-roofline --no-parity > $O/r3_final_bench_hpf_fixedA.json 2>/dev/null; echo "hpfA rc=$?"
bash tools/gpu_timeline.sh r3_final_hpf --mode hpf > /dev/null 2>&1; echo "timeline rc=$?"
python - <<'P'
import json
for n in ("default","hpf","hpf_fixedA"):
    d=json.loads(open(f"gpurun_out/r3_final_bench_{n}.json").readline())
    print(n, d["ms_per_step"], d["value"], d.get("precise",{}).get("ms_per_step"), (d.get("parity") or {}).get("dec_logits_rel_l2"))
P
