#!/bin/bash
# round-3 final gate on the GPU box: the -m gpu suite, smoke(), and the driver's default bench command
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 ) > gpurun_out/r3_final_gputests.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> gpurun_out/r3_final_gputests.txt
cat gpurun_out/r3_final_gputests.txt
( time timeout 900 python bench.py > gpurun_out/r3_final_bench_default.json 2> gpurun_out/r3_final_bench_default.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('gpurun_out/r3_final_bench_default.json').readline())
print(d['ms_per_step'], d['value'], 'roofline', d['roofline']['frac'], d['roofline'].get('traffic'), 'hpf', d['precise']['ms_per_step'], d['precise']['parity']['dec_logits_rel_l2'], 'cpu', d['cpu_baseline']['value'])"
