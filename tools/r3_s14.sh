#!/bin/bash
# round-3 session 14: LDS bank-conflict counters of the 3x3 weight-gradient kernel
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES -d gpurun_out/pmc_wg -o wg --output-format csv -- python tools/pmc_wgrad.py > gpurun_out/s14.log 2>&1
ls -R gpurun_out/pmc_wg | head -20
python - <<'P'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_wg/**/*counter_collection.csv', recursive=True)
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r.get('Kernel_Name', '')[:60]
        agg[(k, r.get('Grid_Size'), r.get('LDS_Block_Size'))][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'wgrad_kernel' in k[0]:
        print(k, dict(v))
P
tail -5 gpurun_out/s14.log
