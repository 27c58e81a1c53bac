#!/bin/bash
# end-of-round evidence: [GPU suite, smoke,] counter passes + reports, kernel-trace summary, default bench line.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_final.sh r2 [notests]'
# Tracked copies go to profiles/ (the .db files are deleted: gpurun_out/ must stay under 64 MiB to be copied back).
tag=${1:-r2}
mkdir -p gpurun_out profiles; cd /root/repo; export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1; tail -2 gpurun_out/${tag}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  AVSR_PMC_INFO=gpurun_out/${tag}_pmc_step_info.json timeout 400 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -o r -- python tools/pmc_step.py > gpurun_out/${tag}_pmc_$c.log 2>&1
done
python tools/pmc_report.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/${tag}_hbm_traffic.txt gpurun_out/${tag}_pmc_step_info.json > /dev/null 2>&1
head -18 gpurun_out/${tag}_hbm_traffic.txt | cut -c1-200
rm -rf gpurun_out/pmc_SQ
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pmc_SQ -o r -- python tools/pmc_step.py > gpurun_out/${tag}_pmc_SQ.log 2>&1
(cd tools && python pmc_mfma.py ../gpurun_out/pmc_SQ ../gpurun_out/${tag}_mfma_busy.txt > /dev/null 2>&1)
head -12 gpurun_out/${tag}_mfma_busy.txt | cut -c1-160
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ
bash tools/gpu_prof.sh ${tag}
cp gpurun_out/${tag}_hbm_traffic.txt gpurun_out/${tag}_hbm_traffic.json gpurun_out/${tag}_mfma_busy.txt gpurun_out/${tag}_kernel_stats.txt profiles/ 2>/dev/null
python bench.py > gpurun_out/${tag}_bench.log 2>&1; tail -1 gpurun_out/${tag}_bench.log | cut -c1-400
