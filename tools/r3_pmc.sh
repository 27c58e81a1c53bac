#!/bin/bash
# Round-3 counter passes (GPU box): HBM-side traffic (FETCH_SIZE, WRITE_SIZE) and MFMA-pipe busy cycles of ONE eager training
# step of the bench workload's middle bucket, each counter in its own rocprofv3 pass (the guide's rule) -> profiles/r3_*.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  d=$O/r3_pmc_$c; rm -rf $d
  AVSR_PMC_INFO=$O/r3_pmc_step_info.json timeout 600 rocprofv3 --kernel-trace --pmc $c -d $d -o r -- python tools/pmc_step.py > $O/r3_pmc_$c.log 2>&1
  echo "$c rc=$? $(tail -1 $O/r3_pmc_$c.log | cut -c1-120)"
done
python tools/pmc_report.py $O/r3_pmc_FETCH_SIZE $O/r3_pmc_WRITE_SIZE $O/r3_hbm_traffic.txt $O/r3_pmc_step_info.json > /dev/null 2> $O/r3_pmc_report.err; echo "report rc=$?"
python tools/pmc_mfma.py $O/r3_pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/r3_mfma_busy.txt > /dev/null 2>> $O/r3_pmc_report.err; echo "mfma rc=$?"
head -30 $O/r3_hbm_traffic.txt | cut -c1-170; head -25 $O/r3_mfma_busy.txt | cut -c1-150
# keep the merged output small: the raw databases stay on the box
find $O -name "*.db" -size +20M -delete
