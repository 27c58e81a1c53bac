#!/bin/bash
# Round-6 counter passes (GPU box), mixed mode (what bench.py times): HBM-side traffic (FETCH_SIZE, WRITE_SIZE), MFMA-pipe busy
# cycles and the L2 (TCC) hit / miss / fabric-request counters of ONE eager training step of the batch bench.py's roofline legs run on (AVSR_PMC_SHAPE=bench: shape 3 of the default run's 8),
# each counter set in its own rocprofv3 pass (the guide's rule) -> gpurun_out/r6_* (copy the summaries to profiles/).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
pass() {  # name, counters...
  local n=$1; shift
  local d=$O/r6_pmc_$n; rm -rf $d
  AVSR_PMC_SHAPE=bench AVSR_PMC_INFO=$O/r6_pmc_step_info.json timeout 400 rocprofv3 --kernel-trace --pmc "$@" -d $d -o r -- python tools/pmc_step.py > $O/r6_pmc_$n.log 2>&1
  echo "$n ($*) rc=$? $(tail -1 $O/r6_pmc_$n.log | cut -c1-120)"
}
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass MFMA SQ_VALU_MFMA_BUSY_CYCLES
pass TCC_HITMISS TCC_HIT_sum TCC_MISS_sum
pass TCC_EA TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum
python tools/pmc_report.py $O/r6_pmc_FETCH_SIZE $O/r6_pmc_WRITE_SIZE $O/r6_hbm_traffic.txt $O/r6_pmc_step_info.json > /dev/null 2> $O/r6_pmc_report.err; echo "report rc=$?"
python tools/pmc_mfma.py $O/r6_pmc_MFMA $O/r6_mfma_busy.txt > /dev/null 2>> $O/r6_pmc_report.err; echo "mfma rc=$?"
python tools/pmc_tcc.py $O/r6_pmc_TCC_HITMISS $O/r6_pmc_TCC_EA $O/r6_tcc.txt 2>> $O/r6_pmc_report.err | cut -c1-200; echo "tcc rc=$?"
head -24 $O/r6_hbm_traffic.txt | cut -c1-170; head -20 $O/r6_mfma_busy.txt | cut -c1-150
tail -5 $O/r6_pmc_report.err
find $O -name "*.db" -size +20M -delete
