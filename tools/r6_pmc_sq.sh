#!/bin/bash
# round 6: where the waves of each kernel spend their cycles (SQ counters, one rocprofv3 pass over one eager step of the bench's batch shape)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL"; do
  n=$(echo $set | cut -d' ' -f1)
  d=$O/r6_pmc_sq_$n; rm -rf $d
  AVSR_PMC_SHAPE=bench timeout 400 rocprofv3 --kernel-trace --pmc $set -d $d -o r -- python tools/pmc_step.py > $O/r6_pmc_sq_$n.log 2>&1
  echo "$n rc=$? $(tail -1 $O/r6_pmc_sq_$n.log | cut -c1-100)"
done
python tools/pmc_sq.py $O/r6_pmc_sq_SQ_WAVE_CYCLES,$O/r6_pmc_sq_SQ_LDS_BANK_CONFLICT $O/r6_sq_waits.txt | head -30 | cut -c1-170
