#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/r5_pmc.sh 2>&1 | tail -60
bash tools/r5_final_gpu.sh 2>&1 | tail -12
