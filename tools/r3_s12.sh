#!/bin/bash
# round-3 session 12: bf16-mode beam-search pin, configs[3] with the babble transform in the timed step, wgrad ablations
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -k beam_search -x 2>&1 | tail -15 > gpurun_out/s12_beam.txt
timeout 300 python tools/microbench_wgrad.py > gpurun_out/s12_wgrad.txt 2>&1
timeout 400 python bench.py --modality audio --babble --no-cpu-baseline > gpurun_out/s12_audio_babble.json 2> gpurun_out/s12_audio_babble.err
timeout 400 python bench.py --modality audio --babble --no-graph --no-cpu-baseline --no-roofline > gpurun_out/s12_audio_babble_eager.json 2>> gpurun_out/s12_audio_babble.err
tail -3 gpurun_out/s12_beam.txt; cat gpurun_out/s12_wgrad.txt; cut -c1-400 gpurun_out/s12_audio_babble.json; cut -c1-300 gpurun_out/s12_audio_babble_eager.json; tail -5 gpurun_out/s12_audio_babble.err
