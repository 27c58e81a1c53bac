cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6; python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > gpurun_out/r4_final_gputests.txt 2>&1; tail -4 gpurun_out/r4_final_gputests.txt
bash tools/r4_decode_evidence.sh 2>&1 | tail -30
timeout 400 python bench.py > gpurun_out/r4_final_bench_default.json 2> gpurun_out/r4_final_bench_default.err; tail -1 gpurun_out/r4_final_bench_default.json | cut -c1-400
