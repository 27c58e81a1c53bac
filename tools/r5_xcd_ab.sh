cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
for t in "1=0" "1=1" "1=2" "1=0" "1=1"; do AVSR_TUNE=$t timeout 300 python bench.py --fixed A --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/s16.json 2>gpurun_out/s16.err; echo "tune $t $(python -c "import json;d=json.load(open('gpurun_out/s16.json'));print(d['ms_per_step'])")"; done
