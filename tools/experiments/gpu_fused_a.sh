# fused forward: parity on hardware + C4/C3 timing with per-launch stages (fused vs staged A/B)
cd /root/repo
TAG=${1:-r02_v3}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused.py tests/test_hip_model.py tests/test_hip_full_size.py -m gpu -q -s 2>&1 | tail -40 > gpurun_out/${TAG}_pytest_gpu.log
AA_FUSED=1 timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c4.log 2> gpurun_out/${TAG}_stages_c4.log
timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c4_staged.log 2> gpurun_out/${TAG}_stages_c4_staged.log
timeout 600 python bench.py --workload c3 --steps 50 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c3.log 2> gpurun_out/${TAG}_stages_c3.log
tail -6 gpurun_out/${TAG}_pytest_gpu.log
for f in c4 c4_staged c3; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_$f.log; done
cat gpurun_out/${TAG}_stages_c4.log | grep stage
