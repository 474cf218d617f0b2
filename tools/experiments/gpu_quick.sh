# quick check after a kernel change: GPU parity tests of the staged path + C4 / C5 / C1 stage tables
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_full_size.py tests/test_tp_mfma.py -m gpu -q 2>&1 | tail -5 > gpurun_out/quick_pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/quick_bench_c4.log 2> gpurun_out/quick_stages_c4.log
timeout 300 python bench.py --workload c5 --steps 3 --warmup 1 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/quick_bench_c5.log 2> gpurun_out/quick_stages_c5.log
timeout 300 python bench.py --workload c1 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/quick_bench_c1.log 2>&1
timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/quick_bench_c3.log 2>&1
tail -3 gpurun_out/quick_pytest.log
grep "stage" gpurun_out/quick_stages_c4.log
for f in c4 c5 c1 c3; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/quick_bench_$f.log; done
grep "edge_\|readout" gpurun_out/quick_stages_c5.log
