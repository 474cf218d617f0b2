# A/B of the row-resident fp64 linear-layer kernels at C5 + the fp64 GPU tests
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_model.py -m gpu -q -k "c5 or f64 or float64 or operator" 2>&1 | tail -6 > gpurun_out/f64rows_pytest.log
timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/f64rows_bench_on.log 2> gpurun_out/f64rows_stages_on.log
AA_F64_ROWS=0 timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/f64rows_bench_off.log 2> gpurun_out/f64rows_stages_off.log
tail -3 gpurun_out/f64rows_pytest.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/f64rows_bench_on.log gpurun_out/f64rows_bench_off.log
paste <(grep "gemm_" gpurun_out/f64rows_stages_on.log | awk '{print $2, $3, $(NF-1)}') <(grep "gemm_" gpurun_out/f64rows_stages_off.log | awk '{print $3, $(NF-1)}')
