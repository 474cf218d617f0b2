cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_full_size.py tests/test_gemm_accuracy.py tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -2
timeout 900 python bench.py --workload c5 --steps 5 --warmup 2 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/r03_o_bench_c5.log 2> gpurun_out/r03_o_stages_c5.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03_o_bench_c5.log; grep "stage. gemm" gpurun_out/r03_o_stages_c5.log | cut -c1-110
