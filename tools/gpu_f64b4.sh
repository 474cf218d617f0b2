# A/B: fp64 linear layers on the 4-block 4x4x4 MFMA vs the 16x16x4 form (C5), 3 vs 2 waves per SIMD
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_model.py -m gpu -q -k "c5 or f64 or float64 or operator" 2>&1 | tail -4 > gpurun_out/f64b4_pytest.log
run() { timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/f64b4_bench_$1.log 2> gpurun_out/f64b4_stages_$1.log; }
run b4
AA_F64_MFMA16=1 run m16
ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_occ2.so run b4_occ2
tail -2 gpurun_out/f64b4_pytest.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/f64b4_bench_b4.log gpurun_out/f64b4_bench_m16.log gpurun_out/f64b4_bench_b4_occ2.log
paste <(grep "gemm_" gpurun_out/f64b4_stages_b4.log | awk '{print $2, $3, $(NF-1)}') <(grep "gemm_" gpurun_out/f64b4_stages_m16.log | awk '{print $3, $(NF-1)}') <(grep "gemm_" gpurun_out/f64b4_stages_b4_occ2.log | awk '{print $3, $(NF-1)}')
