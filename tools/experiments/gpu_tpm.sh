# A/B of the w0-recomputing tensor-product kernels (aa_tp_mfma.hip): GPU tests, C4 stage tables with and without
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tp_mfma.py -m gpu -q 2>&1 | tail -8 > gpurun_out/tpm_pytest.log
AA_TP_MFMA=1 timeout 300 python bench.py --steps 10 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/tpm_bench_on.log 2> gpurun_out/tpm_stages_on.log
AA_TP_MFMA=0 timeout 300 python bench.py --steps 10 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/tpm_bench_off.log 2> gpurun_out/tpm_stages_off.log
tail -4 gpurun_out/tpm_pytest.log
grep -h "tp_m\|gc_64x64_64x64_64x256" gpurun_out/tpm_stages_on.log gpurun_out/tpm_stages_off.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/tpm_bench_on.log gpurun_out/tpm_bench_off.log
grep -o '"parity_sample": {[^}]*}' gpurun_out/tpm_bench_on.log
