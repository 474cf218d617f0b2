cd /root/repo
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r01_v19_pytest_gpu.log
python bench.py --steps 50 --warmup 10 --emulate-shard 3/8 --stages --no-cpu-baseline > gpurun_out/r01_v19_bench_shard.log 2> gpurun_out/r01_v19_stages_shard.log
python bench.py --steps 50 --warmup 10 --workload c3 --stages --no-cpu-baseline > gpurun_out/r01_v19_bench_c3.log 2> gpurun_out/r01_v19_stages_c3.log
tail -2 gpurun_out/r01_v19_pytest_gpu.log; grep stage gpurun_out/r01_v19_stages_shard.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/r01_v19_bench_shard.log gpurun_out/r01_v19_bench_c3.log
