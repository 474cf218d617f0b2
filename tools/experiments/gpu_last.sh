cd /root/repo
TAG=${1:-r01_v29}
timeout 90 python -m pytest tests -m gpu -q < /dev/null 2>&1 | tail -4 > gpurun_out/${TAG}_pytest_gpu.log
timeout 30 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/${TAG}_smoke.log 2>&1
timeout 80 python bench.py < /dev/null > gpurun_out/${TAG}_bench_default.log 2> gpurun_out/${TAG}_bench_default.err
timeout 30 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --no-profile < /dev/null > gpurun_out/${TAG}_bench_c5.log 2>&1
tail -2 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_default.log gpurun_out/${TAG}_bench_c5.log
grep -o '"cpu_baseline": {[^}]*}' gpurun_out/${TAG}_bench_default.log | cut -c1-300
