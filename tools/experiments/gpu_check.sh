cd /root/repo
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r01_v18_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r01_v18_smoke.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r01_v18_bench.log 2>&1
tail -3 gpurun_out/r01_v18_pytest_gpu.log; tail -2 gpurun_out/r01_v18_smoke.log; tail -1 gpurun_out/r01_v18_bench.log
