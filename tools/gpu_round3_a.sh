# round 3, GPU call A: default-path certification (un-pinned GPU suite), the virial stress hunt, baseline bench
cd /root/repo
TAG=${1:-r03_a}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/${TAG}_pytest_gpu.log
timeout 600 python tools/virial_stress.py 200 > gpurun_out/${TAG}_virial_stress.log 2>&1
AMD_SERIALIZE_KERNEL=3 timeout 600 python tools/virial_stress.py 100 > gpurun_out/${TAG}_virial_stress_serialized.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --stages > gpurun_out/${TAG}_bench_c4.log 2> gpurun_out/${TAG}_stages_c4.log
tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -4 gpurun_out/${TAG}_virial_stress.log; tail -2 gpurun_out/${TAG}_virial_stress_serialized.log
tail -1 gpurun_out/${TAG}_smoke.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_c4.log; grep stage gpurun_out/${TAG}_stages_c4.log
