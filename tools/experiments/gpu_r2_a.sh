# round 2, first GPU pass: parity tests (incl. full-size C4/C5 blocks, GEMM accuracy, op-seam modifier), bench lines
cd /root/repo
TAG=${1:-r02_v1}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -80 > gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --stages > gpurun_out/${TAG}_bench_c4.log 2> gpurun_out/${TAG}_stages_c4.log
timeout 900 python bench.py --steps 5 --warmup 2 --workload c5 --stages > gpurun_out/${TAG}_bench_c5.log 2> gpurun_out/${TAG}_stages_c5.log
tail -5 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log
for f in c4 c5; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_$f.log; grep -o '"parity_sample": {[^}]*}' gpurun_out/${TAG}_bench_$f.log; done
