cd /root/repo
TAG=${1:-r01_final}
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
python bench.py --steps 20 --warmup 5 --stages > gpurun_out/${TAG}_bench_c4.log 2> gpurun_out/${TAG}_stages_c4.log
python bench.py --steps 50 --warmup 10 --workload c3 --no-cpu-baseline > gpurun_out/${TAG}_bench_c3.log 2>&1
python bench.py --steps 5 --warmup 2 --workload c5 --no-cpu-baseline > gpurun_out/${TAG}_bench_c5.log 2>&1
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_bench_torchrun1.log 2>&1
bash tools/profile_gpu.sh c4 ${TAG} > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_c4/summary.txt gpurun_out/${TAG}_rocprofv3_c4_summary.txt
rm -rf gpurun_out/prof_${TAG}_c4/trace gpurun_out/prof_${TAG}_c4/pmc_*
tail -2 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log
for f in c4 c3 c5 torchrun1; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_$f.log; done
head -12 gpurun_out/${TAG}_rocprofv3_c4_summary.txt
