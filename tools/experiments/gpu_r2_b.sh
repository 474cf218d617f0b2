# round 2, pass B: full GPU test suite, shard sweep, torchrun 1-rank, C5 + C4 rocprofv3/PMC profiles
cd /root/repo
TAG=${1:-r02_v7}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.log
timeout 600 python bench.py --shard-sweep 8 --steps 10 --warmup 3 > gpurun_out/${TAG}_shard_sweep8_c4.json 2> gpurun_out/${TAG}_shard_sweep8_c4.err
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 > gpurun_out/${TAG}_bench_torchrun1.log 2>&1
timeout 600 python bench.py --emulate-shard 3/8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_shard3of8.log 2>&1
bash tools/profile_gpu.sh c5 ${TAG} > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_c5/summary.txt gpurun_out/${TAG}_rocprofv3_c5_summary.txt
bash tools/profile_gpu.sh c4 ${TAG} > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_c4/summary.txt gpurun_out/${TAG}_rocprofv3_c4_summary.txt
rm -rf gpurun_out/prof_${TAG}_c4/trace gpurun_out/prof_${TAG}_c4/pmc_* gpurun_out/prof_${TAG}_c5/trace gpurun_out/prof_${TAG}_c5/pmc_*
tail -4 gpurun_out/${TAG}_pytest_gpu.log
python -c "import json;d=json.load(open('gpurun_out/${TAG}_shard_sweep8_c4.json'));print({k:d[k] for k in ('max_ms','mean_ms','imbalance_max_over_mean')})"
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_torchrun1.log gpurun_out/${TAG}_bench_shard3of8.log
head -14 gpurun_out/${TAG}_rocprofv3_c5_summary.txt
