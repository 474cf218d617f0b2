# end-of-round evidence: GPU tests, smoke, stress, bench lines (C4 default incl. baselines, C5, C1, C2, C3), shard runs,
# operator-seam and whole-model training steps, fp64 matrix-core probes, native-op overhead, profiles with PMC traffic
cd /root/repo
TAG=${1:-r03_final}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
timeout 600 python tools/virial_stress.py 200 > gpurun_out/${TAG}_virial_stress.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --stages > gpurun_out/${TAG}_bench_c4.log 2> gpurun_out/${TAG}_stages_c4.log
timeout 900 python bench.py --workload c5 --steps 5 --warmup 2 --stages > gpurun_out/${TAG}_bench_c5.log 2> gpurun_out/${TAG}_stages_c5.log
timeout 300 python bench.py --workload c1 --steps 200 --warmup 20 --stages --no-cpu-baseline --sustain 0 > gpurun_out/${TAG}_bench_c1.log 2> gpurun_out/${TAG}_stages_c1.log
timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c2.log 2>&1
timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --sustain 0 > gpurun_out/${TAG}_bench_c3.log 2>&1
timeout 600 python bench.py --shard-sweep 8 --steps 10 --warmup 3 > gpurun_out/${TAG}_shard_sweep8_c4.json 2> /dev/null
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 10 --warmup 3 --emulate-shard 3/8 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 > gpurun_out/${TAG}_bench_torchrun1_shard3of8.log 2>&1
timeout 900 python bench.py --mode train-op --workload c3 --steps 8 --warmup 2 > gpurun_out/${TAG}_train_op_c3.json 2>/dev/null
timeout 900 python bench.py --mode train-op --workload c4 --steps 8 --warmup 2 > gpurun_out/${TAG}_train_op_c4.json 2>/dev/null
AA_TP_GENERIC=1 timeout 900 python bench.py --mode train-op --workload c3 --steps 4 --warmup 1 > gpurun_out/${TAG}_train_op_c3_generic.json 2>/dev/null
timeout 900 python bench.py --mode train-step --workload c3 --steps 5 --warmup 2 > gpurun_out/${TAG}_train_step_c3.json 2>/dev/null
timeout 900 python bench.py --mode train-step --workload c4 --steps 3 --warmup 1 > gpurun_out/${TAG}_train_step_c4.json 2>/dev/null
for b in f64_gemm_loop mfma_f64_bcast_probe; do timeout 120 tools/ubench/$b.bin > gpurun_out/${TAG}_ubench_$b.log 2>&1; done
timeout 120 tools/ubench/mfma_f64_bcast_map.bin > /dev/null 2> gpurun_out/${TAG}_ubench_mfma_f64_bcast_map.log
# dense small systems: team form of the fused forward (default) vs the staged forward
for rc in 6.0 7.0; do for cells in 3 4 6; do
  a=$(AA_BENCH_RCUT=$rc AA_BENCH_CELLS=$cells timeout 300 python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  b=$(AA_FUSED=0 AA_BENCH_RCUT=$rc AA_BENCH_CELLS=$cells timeout 300 python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "r_max $rc cells $cells default: $a staged: $b"
done; done > gpurun_out/${TAG}_dense_small.log
timeout 600 python tools/op_overhead.py > gpurun_out/${TAG}_op_overhead.json 2>/dev/null
bash tools/profile_gpu.sh c4 ${TAG} > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_c4/summary.txt gpurun_out/${TAG}_rocprofv3_c4_summary.txt
bash tools/profile_gpu.sh c5 ${TAG} > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_c5/summary.txt gpurun_out/${TAG}_rocprofv3_c5_summary.txt
rm -rf gpurun_out/prof_${TAG}_c4/trace gpurun_out/prof_${TAG}_c4/pmc_* gpurun_out/prof_${TAG}_c5/trace gpurun_out/prof_${TAG}_c5/pmc_*
tail -3 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log; tail -1 gpurun_out/${TAG}_virial_stress.log
for f in c4 c5 c1 c2 c3; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_$f.log)"; done
grep -o '"speedup_vs_gpu_reference": [0-9.]*' gpurun_out/${TAG}_bench_c4.log gpurun_out/${TAG}_bench_c5.log
grep -o '"sustained": {[^}]*}' gpurun_out/${TAG}_bench_c4.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_torchrun1_shard3of8.log
for f in train_op_c3 train_op_c4 train_step_c3 train_step_c4; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_$f.json | head -1)"; done
cat gpurun_out/${TAG}_ubench_f64_gemm_loop.log
cat gpurun_out/${TAG}_dense_small.log
tail -2 gpurun_out/${TAG}_rocprofv3_c4_summary.txt
