# round 3, GPU call B: fused reverse tail A/B (C4 stages), GPU suite
cd /root/repo
TAG=${1:-r03_b}
mkdir -p gpurun_out
for v in tail keepedge staged; do
  unset AA_FUSED_TAIL AA_TAIL_KEEP_EDGE
  [ $v = staged ] && export AA_FUSED_TAIL=0
  [ $v = keepedge ] && export AA_TAIL_KEEP_EDGE=1
  timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c4_$v.log 2> gpurun_out/${TAG}_stages_c4_$v.log
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_c4_$v.log) $(grep -o '"parity_sample": {[^}]*}' gpurun_out/${TAG}_bench_c4_$v.log | head -c 200)"
  grep stage gpurun_out/${TAG}_stages_c4_$v.log
done
unset AA_FUSED_TAIL AA_TAIL_KEEP_EDGE
timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-reference --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.log
tail -5 gpurun_out/${TAG}_pytest_gpu.log
