cd /root/repo
TAG=${1:-r03_f}
mkdir -p gpurun_out
for v in auto staged; do
  unset AA_FUSED; [ $v = staged ] && export AA_FUSED=0
  timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c4_$v.log 2> gpurun_out/${TAG}_stages_c4_$v.log
  echo "== $v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_c4_$v.log)"
  grep "stage" gpurun_out/${TAG}_stages_c4_$v.log
done
unset AA_FUSED
for w in c3 c2 c1; do echo "$w $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
timeout 900 python -m pytest tests/test_hip_model.py tests/test_gemm_accuracy.py tests/test_hip_full_size.py -m gpu -q -x 2>&1 | tail -2
