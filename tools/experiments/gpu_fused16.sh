cd /root/repo
TAG=${1:-r02_f16}
export AA_FUSED=2
for mode in hold recompute; do
  if [ $mode = recompute ]; then export AA_FUSED_RECOMPUTE=1; else unset AA_FUSED_RECOMPUTE; fi
  ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_timing.so timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 > gpurun_out/${TAG}_timing_c4_$mode.log 2>&1
  echo "== $mode"; grep "timing" gpurun_out/${TAG}_timing_c4_$mode.log
  timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c4_$mode.log 2> gpurun_out/${TAG}_stages_c4_$mode.log
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_c4_$mode.log; grep fused_fwd gpurun_out/${TAG}_stages_c4_$mode.log
done
unset AA_FUSED_RECOMPUTE
unset AA_FUSED
timeout 900 python -m pytest tests/test_fused.py -m gpu -q 2>&1 | tail -3
AA_FUSED=2 timeout 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_model.py -m gpu -q -k "c4 or golden" 2>&1 | tail -3
