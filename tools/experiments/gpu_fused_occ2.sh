cd /root/repo
TAG=${1:-r02_t}
for mode in hold recompute; do
  if [ $mode = recompute ]; then export AA_FUSED_RECOMPUTE=1; else unset AA_FUSED_RECOMPUTE; fi
  ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_occ2.so timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_occ2_c4_$mode.log 2> gpurun_out/${TAG}_occ2_stages_c4_$mode.log
  echo "== $mode"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_occ2_c4_$mode.log; grep "fused" gpurun_out/${TAG}_occ2_stages_c4_$mode.log
done
