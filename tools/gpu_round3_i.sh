cd /root/repo
for rep in 1 2 3; do
for lib in old new; do
  if [ $lib = old ]; then export ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_old.so; else unset ALLEGRO_AMD_LIBRARY; fi
  r=$(timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 2> gpurun_out/ab_$lib.log | grep -o '"ms_per_step": [0-9.]*')
  echo "$lib $r fused $(grep 'stage. fused' gpurun_out/ab_$lib.log | awk '{print $3}') B3 $(grep 'gc_64x64_64x64_128' gpurun_out/ab_$lib.log | awk '{print $3}') B2 $(grep 'gc_64x64_64x128' gpurun_out/ab_$lib.log | awk '{print $3}') B1 $(grep 'gc_256' gpurun_out/ab_$lib.log | awk '{print $3}') tpf $(grep 'tp_mom_bwd_first' gpurun_out/ab_$lib.log | awk '{print $3}')"
done; done
