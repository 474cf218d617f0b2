cd /root/repo
timeout 900 python -m pytest tests/test_hip_full_size.py tests/test_hip_model.py -x -q -m gpu -k "c5 or op or full" 2>&1 | tail -2
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_old.so; else unset ALLEGRO_AMD_LIBRARY; fi
  r=$(timeout 600 python bench.py --workload c5 --steps 5 --warmup 2 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 2> gpurun_out/ab5_$lib.log | grep -o '"ms_per_step": [0-9.]*')
  echo "$lib $r tp_op_fwd $(grep 'stage. tp_op_fwd' gpurun_out/ab5_$lib.log | awk '{printf "%s ", $3}') tp_op_bwd $(grep 'stage. tp_op_bwd' gpurun_out/ab5_$lib.log | awk '{printf "%s ", $3}')"
done; done
