# A/B: non-temporal hints on the streamed operands of the moments kernels (variant library built with -DAA_EXP_NT)
cd /root/repo
mkdir -p gpurun_out
for v in base nt; do
  if [ $v = nt ]; then export ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_nt.so; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/nt_bench_$v.log 2> gpurun_out/nt_stages_$v.log
  timeout 300 python bench.py --workload c3 --steps 50 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/nt_bench_c3_$v.log 2> gpurun_out/nt_stages_c3_$v.log
done
for v in base nt; do echo $v; grep -h "tp_m" gpurun_out/nt_stages_$v.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/nt_bench_$v.log gpurun_out/nt_bench_c3_$v.log; done
