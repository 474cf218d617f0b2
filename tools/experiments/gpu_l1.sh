cd /root/repo
export AA_BENCH_LMAX=1
for f in 0 2 1; do
  if [ $f = 0 ]; then unset AA_FUSED; else export AA_FUSED=$f; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/l1_$f.log 2> gpurun_out/l1_stages_$f.log
  echo "== AA_FUSED=$f"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/l1_$f.log; grep stage gpurun_out/l1_stages_$f.log | head -7
done
