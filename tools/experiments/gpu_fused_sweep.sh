# staged vs fused forward (32- / 16-edge tiles) over box sizes, after the anchored scalar accumulation
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r02_v23_fused_sweep.log
for n in 2 5 8 11 23; do
  for f in 0 1 2; do
    ms=$(AA_BENCH_CELLS=$n AA_FUSED=$f timeout 300 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "cells=$n atoms=$((8*n*n*n)) AA_FUSED=$f $ms" >> gpurun_out/r02_v23_fused_sweep.log
  done
done
cat gpurun_out/r02_v23_fused_sweep.log
