# small systems (launch-latency regime): staged vs fused forward (AA_FUSED=2) over box sizes (Si n^3 cells, 8 n^3 atoms)
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/small_sweep.log
for n in 2 3 4 5 6 8 11; do
  for f in 0 1 2; do
    ms=$(AA_BENCH_CELLS=$n AA_FUSED=$f timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "cells=$n atoms=$((8*n*n*n)) AA_FUSED=$f $ms" >> gpurun_out/small_sweep.log
  done
done
cat gpurun_out/small_sweep.log
