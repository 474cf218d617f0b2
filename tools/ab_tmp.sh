#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r04_v17
timeout 900 python -m pytest tests -m gpu -q -k "operator_path or c5_full or gemm" > gpurun_out/${TAG}_pytest_quick.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_quick.log
for rep in 1 2; do for mode in proj noproj; do
  if [ $mode = noproj ]; then export AA_OP_PROJ=0; else unset AA_OP_PROJ; fi
  r=$(timeout 600 python bench.py --workload c5 --steps 5 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --sustain 0 2> gpurun_out/${TAG}_ab_c5_$mode.log | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "$mode $r"
done; done | tee gpurun_out/${TAG}_ab_c5.txt
unset AA_OP_PROJ
grep '^\[stage\]' gpurun_out/${TAG}_ab_c5_proj.log
grep '^\[stage\] tp_op' gpurun_out/${TAG}_ab_c5_noproj.log
