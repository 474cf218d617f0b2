#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r04_v20
timeout 900 python -m pytest tests -m gpu -q -k "operator_path or c5_full" > gpurun_out/${TAG}_pytest_quick.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_quick.log
for rep in 1 2; do for mode in mfma vector; do
  if [ $mode = vector ]; then export AA_OP_ENV_VECTOR=1; else unset AA_OP_ENV_VECTOR; fi
  r=$(timeout 600 python bench.py --workload c5 --steps 5 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --sustain 0 2> gpurun_out/${TAG}_ab_c5_$mode.log | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "$mode $r | $(grep '^\[stage\] tp_op_edge_env' gpurun_out/${TAG}_ab_c5_$mode.log | awk '{printf "%s ", $3}')"
done; done | tee gpurun_out/${TAG}_ab_c5.txt
