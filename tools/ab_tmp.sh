#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r04_v21
timeout 900 python -m pytest tests -m gpu -q -k "operator_path or c5_full" > gpurun_out/${TAG}_pytest_quick.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_quick.log
for rep in 1 2; do
  r=$(timeout 600 python bench.py --workload c5 --steps 5 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --sustain 0 2> gpurun_out/${TAG}_stages_c5.log | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "$r | $(grep '^\[stage\] edge_backward' gpurun_out/${TAG}_stages_c5.log | awk '{printf "%s ", $3}')"
done | tee gpurun_out/${TAG}_c5.txt
