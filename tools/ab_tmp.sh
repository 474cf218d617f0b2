#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r04_v22
timeout 1200 python -m pytest tests/test_fused.py tests/test_hip_model.py -m gpu -q -x > gpurun_out/${TAG}_pytest_quick.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_quick.log
for rep in 1 2; do for mode in fused staged_folded staged_unfolded; do
  unset AA_FUSED AA_STAGED_NOFOLD
  if [ $mode != fused ]; then export AA_FUSED=0; fi
  if [ $mode = staged_unfolded ]; then export AA_STAGED_NOFOLD=1; fi
  r=$(timeout 600 python bench.py --workload c4 --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --sustain 0 2> gpurun_out/${TAG}_ab_c4_$mode.log | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "$mode $r | $(grep '^\[stage\]' gpurun_out/${TAG}_ab_c4_$mode.log | awk '{printf "%s %s  ", $2, $3}')"
done; done | tee gpurun_out/${TAG}_ab_c4.txt
