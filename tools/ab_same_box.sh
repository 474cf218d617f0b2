#!/bin/bash
# Same-box A/B of two builds of the device library (box-to-box variation on the gpurun pool is +-5 %, more than most kernel
# changes are worth): alternates `old` and `new` three times within ONE call and prints the step and the per-launch times.
#   1. build the baseline:   git stash; python -m allegro_amd.build; cp allegro_amd/liballegro_amd.so allegro_amd/liballegro_amd_old.so; git stash pop
#   2. build the candidate:  python -m allegro_amd.build
#   3. gpurun -- 'bash tools/ab_same_box.sh c4'          (workload: c4 | c5 | c3 ...)
cd "$(dirname "$0")/.."
WL=${1:-c4}
mkdir -p gpurun_out
for rep in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export ALLEGRO_AMD_LIBRARY=$PWD/allegro_amd/liballegro_amd_old.so; else unset ALLEGRO_AMD_LIBRARY; fi
    r=$(timeout 600 python bench.py --workload $WL --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 2> gpurun_out/ab_$lib.log | grep -o '"ms_per_step": [0-9.]*')
    echo "$lib $r | $(grep '^\[stage\]' gpurun_out/ab_$lib.log | awk '{printf "%s %s  ", $2, $3}')"
  done
done
