#!/bin/bash
# Samples clocks and power (rocm-smi) while the C4 bench loop runs: is the step clock- or power-limited?
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-r04}
(timeout 300 python bench.py --workload c4 --steps 6000 --warmup 50 --no-cpu-baseline --no-gpu-reference --no-secondary --no-profile --sustain 0 > gpurun_out/${TAG}_power_bench.log 2>&1) &
BP=$!
sleep 30   # import + setup
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 3
done | tee gpurun_out/${TAG}_power_samples.txt
wait $BP
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_power_bench.log | head -1
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo " (idle)"
