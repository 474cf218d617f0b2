# automatic fused-forward selection: GPU tests of the fused paths, size sweep default vs AA_FUSED=0
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused.py -m gpu -q 2>&1 | tail -5 > gpurun_out/auto_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/auto_smoke.log 2>&1
: > gpurun_out/r02_v15_small_sweep.log
for n in 2 3 4 5 6 8; do
  for f in auto 0; do
    if [ $f = auto ]; then unset AA_FUSED; else export AA_FUSED=0; fi
    ms=$(AA_BENCH_CELLS=$n timeout 300 python bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "cells=$n atoms=$((8*n*n*n)) AA_FUSED=$f $ms" >> gpurun_out/r02_v15_small_sweep.log
  done
done
unset AA_FUSED
tail -3 gpurun_out/auto_pytest.log; tail -1 gpurun_out/auto_smoke.log; cat gpurun_out/r02_v15_small_sweep.log
