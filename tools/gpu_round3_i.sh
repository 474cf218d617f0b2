cd /root/repo
timeout 900 python -m pytest tests/test_fused.py tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -3
for rc in 6.0 7.0; do for cells in 3 4 6 8; do
  a=$(AA_BENCH_RCUT=$rc AA_BENCH_CELLS=$cells timeout 300 python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  b=$(AA_FUSED=0 AA_BENCH_RCUT=$rc AA_BENCH_CELLS=$cells timeout 300 python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "r_max $rc cells $cells default: $a staged: $b"
done; done
