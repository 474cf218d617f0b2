cd /root/repo
export AA_BUILD_EXPERIMENTAL=1
for wl in "c1 0" "c2 0" "c2 3" "c2 4" "c2 6" "c2 8"; do set -- $wl; w=$1; cells=$2
  for t in 0 1; do
    if [ $cells = 0 ]; then unset AA_BENCH_CELLS; else export AA_BENCH_CELLS=$cells; fi
    r=$(AA_FUSED_TAIL=$t timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "workload $w cells $cells fused_tail=$t $r"
  done
done
unset AA_BENCH_CELLS
AA_FUSED_TAIL=1 timeout 300 python bench.py --workload c2 --steps 50 --warmup 10 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 2>&1 >/dev/null | grep stage | cut -c1-70
