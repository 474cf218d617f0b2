cd /root/repo
TAG=${1:-r03_d}
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c4.log 2> gpurun_out/${TAG}_stages_c4.log
echo "== $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_c4.log)"
grep "stage" gpurun_out/${TAG}_stages_c4.log
for w in c3 c2 c1; do echo "$w $(timeout 300 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done
