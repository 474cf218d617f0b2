cd /root/repo
TAG=${1:-r02_c5}
timeout 900 python bench.py --workload c5 --steps 5 --warmup 2 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c5.log 2> gpurun_out/${TAG}_stages_c5.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_c5.log
grep stage gpurun_out/${TAG}_stages_c5.log | awk '{n[$2]++; t[$2]+=$3} END {for (k in t) printf "%-28s %3d %10.1f us\n", k, n[k], t[k]}' | sort -k3 -n -r | head -12
timeout 600 python -m pytest tests/test_hip_full_size.py tests/test_op_seam.py -m gpu -q 2>&1 | tail -3
