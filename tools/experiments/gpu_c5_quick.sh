cd /root/repo
timeout 60 python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline --stages < /dev/null > gpurun_out/c5_quick.log 2> gpurun_out/c5_quick_stages.log
grep -o '"ms_per_step": [0-9.]*' gpurun_out/c5_quick.log
grep "gemm_" gpurun_out/c5_quick_stages.log | awk '{s+=$3} END{print "gemm total us", s, NR}'
