cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training.py tests/test_dense_contracter.py -x -q -m gpu 2>&1 | tail -5
for wl in c3 c4; do
timeout 900 python bench.py --mode train-step --workload $wl --steps 5 --warmup 2 > gpurun_out/r03_k_train_step_$wl.json 2> gpurun_out/r03_k_train_step_$wl.err; tail -3 gpurun_out/r03_k_train_step_$wl.err; cut -c1-1500 gpurun_out/r03_k_train_step_$wl.json
done
timeout 600 python bench.py --mode train-op --workload c4 --steps 10 --warmup 3 2>/dev/null | cut -c1-700
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03_k_prof -o trainstep --output-format csv -- python /root/repo/bench.py --mode train-step --workload c3 --steps 5 --warmup 2 > /dev/null 2>&1
f=$(find /root/repo/gpurun_out/r03_k_prof -name "*kernel_stats.csv" | head -1); cut -c1-150 "$f" | head -30
