cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dense_contracter.py tests/test_op_seam.py tests/test_hip_contracter.py -x -q -m gpu 2>&1 | tail -5
for wl in c3 c4; do
timeout 600 python bench.py --mode train-op --workload $wl --steps 10 --warmup 3 > gpurun_out/r03_j_train_op_$wl.json 2> gpurun_out/r03_j_train_op_$wl.err; tail -2 gpurun_out/r03_j_train_op_$wl.err; cut -c1-900 gpurun_out/r03_j_train_op_$wl.json
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03_j_prof -o trainop --output-format csv -- python /root/repo/bench.py --mode train-op --workload c4 --steps 10 --warmup 3 > /dev/null 2>&1
f=$(find /root/repo/gpurun_out/r03_j_prof -name "*kernel_stats.csv" | head -1); cut -c1-150 "$f" | head -16
