cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dense_contracter.py tests/test_op_seam.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --mode train-op --steps 20 --warmup 5 > gpurun_out/r03_i_train_op_c3.json 2> gpurun_out/r03_i_train_op_c3.err; tail -3 gpurun_out/r03_i_train_op_c3.err; cat gpurun_out/r03_i_train_op_c3.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r03_i_prof -o trainop --output-format csv -- python /root/repo/bench.py --mode train-op --steps 10 --warmup 3 > /dev/null 2>&1
f=$(find /root/repo/gpurun_out/r03_i_prof -name "*kernel_stats.csv" | head -1); cut -c1-150 "$f" | head -24
