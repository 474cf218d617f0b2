cd /root/repo
TAG=${1:-r03_e}
mkdir -p gpurun_out
timeout 600 python tools/op_overhead.py > gpurun_out/${TAG}_op_overhead.json 2> gpurun_out/${TAG}_op_overhead.err; cat gpurun_out/${TAG}_op_overhead.json
timeout 900 python bench.py --mode train-op --workload c3 --steps 8 --warmup 2 > gpurun_out/${TAG}_train_op_c3.json 2> gpurun_out/${TAG}_train_op_c3.err; cat gpurun_out/${TAG}_train_op_c3.json; tail -3 gpurun_out/${TAG}_train_op_c3.err
timeout 600 python -m pytest tests/test_op_seam.py tests/test_hip_contracter.py tests/test_export.py tests/test_pair_allegro.py -m gpu -q 2>&1 | tail -3
