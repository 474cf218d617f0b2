cd /root/repo
TAG=${1:-r03_g}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_dense_contracter.py tests/test_hip_contracter.py tests/test_op_seam.py -m gpu -q 2>&1 | tail -4
timeout 900 python bench.py --mode train-op --workload c3 --steps 8 --warmup 2 > gpurun_out/${TAG}_train_op_c3.json 2> gpurun_out/${TAG}_train_op_c3.err; cat gpurun_out/${TAG}_train_op_c3.json
AA_TP_GENERIC=1 timeout 900 python bench.py --mode train-op --workload c3 --steps 8 --warmup 2 > gpurun_out/${TAG}_train_op_c3_generic.json 2>/dev/null; cat gpurun_out/${TAG}_train_op_c3_generic.json
