cd /root/repo
TAG=${1:-r02_t}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -100 > gpurun_out/${TAG}_pytest_gpu.log
tail -8 gpurun_out/${TAG}_pytest_gpu.log
