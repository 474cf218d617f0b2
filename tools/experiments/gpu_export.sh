cd /root/repo
python -m pytest tests/test_export.py tests/test_neighbor_list.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r01_v20_pytest_export.log
tail -25 gpurun_out/r01_v20_pytest_export.log
