cd /root/repo
timeout 900 python -m pytest tests/test_fused.py -x -q -m gpu 2>&1 | tail -3
