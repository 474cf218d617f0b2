#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r04_v25
AA_POISON=1 timeout 1500 python -m pytest tests/test_hip_model.py tests/test_fused.py tests/test_hip_full_size.py -m gpu -q > gpurun_out/${TAG}_pytest_poison.log 2>&1; tail -5 gpurun_out/${TAG}_pytest_poison.log
