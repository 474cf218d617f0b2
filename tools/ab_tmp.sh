#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_round.sh ab r04_v27 c4
