#!/bin/bash
# the whole GPU suite, then the default bench (one gpurun call; ~8 min)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r03_gpu_tests.log
bash scripts/gpu_round3.sh bench
