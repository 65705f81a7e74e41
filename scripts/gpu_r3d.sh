#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "batch or long or rows" 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r3d_tests.log
{ timeout 300 python scripts/bench_batch.py 4,5,6,7,8 1000 1000 fp32 2>&1 | grep -v amdgpu.ids
  timeout 300 python scripts/bench_batch.py 4,5,6,7,8 1000 1000 fp16 2>&1 | grep -v amdgpu.ids; } | tee gpurun_out/r03_batch_table_v3.log | grep -A1 aggregate | cut -c1-230
