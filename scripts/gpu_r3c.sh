#!/bin/bash
# kernel + parity suites, then the batch table (one gpurun call)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 900 -x 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r3c_tests.log
{ timeout 400 python scripts/bench_batch.py 1,2,3,4,5,8,12,16,32 1000 1000 fp32 2>&1 | grep -v amdgpu.ids
  timeout 400 python scripts/bench_batch.py 1,2,3,4,5,8,12,16,32 1000 1000 fp16 2>&1 | grep -v amdgpu.ids; } | tee gpurun_out/r03_batch_table_v2.log | grep aggregate | cut -c1-200
