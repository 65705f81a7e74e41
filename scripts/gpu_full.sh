#!/bin/bash
# the whole GPU suite, then prefill timings (one gpurun call; ~7 min); `bash scripts/gpu_round3.sh bench` for the bench line
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r03_gpu_tests.log
{ timeout 200 python scripts/prefill_time.py fp16 1,8 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | grep -v amdgpu.ids | tail -1; } | tee gpurun_out/r03_prefill_time_final.log
