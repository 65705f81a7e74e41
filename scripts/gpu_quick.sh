#!/bin/bash
# quick GPU validation: unit + parity tests, optional extra command
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 300 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -p no:cacheprovider --timeout 900 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/test_parity.log
if [ $# -gt 0 ]; then echo "== extra: $*"; timeout 900 "$@" 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/extra.log; fi
