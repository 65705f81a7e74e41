#!/bin/bash
# Round-3 follow-up measurements (one gpurun call): V^T written by the q/k/v GEMM epilogue + hoisted modulation loads (DiT), register
# prefetch in the split-fp16 prefix attention (fast-mode prefill)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
filt() { grep -v amdgpu.ids; }
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 300 -k "gemm_hh or f16s or flash" 2>&1 | filt | tail -6
timeout 600 python -m pytest tests/test_gpu_dit.py -q -m gpu -s -p no:cacheprovider --timeout 400 2>&1 | filt | tail -12
timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -p no:cacheprovider --timeout 400 -k "fast_mode or f16s or fp16" 2>&1 | filt | tail -12
echo "--- ER_PREFILL_ATTN_F16S=1 (prefetch)"
ER_PREFILL_ATTN_F16S=1 timeout 300 python scripts/prefill_time.py fp16 1,2,8,32 2>&1 | filt | tail -4
echo "--- ER_PREFILL_ATTN_F16S=1 ER_F16S_PREFETCH=0"
ER_PREFILL_ATTN_F16S=1 ER_F16S_PREFETCH=0 timeout 300 python scripts/prefill_time.py fp16 1,8 2>&1 | filt | tail -2
echo "--- default"
timeout 300 python scripts/prefill_time.py fp16 1 2>&1 | filt | tail -1
} 2>&1 | tee gpurun_out/r3b.log
