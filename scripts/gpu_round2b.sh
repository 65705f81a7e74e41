#!/bin/bash
# Round-2 exploration call (second half of the round): decode version 3 (balanced attention chunks + merge fused into
# out_proj), 6-wave qkv workgroups, heads-fastest grid of the batched attention kernel.
#   gpurun --timeout 900 -- 'bash scripts/gpu_round2b.sh unit explore parity'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
filt() { grep -v amdgpu.ids; }
for SEC in "$@"; do
  echo "=================== $SEC"
  case "$SEC" in
    unit)
      timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 300 -k "${UNIT_K:-outproj3}" 2>&1 | filt | tail -15 | tee gpurun_out/r2b_unit.log ;;
    explore)
      timeout 900 python scripts/explore_r2b.py ${EXPLORE_SECTIONS:-tune32 tune16 sweep1 sweep32} 2>&1 | filt | tee gpurun_out/r2b_explore.log ;;
    parity)
      timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -p no:cacheprovider --timeout 600 \
        -k "${PARITY_K:-decode_v3 or (full_size and 3)}" 2>&1 | filt | tail -40 | tee gpurun_out/r2b_parity.log ;;
    *) echo "unknown section $SEC" ;;
  esac
done
echo "== done"
