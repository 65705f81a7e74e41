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
    gemmtest)
      timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dit.py -q -m gpu -p no:cacheprovider --timeout 600 -k "gemm or flash or fp16_mfma or forward" 2>&1 | filt | tail -15 | tee gpurun_out/r2b_gemmtest.log ;;
    gemmab)     # tile-shape A/B of the once-per-sample GEMMs: ER_GEMM_TILE=1 is the old fixed 128x128 tile, 0 = workgroups-per-CU driven choice
      for T in 1 0; do
        echo "--- ER_GEMM_TILE=$T"
        ER_GEMM_TILE=$T timeout 300 python scripts/prefill_time.py fp32 1,8 2>&1 | filt | tail -3
        ER_GEMM_TILE=$T timeout 300 python scripts/prefill_time.py fp16 1,8 2>&1 | filt | tail -3
        ER_GEMM_TILE=$T timeout 600 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1
      done 2>&1 | tee gpurun_out/r2b_gemmab.log ;;
    kernarg)    # where the kernel arguments live (host-visible vs device memory) changes what a dependent launch has to fetch first
      for V in 0 1; do
        echo "--- HIP_FORCE_DEV_KERNARG=$V"
        HIP_FORCE_DEV_KERNARG=$V TUNE_TOKENS=1000 TUNE_CONFIGS='[]' timeout 300 python scripts/tune_decode.py 2>&1 | filt | tail -1 | cut -c1-420
      done 2>&1 | tee gpurun_out/r2b_kernarg.log ;;
    gemmab2)    # forced tile shapes (2 = 64x128, 3 = 64x64) and the 2-wave / 4-wave fp32 flash attention, B = 1 prefill + DiT
      for T in 0 2 3; do
        echo "--- ER_GEMM_TILE=$T"
        ER_GEMM_TILE=$T timeout 300 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1
        ER_GEMM_TILE=$T timeout 300 python scripts/prefill_time.py fp16 1 2>&1 | filt | tail -1
        ER_GEMM_TILE=$T timeout 600 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1
      done 2>&1 | tee gpurun_out/r2b_gemmab2.log
      echo "--- ER_FLASH32_NWV=4 (auto picks 2 at B = 1)"
      ER_FLASH32_NWV=4 timeout 300 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1 | tee -a gpurun_out/r2b_gemmab2.log ;;
    nwqkv)
      TUNE_TOKENS=4000 TUNE_CONFIGS='[{"ER_NW_QKV":6},{},{"ER_NW_QKV":6}]' timeout 600 python scripts/tune_decode.py 2>&1 | filt | cut -c1-330 | tee gpurun_out/r2b_nwqkv.log ;;
    long32)     # BASELINE configs[2] (B = 32, sample, T = 16000, fp16): streaming vs split attention at long contexts, then the full-size run
      SWEEP_B=32 SWEEP_T=16000 SWEEP_PRECISION=fp16 SWEEP_CONTEXTS='[6000, 10000, 14000, 18000]' SWEEP_CONFIGS='[{"ER_ATTN_V_BATCHED": 1}]' \
        timeout 300 python scripts/attn_sweep.py 2>&1 | filt | tee gpurun_out/r2b_long32_sweep.log
      timeout 260 python scripts/bench_batch.py 32 16000 4000 fp16 sample 2>&1 | filt | tee gpurun_out/r2b_config2_full.log ;;
    *) echo "unknown section $SEC" ;;
  esac
done
echo "== done"
