#!/bin/bash
# Round-5 measurement sections: usage  bash scripts/gpu_round5.sh <section> [...]   (outputs under gpurun_out/, copied to profiles/ by hand)
#   pfprobe  scripts/probes/prefetch_stream_probe (run-ahead prefetch walker next to the launch chain)
#   loadpat  scripts/probes/attn_load_pattern_probe (192-byte fp16 key rows: thirds vs contiguous wave loads)
#   ksplit   staged key-range split of the exact prefill attention: its gated tests + prefill A/B
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
filt() { grep -v amdgpu.ids; }
for SEC in "$@"; do
  case $SEC in
    pfprobe) timeout 240 scripts/probes/prefetch_stream_probe 20 2>&1 | tee gpurun_out/r05_prefetch_stream_probe.log ;;
    loadpat) timeout 120 scripts/probes/attn_load_pattern_probe 2>&1 | tee gpurun_out/r05_attn_load_pattern_probe.log ;;
    attn)    timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "attn or key_range or fast_mode or fp16 or batch" 2>&1 | filt | tail -8 | tee gpurun_out/r05_attn_tests.log ;;
    ab16)    timeout 600 python scripts/ab_decode.py fp16 2000 r4=edgerunner_amd/lib_r4_baseline.so new= r4b=edgerunner_amd/lib_r4_baseline.so newb= 2>&1 | filt | tee gpurun_out/r05_ab_fp16.log ;;
    ab32)    timeout 600 python scripts/ab_decode.py fp32 2000 r4=edgerunner_amd/lib_r4_baseline.so new= 2>&1 | filt | tee gpurun_out/r05_ab_fp32.log ;;
    abb)     { for L in edgerunner_amd/lib_r4_baseline.so edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_r4_baseline.so edgerunner_amd/libedgerunner_hip.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_batch.py 16,32 1000 1000 fp16 2>&1 | filt | grep aggregate | cut -c1-220; done; } | tee gpurun_out/r05_ab_batch.log ;;
    abom)    { timeout 600 python scripts/ab_decode.py fp32 2000 new= omw=edgerunner_amd/lib_om_wfirst.so newb= omwb=edgerunner_amd/lib_om_wfirst.so; timeout 400 python scripts/ab_decode.py fp16 2000 new= omw=edgerunner_amd/lib_om_wfirst.so; } 2>&1 | filt | tee gpurun_out/r05_ab_om_wfirst.log ;;
    abgm)    { for L in edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_gm_wfirst.so edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_gm_wfirst.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_batch.py 16,32 600 1000 fp16 2>&1 | filt | grep -E "aggregate|per-kind" | cut -c1-260; done; } | tee gpurun_out/r05_ab_gm_wfirst.log ;;
    gmtl)    { for K in 0 1 2; do for B in 16 32; do timeout 60 scripts/probes/gemv_mfma_timeline_probe $B $K; timeout 60 scripts/probes/gemv_mfma_timeline_probe_stamped $B $K; done; done; } 2>&1 | tee gpurun_out/r05_gemv_mfma_timeline.log ;;
    gemm)    { timeout 300 scripts/probes/gemm_hh_probe 2>&1; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 400 -x -k "gemm_hh" 2>&1 | filt | tail -5; } | tee gpurun_out/r05_gemm_hh256_probe.log ;;
    dit)     { timeout 900 python -m pytest tests/test_gpu_dit.py -q -m gpu -s -p no:cacheprovider --timeout 600 2>&1 | filt | tail -12
               for G in 0 1 0 1; do echo "ER_GEMM256=$G"; ER_GEMM256=$G timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1; done; } | tee gpurun_out/r05_dit.log ;;
    abx32)   { for L in edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_batch.py 16,32 600 1000 fp16 2>&1 | filt | grep -E "aggregate|per-kind" | cut -c1-260; done; } | tee gpurun_out/r05_ab_mfma_k32.log ;;
    btests)  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "batch or fast_mode or fp16 or mfma or tiled or xt" 2>&1 | filt | tail -8 | tee gpurun_out/r05_batch_tests.log ;;
    ksplit)  { ER_TEST_CANDIDATES=1 timeout 600 python -m pytest tests -q -m gpu -k candidate -p no:cacheprovider --timeout 400 2>&1 | filt | tail -6
               for K in 0 1 0 1; do echo "ER_FLASH32_KSPLIT=$K"; ER_FLASH32_KSPLIT=$K timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1; done; } | tee gpurun_out/r05_ksplit.log ;;
  esac
done
