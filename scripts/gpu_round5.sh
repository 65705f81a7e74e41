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
    bench)  timeout 900 python bench.py 2> gpurun_out/r05_bench.err | tail -1 > gpurun_out/r05_bench.json; filt < gpurun_out/r05_bench.err | tail -5; head -c 600 gpurun_out/r05_bench.json; echo ;;
    bench16) timeout 600 python bench.py --precision fp16 --steps 2 --warmup 1 2> gpurun_out/r05_bench16.err | tail -1 > gpurun_out/r05_bench_fp16.json; filt < gpurun_out/r05_bench16.err | tail -3; head -c 600 gpurun_out/r05_bench_fp16.json; echo ;;
    bench3) timeout 900 python bench.py --config 3 --steps 1 --warmup 1 2> gpurun_out/r05_bench3.err | tail -1 > gpurun_out/r05_bench_config3.json; filt < gpurun_out/r05_bench3.err | tail -3; head -c 600 gpurun_out/r05_bench_config3.json; echo ;;
    bench2) timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 2> gpurun_out/r05_bench2.err | tail -1 > gpurun_out/r05_bench_config2.json; filt < gpurun_out/r05_bench2.err | tail -3; head -c 600 gpurun_out/r05_bench_config2.json; echo ;;
    suite)  timeout 1700 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 2>&1 | filt | tail -15 | tee gpurun_out/r05_gpu_tests.log ;;
    prof|prof3|prof16)
      if [ $SEC = prof ]; then TAG=bench; ARGS="--no-fast-extra --cpu-steps 0"; elif [ $SEC = prof16 ]; then TAG=bench_fp16; ARGS="--precision fp16"; else TAG=config3; ARGS="--config 3"; fi
      rm -rf /tmp/prof_$TAG
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o r05 -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $ROOT/gpurun_out/r05_rocprof_$TAG.json 2> $ROOT/gpurun_out/r05_rocprof_$TAG.err)
      echo "rocprof rc=$?"
      find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_${TAG}_kernel_stats.csv
      python scripts/roofline_from_rocprof.py gpurun_out/r05_${TAG}_kernel_stats.csv gpurun_out/r05_rocprof_$TAG.json --tol 0.10 2>&1 | tee gpurun_out/r05_roofline_check_$TAG.log | head -30 ;;
    pmc|pmc3|pmc16)
      if [ $SEC = pmc ]; then TAG=bench; OUT=r05_pmc_hbm_summary.json; ARGS="--no-fast-extra --cpu-steps 0"; elif [ $SEC = pmc16 ]; then TAG=fp16; OUT=r05_pmc_hbm_fp16_summary.json; ARGS="--precision fp16"; else TAG=config3; OUT=r05_pmc_hbm_config3_summary.json; ARGS="--config 3"; fi
      for CNT in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${TAG}_$CNT
        (cd /tmp && ER_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$CNT -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --tokens 12 --resume-len 2000 $ARGS > /tmp/pmc_${TAG}_$CNT.json 2> $ROOT/gpurun_out/r05_pmc_${TAG}_$CNT.err)
        echo "pmc $CNT rc=$?"
      done
      # contexts of the 12 decode steps: 4051..4062 keys -> mean 4056.5
      python scripts/pmc_summary.py pmc --attn-context 4056.5 $(find /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/$OUT 2> gpurun_out/r05_pmc_${TAG}_summary.err
      python - <<PY
import json
d=json.load(open("gpurun_out/$OUT"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1].get("hbm_read_bytes_per_launch",0))[:8]:
    print(f"{k[:70]:70s} read {v.get('hbm_read_bytes_per_launch',0)/1e6:9.2f} MB  write {v.get('hbm_write_bytes_per_launch',0)/1e6:8.2f} MB")
PY
      ;;
    pmcsq)   # SQ counters of the exact prefill (fp32 LDS-DMA GEMM, fp32 flash attention) and of the DiT front-end: MFMA-busy share
      for W in prefill dit; do
        rm -rf /tmp/pmcsq_$W
        if [ $W = prefill ]; then CMD="python $ROOT/scripts/prefill_time.py fp32 1"; else CMD="python $ROOT/scripts/bench_dit.py 16 3 fp16"; fi
        (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmcsq_$W -o pmc -- $CMD > $ROOT/gpurun_out/r05_pmcsq_$W.log 2>&1)
        echo "pmcsq $W rc=$?"
        python scripts/pmc_summary.py pmc $(find /tmp/pmcsq_$W -name "*counter_collection.csv") > gpurun_out/r05_pmc_sq_$W.json 2> gpurun_out/r05_pmcsq_$W.err
        python - <<PY
import json
d=json.load(open("gpurun_out/r05_pmc_sq_$W.json"))
for k,v in d["kernels"].items():
    if not any(t in k for t in ("gemm_hh","flash_attn","gemm_f32d","gemm_f16","ln_modulate","gemm_f32_mfma")): continue
    g=v.get("GRBM_GUI_ACTIVE",{}).get("mean",0); m=v.get("SQ_VALU_MFMA_BUSY_CYCLES",{}).get("mean",0)
    bc=v.get("SQ_LDS_BANK_CONFLICT",{}).get("mean",0); ia=v.get("SQ_LDS_IDX_ACTIVE",{}).get("mean",1)
    print(f"{k[:70]:70s} disp {v['GRBM_GUI_ACTIVE']['dispatches']:5d} mfma_busy {m/(g*128) if g else 0:.3f} lds_conflict_share {bc/ia if ia else 0:.3f}")
PY
      done ;;
    ditprof) rm -rf /tmp/prof_dit
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dit -o r05 -- python $ROOT/scripts/bench_dit.py 16 10 fp16 > $ROOT/gpurun_out/r05_rocprof_dit.json 2> $ROOT/gpurun_out/r05_rocprof_dit.err)
      echo "rocprof rc=$?"; tail -1 gpurun_out/r05_rocprof_dit.json
      find /tmp/prof_dit -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_dit_fp16_kernel_stats.csv; head -12 gpurun_out/r05_dit_fp16_kernel_stats.csv | cut -c1-160 ;;
    prefill) { timeout 200 python scripts/prefill_time.py fp16 1,8 2>&1 | filt | tail -2
               timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1; } | tee gpurun_out/r05_prefill_time_final.log ;;
    abfc2)   timeout 900 python scripts/ab_decode.py fp16 2000 rw2= rw4=:ER_RW_FC2=4 rw6=:ER_RW_FC2=6 rw2b= rw4b=:ER_RW_FC2=4 rw6b=:ER_RW_FC2=6 2>&1 | filt | tee gpurun_out/r05_ab_fc2_rows.log ;;
    abom2)   timeout 900 python scripts/ab_decode.py fp16 2000 rpw1= rpw2=:ER_OM_RPW=2 rpw1b= rpw2b=:ER_OM_RPW=2 2>&1 | filt | tee gpurun_out/r05_ab_om_rpw_fp16.log ;;
    f32d)    { timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 400 -x -k "gemm_f32" 2>&1 | filt | tail -4
               for L in edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1; done; } | tee gpurun_out/r05_gemm_f32d_b128.log ;;
    pwave)   { timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 400 -x -k "outproj3 or fast_mode or fp16_cache" 2>&1 | filt | tail -4
               for L in edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L SWEEP_PRECISION=fp16 SWEEP_CONTEXTS="[2176, 3176, 3926, 4050, 4176, 4426, 5176, 5926]" timeout 300 python scripts/attn_sweep.py 2>&1 | filt | tail -12; done
               timeout 600 python scripts/ab_decode.py fp16 4000 prev=edgerunner_amd/lib_prev.so new= 2>&1 | filt; } | tee gpurun_out/r05_attn_per_wave_steps.log ;;
    batch)  { timeout 500 python scripts/bench_batch.py 1,2,4,8,12,16,32 1000 1000 fp16 2>&1 | filt
              timeout 500 python scripts/bench_batch.py 1,4,8,16,32 1000 1000 fp32 2>&1 | filt; } | tee gpurun_out/r05_batch_table.log | grep aggregate | cut -c1-200 ;;
    abdiv)   { timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "attn or T4000 or fast_mode" 2>&1 | filt | tail -4
               timeout 600 python scripts/ab_decode.py fp32 2000 prev=edgerunner_amd/lib_prev.so new= prevb=edgerunner_amd/lib_prev.so newb= 2>&1 | filt
               for L in edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_batch.py 32 1000 1000 fp16 2>&1 | filt | grep -E "aggregate" | cut -c1-230; done; } | tee gpurun_out/r05_ab_div3.log ;;
    midb)    { timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "attn or fp16 or fast_mode or batch or long" 2>&1 | filt | tail -4
               for L in edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_prev.so edgerunner_amd/libedgerunner_hip.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_batch.py 2,4,8,12 1000 1000 fp16 2>&1 | filt | grep -E "aggregate" | cut -c1-200; done; } | tee gpurun_out/r05_ab_midbatch_pair.log ;;
    abfc232) timeout 900 python scripts/ab_decode.py fp32 2000 rw2= rw3=:ER_RW_FC2_F32=3 rw4=:ER_RW_FC2_F32=4 rw2b= rw3b=:ER_RW_FC2_F32=3 rw4b=:ER_RW_FC2_F32=4 2>&1 | filt | tee gpurun_out/r05_ab_fc2_rows_fp32.log ;;
    dittile) { for E in "" "ER_GEMM256_MIN_TILES=192" "ER_GEMM_HH_TILE=1" "ER_GEMM_HH_TILE=2" "ER_GEMM_HH_TILE=3" "" "ER_GEMM256_MIN_TILES=192" "ER_GEMM256_MIN_TILES=64"; do echo "== $E"; env $E timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1; done; } | tee gpurun_out/r05_dit_tile_rules.log ;;
    abvf)    { for L in edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_vf.so edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_vf.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1; ER_LIB_PATH=$ROOT/$L timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1; ER_LIB_PATH=$ROOT/$L timeout 200 python scripts/prefill_time.py fp16 1,8 2>&1 | filt | tail -2; done; } | tee gpurun_out/r05_ab_mfma_vgpr_form.log ;;
    abilp)   { timeout 600 python scripts/ab_decode.py fp32 2000 base= ilp=edgerunner_amd/lib_ilp.so 2>&1 | filt
               timeout 600 python scripts/ab_decode.py fp16 2000 base= ilp=edgerunner_amd/lib_ilp.so 2>&1 | filt
               for L in edgerunner_amd/libedgerunner_hip.so edgerunner_amd/lib_ilp.so; do echo "== $L"; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1; ER_LIB_PATH=$ROOT/$L timeout 300 python scripts/bench_batch.py 32 600 1000 fp16 2>&1 | filt | grep aggregate | cut -c1-200; ER_LIB_PATH=$ROOT/$L timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1; done; } | tee gpurun_out/r05_ab_sched_max_ilp.log ;;
    abfa8)   { timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dit.py -q -m gpu -p no:cacheprovider --timeout 600 -x -k "flash_attn_hh or dit or DiT or forward or sampler or latents" 2>&1 | filt | tail -4
               for W in 4 8 4 8; do echo "== ER_FA_HH_WAVES=$W"; ER_FA_HH_WAVES=$W timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1; done; } | tee gpurun_out/r05_ab_fa_hh_waves.log ;;
    ksplit)  { ER_TEST_CANDIDATES=1 timeout 600 python -m pytest tests -q -m gpu -k candidate -p no:cacheprovider --timeout 400 2>&1 | filt | tail -6
               for K in 0 1 0 1; do echo "ER_FLASH32_KSPLIT=$K"; ER_FLASH32_KSPLIT=$K timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1; done; } | tee gpurun_out/r05_ksplit.log ;;
  esac
done
