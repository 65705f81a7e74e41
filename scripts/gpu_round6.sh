#!/bin/bash
# Round-6 measurement sections: usage  bash scripts/gpu_round6.sh <section> [...]   (outputs under gpurun_out/, copied to profiles/ by hand)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
filt() { grep -v amdgpu.ids; }
for SEC in "$@"; do
  case $SEC in
    dittrace) # per-dispatch kernel trace of the DiT front-end, grouped by (kernel, grid): which GEMM shape costs what
      rm -rf /tmp/trace_dit
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_dit -o r06 -- python $ROOT/scripts/bench_dit.py 16 6 fp16 > $ROOT/gpurun_out/r06_dittrace.json 2> $ROOT/gpurun_out/r06_dittrace.err)
      echo "rocprof rc=$?"; tail -1 gpurun_out/r06_dittrace.json
      python scripts/trace_by_shape.py $(find /tmp/trace_dit -name "*kernel_trace.csv" | head -1) gemm_hh flash_attn_hh ln_modulate > gpurun_out/r06_dit_trace_by_shape.json
      python - <<PY
import json
d=json.load(open("gpurun_out/r06_dit_trace_by_shape.json"))
for r in d["groups"][:14]: print(f"{r['kernel'][:60]:60s} grid {r['grid_threads']:>14s} wg {r['workgroup']:>4s} n {r['launches']:5d} mean {r['mean_us']:8.2f} med {r['median_us']:8.2f} min {r['min_us']:8.2f} us")
PY
      ;;
    dittrace_ab) # per-kernel in-situ comparison of two library builds on ONE box: LIBS="a.so b.so ..."
      for L in ${LIBS:-edgerunner_amd/lib_r5_baseline.so edgerunner_amd/libedgerunner_hip.so}; do
        T=$(basename $L .so); rm -rf /tmp/trace_$T
        (cd /tmp && ER_LIB_PATH=$ROOT/$L timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$T -o r06 -- python $ROOT/scripts/bench_dit.py 16 6 fp16 > /dev/null 2> /dev/null)
        python scripts/trace_by_shape.py $(find /tmp/trace_$T -name "*kernel_trace.csv" | head -1) gemm_hh flash_attn_hh ln_modulate > gpurun_out/r06_dit_trace_$T.json
        echo "== $L"
        python - <<PY
import json
d=json.load(open("gpurun_out/r06_dit_trace_$T.json"))
for r in d["groups"][:6]: print(f"{r['kernel'][:56]:56s} grid {r['grid_threads']:>12s} n {r['launches']:5d} mean {r['mean_us']:8.2f} med {r['median_us']:8.2f} min {r['min_us']:8.2f} us")
PY
      done ;;
    gemm)    timeout 300 scripts/probes/gemm_hh_probe 2>&1 | tee gpurun_out/r06_gemm_hh_probe.log ;;
    gemmt)   timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 400 -x -k "gemm_hh" 2>&1 | filt | tail -8 | tee gpurun_out/r06_gemm_hh_tests.log ;;
    dit)     { timeout 900 python -m pytest tests/test_gpu_dit.py -q -m gpu -s -p no:cacheprovider --timeout 600 2>&1 | filt | tail -12
               for G in "${DIT_AB:-ER_GEMM_STREAM=0 ER_GEMM_STREAM=1 ER_GEMM_STREAM=0 ER_GEMM_STREAM=1}"; do for V in $G; do echo "$V"; env $V timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1; done; done; } | tee gpurun_out/r06_dit.log ;;
    bench)  timeout 900 python bench.py 2> gpurun_out/r06_bench.err | tail -1 > gpurun_out/r06_bench.json; filt < gpurun_out/r06_bench.err | tail -12; head -c 400 gpurun_out/r06_bench.json; echo
            python - <<PY
import json
d=json.load(open("gpurun_out/r06_bench.json"))
print("value", d["value"], "detok", d.get("detokenise_ms"))
for k in ("config3_shard_exact_fp32","config3_shard_fp16"):
    v=d.get(k,{}); print(k, {q:v.get(q) for q in ("value","decode_only_tokens_per_s","ms_per_step","error")}, (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("whole_step"), v.get("greedy_ids_vs_exact_fp32"))
print("fast", d.get("fast_mode_fp16",{}).get("decode_only_tokens_per_s"), "dit", d.get("dit_front_end_fp16",{}).get("ms_per_cfg_forward"), d.get("dit_front_end_fp16",{}).get("roofline",{}).get("frac"))
PY
            ;;
    bench16) timeout 600 python bench.py --precision fp16 --steps 2 --warmup 1 2> gpurun_out/r06_bench16.err | tail -1 > gpurun_out/r06_bench_fp16.json; filt < gpurun_out/r06_bench16.err | tail -3; head -c 400 gpurun_out/r06_bench_fp16.json; echo ;;
    bench3) timeout 900 python bench.py --config 3 --steps 1 --warmup 1 2> gpurun_out/r06_bench3.err | tail -1 > gpurun_out/r06_bench_config3.json; filt < gpurun_out/r06_bench3.err | tail -3; head -c 400 gpurun_out/r06_bench_config3.json; echo ;;
    bench3x) timeout 900 python bench.py --config 3 --precision fp32 --steps 1 --warmup 1 2> gpurun_out/r06_bench3x.err | tail -1 > gpurun_out/r06_bench_config3_exact.json; filt < gpurun_out/r06_bench3x.err | tail -3; head -c 400 gpurun_out/r06_bench_config3_exact.json; echo ;;
    bench2) timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 2> gpurun_out/r06_bench2.err | tail -1 > gpurun_out/r06_bench_config2.json; filt < gpurun_out/r06_bench2.err | tail -3; head -c 400 gpurun_out/r06_bench_config2.json; echo ;;
    prof|prof3|prof16)
      if [ $SEC = prof ]; then TAG=bench; ARGS="--no-fast-extra --cpu-steps 0"; elif [ $SEC = prof16 ]; then TAG=bench_fp16; ARGS="--precision fp16"; else TAG=config3; ARGS="--config 3"; fi
      rm -rf /tmp/prof_$TAG
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o r06 -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $ROOT/gpurun_out/r06_rocprof_$TAG.json 2> $ROOT/gpurun_out/r06_rocprof_$TAG.err)
      echo "rocprof rc=$?"
      find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_${TAG}_kernel_stats.csv
      python scripts/roofline_from_rocprof.py gpurun_out/r06_${TAG}_kernel_stats.csv gpurun_out/r06_rocprof_$TAG.json --tol 0.10 2>&1 | tee gpurun_out/r06_roofline_check_$TAG.log | head -30 ;;
    pmc|pmc3|pmc16|pmc3x)
      if [ $SEC = pmc ]; then TAG=bench; OUT=r06_pmc_hbm_summary.json; ARGS="--no-fast-extra --cpu-steps 0"; elif [ $SEC = pmc16 ]; then TAG=fp16; OUT=r06_pmc_hbm_fp16_summary.json; ARGS="--precision fp16"; elif [ $SEC = pmc3 ]; then TAG=config3; OUT=r06_pmc_hbm_config3_summary.json; ARGS="--config 3"; else TAG=config3x; OUT=r06_pmc_hbm_config3_exact_summary.json; ARGS="--config 3 --precision fp32"; fi
      for CNT in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${TAG}_$CNT
        (cd /tmp && ER_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$CNT -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --tokens 12 --resume-len 2000 $ARGS > /tmp/pmc_${TAG}_$CNT.json 2> $ROOT/gpurun_out/r06_pmc_${TAG}_$CNT.err)
        echo "pmc $CNT rc=$?"
      done
      # contexts of the 12 decode steps: 4051..4062 keys -> mean 4056.5
      python scripts/pmc_summary.py pmc --attn-context 4056.5 $(find /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/$OUT 2> gpurun_out/r06_pmc_${TAG}_summary.err
      python - <<PY
import json
d=json.load(open("gpurun_out/$OUT"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1].get("hbm_read_bytes_per_launch",0))[:8]:
    print(f"{k[:70]:70s} read {v.get('hbm_read_bytes_per_launch',0)/1e6:9.2f} MB  write {v.get('hbm_write_bytes_per_launch',0)/1e6:8.2f} MB")
PY
      ;;
    pmcsq)   # SQ counters of the DiT front-end: MFMA-busy share per kernel
      rm -rf /tmp/pmcsq_dit
      (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmcsq_dit -o pmc -- python $ROOT/scripts/bench_dit.py 16 3 fp16 > $ROOT/gpurun_out/r06_pmcsq_dit.log 2>&1)
      echo "pmcsq dit rc=$?"
      python scripts/pmc_summary.py pmc $(find /tmp/pmcsq_dit -name "*counter_collection.csv") > gpurun_out/r06_pmc_sq_dit.json 2> gpurun_out/r06_pmcsq_dit.err
      python - <<PY
import json
d=json.load(open("gpurun_out/r06_pmc_sq_dit.json"))
for k,v in d["kernels"].items():
    if not any(t in k for t in ("gemm_hh","flash_attn","ln_modulate")): continue
    g=v.get("GRBM_GUI_ACTIVE",{}).get("mean",0); m=v.get("SQ_VALU_MFMA_BUSY_CYCLES",{}).get("mean",0)
    print(f"{k[:70]:70s} disp {v['GRBM_GUI_ACTIVE']['dispatches']:5d} mfma_busy {m/(g*128) if g else 0:.3f} gui_active_cycles {g:.0f}")
PY
      ;;
    ditprof) rm -rf /tmp/prof_dit
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dit -o r06 -- python $ROOT/scripts/bench_dit.py 16 10 fp16 > $ROOT/gpurun_out/r06_rocprof_dit.json 2> $ROOT/gpurun_out/r06_rocprof_dit.err)
      echo "rocprof rc=$?"; tail -1 gpurun_out/r06_rocprof_dit.json
      find /tmp/prof_dit -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_dit_fp16_kernel_stats.csv; head -12 gpurun_out/r06_dit_fp16_kernel_stats.csv | cut -c1-160 ;;
    clock)   # shader clock under the DiT GEMMs vs under the decode attention: rocm-smi while the workload runs
      { timeout 120 python scripts/bench_dit.py 16 40 fp16 > /dev/null 2>&1 & sleep 25; for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4; sleep 2; done; wait; } 2>&1 | tee gpurun_out/r06_clock_under_dit.log ;;
    suite)  # the whole GPU suite with the values the tests print: -s -rA, the value lines kept verbatim
            timeout 1700 python -m pytest tests -m gpu -s -rA -p no:cacheprovider --timeout 900 2>&1 | filt > gpurun_out/r06_gpu_tests_full.log
            tail -4 gpurun_out/r06_gpu_tests_full.log | tee gpurun_out/r06_gpu_tests.log
            grep -E "dlogit|abs err|max err|err vs|rel err|tok/s|diverg|swaps|cond err|^PASSED|^FAILED|^ERROR|passed|failed" gpurun_out/r06_gpu_tests_full.log | grep -v "^tests/.*\.py \." > gpurun_out/r06_parity_values.log
            wc -l gpurun_out/r06_parity_values.log; grep -c "^PASSED" gpurun_out/r06_parity_values.log ;;
    *) echo "unknown section $SEC" ;;
  esac
done
