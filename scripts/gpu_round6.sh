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
    suite)  # the whole GPU suite with the values the tests print: -s -rA, the value lines kept verbatim
            timeout 1700 python -m pytest tests -m gpu -s -rA -p no:cacheprovider --timeout 900 2>&1 | filt > gpurun_out/r06_gpu_tests_full.log
            tail -4 gpurun_out/r06_gpu_tests_full.log | tee gpurun_out/r06_gpu_tests.log
            grep -E "dlogit|abs err|max err|err vs|rel err|tok/s|diverg|swaps|cond err|^PASSED|^FAILED|^ERROR|passed|failed" gpurun_out/r06_gpu_tests_full.log | grep -v "^tests/.*\.py \." > gpurun_out/r06_parity_values.log
            wc -l gpurun_out/r06_parity_values.log; grep -c "^PASSED" gpurun_out/r06_parity_values.log ;;
    *) echo "unknown section $SEC" ;;
  esac
done
