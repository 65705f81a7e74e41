#!/bin/bash
# One gpurun call of round 2.  Usage (build container):
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round2.sh tests tune bench prof pmc probes'
# Sections run in the order given; every section writes its log under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
filt() { grep -v amdgpu.ids; }
for SEC in "$@"; do
  echo "=================== $SEC"
  case "$SEC" in
    kernels)
      timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 300 2>&1 | filt | tail -30 | tee gpurun_out/test_kernels.log ;;
    parity)
      timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -p no:cacheprovider --timeout 1200 2>&1 | filt | tail -80 | tee gpurun_out/test_parity.log ;;
    scripts)
      timeout 1800 python -m pytest tests/test_gpu_scripts.py -q -m gpu -s -p no:cacheprovider --timeout 1200 2>&1 | filt | tail -60 | tee gpurun_out/test_scripts.log ;;
    dit)
      timeout 1800 python -m pytest tests/test_gpu_dit.py -q -m gpu -s -p no:cacheprovider --timeout 1200 2>&1 | filt | tail -40 | tee gpurun_out/test_dit.log ;;
    alltests)
      timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 1500 --durations=15 2>&1 | filt | tail -60 | tee gpurun_out/test_all.log ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | filt | tail -5 | tee gpurun_out/smoke.log ;;
    tune)
      TUNE_CONFIGS=${TUNE_CONFIGS:-'[{"ER_ATTN_V":1},{"ER_COMBINE_V":1},{"ER_ATTN_V":1,"ER_COMBINE_V":1}]'} \
        timeout 1200 python scripts/tune_decode.py 2>&1 | filt | tee gpurun_out/tune.log ;;
    bench)
      timeout 1200 python bench.py --steps ${BENCH_STEPS:-2} --warmup 1 2> gpurun_out/bench.err | tee gpurun_out/bench.json
      filt < gpurun_out/bench.err | tail -20 ;;
    bench32)
      timeout 1200 python bench.py --steps 1 --warmup 1 --batch-per-gpu 32 --precision ${B32_PRECISION:-fp16} --cpu-steps 0 2> gpurun_out/bench32.err | tee gpurun_out/bench32.json
      filt < gpurun_out/bench32.err | tail -8 ;;
    prof)
      rm -rf /tmp/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r02 -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-fast-extra > $ROOT/gpurun_out/rocprof_bench.json 2> $ROOT/gpurun_out/rocprof.err)
      mkdir -p gpurun_out/prof; find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
      head -12 gpurun_out/prof/*kernel_stats.csv 2>/dev/null | cut -c1-160; tail -1 gpurun_out/rocprof_bench.json | cut -c1-600
      # compare with the UNPROFILED bench line of this call when there is one (section `bench` before `prof`): the HIP-event sweep
      # inside a traced process can be inflated by the tracer (final run of round 2: 17 us vs 11 us in the CSV and in the unprofiled line)
      REF=gpurun_out/rocprof_bench.json; [ -s gpurun_out/bench.json ] && REF=gpurun_out/bench.json
      python scripts/roofline_from_rocprof.py $(ls gpurun_out/prof/*kernel_stats.csv | head -1) $REF 2>&1 | tee gpurun_out/roofline_check.log ;;
    profpre)   # kernel breakdown of encode + prefill (short decode), exact and fast mode
      for PR in fp32 fp16; do
        rm -rf /tmp/profpre_$PR
        (cd /tmp && TUNE_PRECISION=$PR TUNE_TOKENS=8 TUNE_CONFIGS='[]' timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profpre_$PR -o pre -- python $ROOT/scripts/tune_decode.py > $ROOT/gpurun_out/profpre_$PR.log 2>&1)
        F=$(find /tmp/profpre_$PR -name "*kernel_stats.csv" | head -1)
        if [ -n "$F" ]; then cp $F gpurun_out/profpre_${PR}_kernel_stats.csv; echo "--- $PR"; head -14 $F | cut -c1-170; fi
      done ;;
    pmcsq)     # MFMA-busy / LDS counters of encode + prefill (short decode), exact and fast mode
      for PR in fp32 fp16; do
        rm -rf /tmp/pmcsq_$PR
        (cd /tmp && TUNE_PRECISION=$PR TUNE_TOKENS=4 TUNE_CONFIGS='[]' ER_NO_GRAPH=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmcsq_$PR -o pmc -- python $ROOT/scripts/tune_decode.py > $ROOT/gpurun_out/pmcsq_$PR.log 2>&1)
        python scripts/pmc_summary.py pmc $(find /tmp/pmcsq_$PR -name "*counter_collection.csv") > gpurun_out/pmcsq_$PR.json 2>> gpurun_out/pmcsq_$PR.log
        python - <<PY
import json
d = json.load(open("gpurun_out/pmcsq_$PR.json"))
for k, v in d["kernels"].items():
    if any(t in k for t in ("gemm", "flash", "layernorm")):
        g = {c: v[c]["mean"] for c in v if isinstance(v[c], dict)}
        busy = g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(g.get("GRBM_GUI_ACTIVE", 1) * 128, 1)
        print("$PR", k[:60], "n=%d" % v["GRBM_GUI_ACTIVE"]["dispatches"], "mfma_busy=%.3f" % busy,
              "lds_conflict/active=%.3f" % (g.get("SQ_LDS_BANK_CONFLICT", 0) / max(g.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
              "gui_active=%.0f" % g.get("GRBM_GUI_ACTIVE", 0))
PY
      done ;;
    pmc)
      bash scripts/gpu_pmc.sh 2>&1 | tail -14 ;;
    probes)
      for P in ${PROBES:-xcd_barrier_probe persistent_chain_probe}; do
        if [ -x scripts/probes/$P ]; then echo "--- $P"; timeout 300 scripts/probes/$P 2>&1 | tee gpurun_out/$P.log | tail -40; fi
      done ;;
    *) echo "unknown section $SEC" ;;
  esac
done
du -sh gpurun_out
echo "== done"
