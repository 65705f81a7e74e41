#!/bin/bash
# Round-4 measurement sections (one gpurun call each or several): usage  bash scripts/gpu_round3.sh <section> [...]
#   bench   python bench.py (configs[1], unprofiled)                     -> gpurun_out/r04_bench.json
#   prof    rocprofv3 --kernel-trace --stats of bench.py configs[1]      -> gpurun_out/r04_bench_kernel_stats.csv + roofline check
#   pmc     separate --pmc FETCH_SIZE / WRITE_SIZE passes, configs[1]    -> gpurun_out/r04_pmc_hbm_summary.json
#   bench3 / prof3 / pmc3   the same for --config 3 (32 clouds per GPU, fp16)
#   bench2  python bench.py --config 2 (B = 32 sample mode, T = 16000)   -> gpurun_out/r04_bench_config2.json
#   suite   the whole GPU test suite (~5.5 min)                          -> gpurun_out/r04_gpu_tests.log
#   tests   kernel units + end-to-end parity only (~2.5 min)             -> gpurun_out/r04_gpu_tests_quick.log
#   batch   scripts/bench_batch.py, B = 1..32, T = 1000, fp32 and fp16   -> gpurun_out/r04_batch_table_v2.log
#   dit     DiT tests + scripts/bench_dit.py 16 10 fp16                  -> gpurun_out/r04_dit.log
#   prefill scripts/prefill_time.py (fp16 B = 1, 8; fp32 B = 1)          -> gpurun_out/r04_prefill_time_final.log
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
filt() { grep -v amdgpu.ids; }
for SEC in "$@"; do
  case $SEC in
    bench)  timeout 900 python bench.py 2> gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench.json; filt < gpurun_out/r04_bench.err | tail -5; head -c 600 gpurun_out/r04_bench.json; echo ;;
    bench3) timeout 900 python bench.py --config 3 --steps 1 --warmup 1 2> gpurun_out/r04_bench3.err | tail -1 > gpurun_out/r04_bench_config3.json; filt < gpurun_out/r04_bench3.err | tail -3; head -c 600 gpurun_out/r04_bench_config3.json; echo ;;
    bench2) timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 2> gpurun_out/r04_bench2.err | tail -1 > gpurun_out/r04_bench_config2.json; filt < gpurun_out/r04_bench2.err | tail -3; head -c 600 gpurun_out/r04_bench_config2.json; echo ;;
    suite)  timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -x 2>&1 | filt | tail -15 | tee gpurun_out/r04_gpu_tests.log ;;
    tests)  timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider --timeout 900 -x 2>&1 | filt | tail -8 | tee gpurun_out/r04_gpu_tests_quick.log ;;
    batch)  { timeout 400 python scripts/bench_batch.py 1,2,3,4,5,6,7,8,12,16,32 1000 1000 fp32 2>&1 | filt
              timeout 400 python scripts/bench_batch.py 1,2,3,4,5,6,7,8,12,16,32 1000 1000 fp16 2>&1 | filt; } | tee gpurun_out/r04_batch_table_v2.log | grep aggregate | cut -c1-200 ;;
    dit)    { timeout 600 python -m pytest tests/test_gpu_dit.py -q -m gpu -s -p no:cacheprovider --timeout 400 2>&1 | filt | tail -12
              timeout 300 python scripts/bench_dit.py 16 10 fp16 2>&1 | filt | tail -1; } | tee gpurun_out/r04_dit.log ;;
    prefill) { timeout 200 python scripts/prefill_time.py fp16 1,8 2>&1 | filt | tail -2
               timeout 200 python scripts/prefill_time.py fp32 1 2>&1 | filt | tail -1; } | tee gpurun_out/r04_prefill_time_final.log ;;
    prof|prof3)
      if [ $SEC = prof ]; then TAG=bench; ARGS="--no-fast-extra --cpu-steps 0"; else TAG=config3; ARGS="--config 3"; fi
      rm -rf /tmp/prof_$TAG
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o r04 -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $ROOT/gpurun_out/r04_rocprof_$TAG.json 2> $ROOT/gpurun_out/r04_rocprof_$TAG.err)
      echo "rocprof rc=$?"
      find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_${TAG}_kernel_stats.csv
      python scripts/roofline_from_rocprof.py gpurun_out/r04_${TAG}_kernel_stats.csv gpurun_out/r04_rocprof_$TAG.json --tol 0.10 2>&1 | tee gpurun_out/r04_roofline_check_$TAG.log | head -30 ;;
    pmc|pmc3)
      if [ $SEC = pmc ]; then TAG=bench; ARGS="--no-fast-extra --cpu-steps 0"; else TAG=config3; ARGS="--config 3"; fi
      for CNT in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${TAG}_$CNT
        (cd /tmp && ER_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$CNT -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --tokens 12 --resume-len 2000 $ARGS > /tmp/pmc_${TAG}_$CNT.json 2> $ROOT/gpurun_out/r04_pmc_${TAG}_$CNT.err)
        echo "pmc $CNT rc=$?"
      done
      # contexts of the 12 decode steps: 4051..4062 keys -> mean 4056.5
      python scripts/pmc_summary.py pmc --attn-context 4056.5 $(find /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/r04_pmc_hbm_${TAG}_summary.json 2> gpurun_out/r04_pmc_${TAG}_summary.err
      python - <<PY
import json
d=json.load(open("gpurun_out/r04_pmc_hbm_${TAG}_summary.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1].get("hbm_read_bytes_per_launch",0))[:8]:
    print(f"{k[:70]:70s} read {v.get('hbm_read_bytes_per_launch',0)/1e6:9.2f} MB  write {v.get('hbm_write_bytes_per_launch',0)/1e6:8.2f} MB")
PY
      ;;
  esac
done
