#!/bin/bash
# Round-3 measurement sections (one gpurun call each or several): usage  bash scripts/gpu_round3.sh <section> [...]
#   bench   python bench.py (configs[1], unprofiled)                     -> gpurun_out/r03_bench.json
#   prof    rocprofv3 --kernel-trace --stats of bench.py configs[1]      -> gpurun_out/r03_bench_kernel_stats.csv + roofline check
#   pmc     separate --pmc FETCH_SIZE / WRITE_SIZE passes, configs[1]    -> gpurun_out/r03_pmc_hbm_summary.json
#   bench3 / prof3 / pmc3   the same for --config 3 (32 clouds per GPU, fp16)
#   bench2  python bench.py --config 2 (B = 32 sample mode, T = 16000)   -> gpurun_out/r03_bench_config2.json
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
filt() { grep -v amdgpu.ids; }
for SEC in "$@"; do
  case $SEC in
    bench)  timeout 900 python bench.py 2> gpurun_out/r03_bench.err | tail -1 > gpurun_out/r03_bench.json; filt < gpurun_out/r03_bench.err | tail -5; head -c 600 gpurun_out/r03_bench.json; echo ;;
    bench3) timeout 900 python bench.py --config 3 --steps 1 --warmup 1 2> gpurun_out/r03_bench3.err | tail -1 > gpurun_out/r03_bench_config3.json; filt < gpurun_out/r03_bench3.err | tail -3; head -c 600 gpurun_out/r03_bench_config3.json; echo ;;
    bench2) timeout 1500 python bench.py --config 2 --steps 1 --warmup 0 2> gpurun_out/r03_bench2.err | tail -1 > gpurun_out/r03_bench_config2.json; filt < gpurun_out/r03_bench2.err | tail -3; head -c 600 gpurun_out/r03_bench_config2.json; echo ;;
    prof|prof3)
      if [ $SEC = prof ]; then TAG=bench; ARGS="--no-fast-extra --cpu-steps 0"; else TAG=config3; ARGS="--config 3"; fi
      rm -rf /tmp/prof_$TAG
      (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o r03 -- python $ROOT/bench.py --steps 1 --warmup 0 $ARGS > $ROOT/gpurun_out/r03_rocprof_$TAG.json 2> $ROOT/gpurun_out/r03_rocprof_$TAG.err)
      echo "rocprof rc=$?"
      find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_${TAG}_kernel_stats.csv
      python scripts/roofline_from_rocprof.py gpurun_out/r03_${TAG}_kernel_stats.csv gpurun_out/r03_rocprof_$TAG.json --tol 0.10 2>&1 | tee gpurun_out/r03_roofline_check_$TAG.log | head -30 ;;
    pmc|pmc3)
      if [ $SEC = pmc ]; then TAG=bench; ARGS="--no-fast-extra --cpu-steps 0"; else TAG=config3; ARGS="--config 3"; fi
      for CNT in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pmc_${TAG}_$CNT
        (cd /tmp && ER_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$CNT -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --tokens 12 --resume-len 2000 $ARGS > /tmp/pmc_${TAG}_$CNT.json 2> $ROOT/gpurun_out/r03_pmc_${TAG}_$CNT.err)
        echo "pmc $CNT rc=$?"
      done
      # contexts of the 12 decode steps: 4051..4062 keys -> mean 4056.5
      python scripts/pmc_summary.py pmc --attn-context 4056.5 $(find /tmp/pmc_${TAG}_FETCH_SIZE /tmp/pmc_${TAG}_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/r03_pmc_hbm_${TAG}_summary.json 2> gpurun_out/r03_pmc_${TAG}_summary.err
      python - <<PY
import json
d=json.load(open("gpurun_out/r03_pmc_hbm_${TAG}_summary.json"))
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1].get("hbm_read_bytes_per_launch",0))[:8]:
    print(f"{k[:70]:70s} read {v.get('hbm_read_bytes_per_launch',0)/1e6:9.2f} MB  write {v.get('hbm_write_bytes_per_launch',0)/1e6:8.2f} MB")
PY
      ;;
  esac
done
