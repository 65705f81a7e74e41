#!/bin/bash
# One gpurun call: kernel unit tests, end-to-end parity, smoke, bench and a rocprofv3 kernel trace.
# Usage (build container):  gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [quick|full|final|tune]'
set -u
MODE=${1:-full}
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
python - <<'PY' > gpurun_out/env.txt 2>&1
import torch, os
print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
print("cpus", os.cpu_count())
PY
echo "== kernel unit tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider --timeout 300 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/test_kernels.log
echo "== parity tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -p no:cacheprovider --timeout 900 2>&1 | grep -v amdgpu.ids | tail -60 | tee gpurun_out/test_parity.log
if [ "$MODE" = "full" ] || [ "$MODE" = "final" ]; then
  echo "== DiT front-end tests"; timeout 900 python -m pytest tests/test_gpu_dit.py -q -m gpu -s -p no:cacheprovider --timeout 600 2>&1 | grep -v amdgpu.ids | tail -20 | tee gpurun_out/test_dit.log
  echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/smoke.log
  echo "== bench"; timeout 900 python bench.py --steps 2 --warmup 1 2> gpurun_out/bench.err | tee gpurun_out/bench.json
  grep -v amdgpu.ids gpurun_out/bench.err | tail -30
  echo "== rocprof"; rm -rf /tmp/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-steps 0 --no-fast-extra > $GRAFT_REPO_ROOT/gpurun_out/rocprof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/rocprof.err)
  mkdir -p gpurun_out/prof; find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
  head -14 gpurun_out/prof/*kernel_stats.csv 2>/dev/null; cat gpurun_out/rocprof_bench.json
fi
if [ "$MODE" = "final" ]; then
  echo "== PMC passes"; bash scripts/gpu_pmc.sh 2>&1 | tail -12
fi
if [ "$MODE" = "tune" ] || [ "$MODE" = "full" ]; then
  echo "== tune"
  TUNE_CONFIGS=${TUNE_CONFIGS:-'[{"ER_ATTN_STEPS":2},{"ER_ATTN_STEPS":8},{"ER_RW_QKV":2}]'} \
    timeout 900 python scripts/tune_decode.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tune.log
fi
du -sh gpurun_out
echo "== done"
