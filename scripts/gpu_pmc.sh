#!/bin/bash
# PMC passes (separate rocprofv3 runs, counters only + kernel trace): HBM bytes per launch of the decode kernels.
# The decode starts at context 4050 (--resume-len 2000) so that the 12 steps run at the MEAN context length of the
# benchmarked T = 4000 run (4050.5): the attention kernel's traffic is measured where bench.py quotes its bytes.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$CNT
  (cd /tmp && ER_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_$CNT -o pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-steps 0 --tokens 12 --resume-len 2000 --no-fast-extra > /tmp/pmc_$CNT.json 2> $ROOT/gpurun_out/pmc_$CNT.err)
  echo "rc=$?"; ls /tmp/pmc_$CNT | head -3; grep -v amdgpu.ids gpurun_out/pmc_$CNT.err | tail -3
done
# contexts of the 12 decode steps: 4051..4062 keys -> mean 4056.5
python scripts/pmc_summary.py pmc --attn-context 4056.5 $(find /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/pmc_summary.json 2> gpurun_out/pmc_summary.err
head -c 3000 gpurun_out/pmc_summary.json; tail -3 gpurun_out/pmc_summary.err
