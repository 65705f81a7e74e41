#!/bin/bash
# PMC passes (separate rocprofv3 runs, counters only + kernel trace): HBM bytes per launch of the decode kernels.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$CNT
  (cd /tmp && ER_NO_GRAPH=1 timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_$CNT -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-steps 0 --tokens 12 --no-fast-extra > gpurun_out_pmc_$CNT.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_$CNT.err)
  echo "rc=$?"; ls -la /tmp/pmc_$CNT | head -6; grep -v amdgpu.ids gpurun_out/pmc_$CNT.err | tail -4
done
python scripts/pmc_summary.py pmc $(find /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv") > gpurun_out/pmc_summary.json 2> gpurun_out/pmc_summary.err
head -c 4000 gpurun_out/pmc_summary.json; tail -3 gpurun_out/pmc_summary.err
du -sh gpurun_out
