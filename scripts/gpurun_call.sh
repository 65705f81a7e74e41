#!/bin/bash
# Build-container helper: rebuild every HIP artefact, check that the library exports the whole header, THEN call gpurun
# (a stale .so once cost a GPU call).  Usage: scripts/gpurun_call.sh <timeout_s> '<remote command>'
set -e
cd "$(dirname "$0")/.."
ER_BUILD_REUSE=1 python -c "import __graft_entry__ as g; g.build()" | tail -1
for P in scripts/probes/*_probe.hip; do
  B=${P%.hip}
  if [ ! -x $B ] || [ $P -nt $B ] || [ edgerunner_amd/csrc/k_gemm.h -nt $B ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-pass-failed -I edgerunner_amd/csrc -o $B $P; fi
done
python -m pytest tests/test_abi.py -q -x 2>&1 | tail -1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
