#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Compiles the reference's own meto tokenizer (its only native code:
# meto/src/bindings.cpp + header-only engines) from the sources where they lie under /root/reference
# into oracle/_ref/ (git-ignored, travels to the GPU box).  Used to validate the native detokeniser
# (edgerunner_amd/csrc/meto_decode.h) and to generate token fixtures.  No reference source is copied.
set -e
REF=${EDGERUNNER_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
mkdir -p "$HERE/_ref"
SUF=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
OUT="$HERE/_ref/_meto$SUF"
if [ -f "$OUT" ] && [ "$OUT" -nt "$REF/meto/src/bindings.cpp" ]; then exit 0; fi
g++ -O2 -std=c++17 -shared -fPIC -w $(python3 -m pybind11 --includes) -I "$REF/meto/include" "$REF/meto/src/bindings.cpp" -o "$OUT"
echo "[oracle] built $OUT"
