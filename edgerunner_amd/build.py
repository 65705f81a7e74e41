"""Builds the HIP extension in-tree: ``edgerunner_amd/libedgerunner_hip.so`` for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the snapshot (it is git-ignored, not
gpurun-ignored).  ``python -m edgerunner_amd.build`` or ``__graft_entry__.build()``.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG, "csrc", "er_api.hip")
LIB = os.path.join(PKG, "libedgerunner_hip.so")
# -amdgpu-kernarg-preload-count: leading SCALAR kernel arguments arrive in SGPRs at wave launch (the single-row decode kernels lead
# their argument lists with the pointers their first loads need: csrc/k_gemv.h)
# -amdgpu-mfma-vgpr-form: MFMA accumulators live in VGPRs (gfx950's register file is unified; no kernel of the library needs more than
# its arch-VGPR budget for them): the softmax / epilogue VALU work reads them in place instead of through v_accvgpr_read / _write copies
# (flash_attn_hh_kernel: 224 of its 1350 instructions; guided DiT forward 9.02 -> 8.73 ms same box, profiles/r05_ab_mfma_vgpr_form.log)
# Both are internal LLVM options of ROCm 7.x's hipcc; tests/test_isa_hygiene.py::
# test_codegen_flags_are_accepted_and_take_effect fails with a clear message if a compiler drops one or stops honouring it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-pass-failed", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-shared", "-fPIC"]


def sources():
    # build.py itself is a source: a change of FLAGS alone must rebuild a stale library
    d = os.path.join(PKG, "csrc")
    return [os.path.join(d, f) for f in sorted(os.listdir(d))] + [os.path.join(PKG, "..", "include", "edgerunner_hip.h"), os.path.abspath(__file__)]


def up_to_date() -> bool:
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(s) <= t for s in sources())


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print("[edgerunner_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=os.path.join(PKG, "csrc"))
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
