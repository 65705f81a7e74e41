"""Code-generation guards for the decode hot path (no GPU needed: hipcc cross-compiles gfx950 here).

Round 3 found 5 % of configs[1] in things the ISA showed and the source did not: wave reductions lowered to
dependent ds_bpermute round trips, a scalar load of pos[b] stalling the LayerNorm prologue through the shared lgkmcnt, kernel
arguments and integer divisions sunk behind the final reduction, a vmcnt(0) in front of the weight stream, 64-bit index divisions
per element.  These tests compile csrc/er_api.hip to device assembly once and assert that none of them is back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "edgerunner_amd", "csrc")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    """{mangled kernel name: [instruction lines]} of the gfx950 code object."""
    from edgerunner_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "er_api.s"
    flags = [f for f in B.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.run([hipcc] + flags + ["--cuda-device-only", "-S", "-o", str(out), "er_api.hip"], check=True, cwd=CSRC,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    funcs = {}
    for name, body in re.findall(r"\n(_Z[^\n:]*):\s*; @[^\n]*\n(.*?)s_endpgm", text, flags=re.S):
        funcs[name] = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    assert len(funcs) > 200, "device assembly not parsed"
    # kernel descriptors: {mangled name: preloaded kernel-argument SGPRs}
    funcs["__preload__"] = {n: int(v) for n, v in re.findall(r"\.amdhsa_kernel (\S+)\n(?:.*\n)*?\s*\.amdhsa_user_sgpr_kernarg_preload_length (\d+)", text)}
    return funcs


def select(kernels, pattern):
    sel = {n: b for n, b in kernels.items() if not n.startswith("__") and re.search(pattern, n)}
    assert sel, f"no kernel matches {pattern}"
    return sel


def count(body, needle):
    return sum(needle in l for l in body)


def test_no_lds_crossbar_in_wave_reductions(kernels):
    """er_common.h's butterflies run on v_permlane*_swap + DPP; a ds_bpermute in these kernels means a __shfl crept back in."""
    hot = select(kernels, r"gemv_kernelI|attn_decode3_kernel|attn_decode2_kernel|attn_decode_kernelI|attn_stream_kernel|"
                          r"attn_combine2_kernel|sample_head_kernel|flash_attn_f32_kernel|flash_attn_f16s_kernel|gemv_batched_kernel|"
                          r"prep_rows_kernel|layernorm_rows_kernel|ln_modulate_rows_kernel")
    bad = {n: count(b, "ds_bpermute") for n, b in hot.items() if count(b, "ds_bpermute")}
    assert not bad, bad
    for n, b in select(kernels, r"outproj_merge_kernel").items():
        assert count(b, "ds_bpermute") <= 16, (n, "only the 16 weight broadcasts may use the crossbar")
        assert count(b, "v_permlane16_swap") >= 2, n


def single_row_gemvs(kernels):
    # gemv_kernel<WT, KS = 1, NB, RW, PRO, EPI, NW>: the LayerNorm-prologue / plain single-row kernels of the decode step
    return select(kernels, r"gemv_kernelI(f|DF16_)Li1ELi[1-8]ELi[12]ELi[012]ELi[0123]ELi\d+EEE")


def test_gemv_tail_is_arithmetic_and_one_store(kernels):
    """Behind the final 64-lane reduction: no kernel-argument loads, no integer division (the destination address is finished at
    entry, k_gemv.h gemv_epi_prefetch)."""
    for n, b in single_row_gemvs(kernels).items():
        last = max(i for i, l in enumerate(b) if "v_permlane32_swap" in l)
        tail = b[last:]
        assert not [l for l in tail if l.startswith("s_load")], (n, "s_load behind the reduction")
        assert not count(tail, "v_rcp_iflag"), (n, "integer division behind the reduction")
        assert len(tail) < 200, (n, len(tail))


def test_gemv_weight_stream_is_issued_before_any_drain(kernels):
    """No s_waitcnt vmcnt(0) in front of the last weight load: everything the prologue / epilogue needs queues FIRST in vmcnt order,
    and nothing up there may depend on the position word (a pin on pos-dependent address arithmetic once put a drain here)."""
    for n, b in single_row_gemvs(kernels).items():
        w = [i for i, l in enumerate(b) if l.startswith("global_load_dwordx4") and l.endswith(" nt")]
        assert w, (n, "weight loads are nontemporal dwordx4 loads")
        drains = [i for i, l in enumerate(b[:w[-1]]) if re.match(r"s_waitcnt vmcnt\(0\)", l)]
        assert not drains, (n, drains)


def test_weight_stream_goes_out_early(kernels):
    """The epilogue's address arithmetic (two integer divisions per row for the KV append) sits BEHIND the weight issue.  (Round 3
    also pinned fc2's input slice in front of its weight loads; measured in round 4 that order was slower - 8.10 vs 7.97 us per
    launch, profiles/r04_ab_b5b758c_and_om_rpw.log - and it was reverted.)"""
    for n, b in select(kernels, r"gemv_kernelI(f|DF16_)Li1ELi1ELi[12]ELi[12]ELi3ELi\d+EEE").items():      # qkv, one row
        w = [i for i, l in enumerate(b) if l.startswith("global_load_dwordx4") and l.endswith(" nt")]
        assert w[-1] < 140, (n, w[-1], "weight loads delayed by epilogue arithmetic")
        assert not count(b[:w[0]], "v_rcp_iflag"), (n, "integer division in front of the weight stream")


def test_qkv_position_word_is_a_vector_load(kernels):
    """pos[b] of the KV-append epilogue must not be a scalar load: scalar loads share lgkmcnt with LDS, and the wait in front of
    the LayerNorm prologue's first barrier then sits out its memory round trip."""
    qkv = select(kernels, r"gemv_kernelI(f|DF16_)Li1ELi1ELi[12]ELi1ELi3ELi\d+EEE")       # PRO_LN, EPI_QKV, one row
    for n, b in qkv.items():
        first_w = min(i for i, l in enumerate(b) if l.startswith("global_load_dwordx4") and l.endswith(" nt"))
        scalar_data_loads = [l for l in b[first_w:] if re.match(r"s_load_dword s\d+, s\[\d+:\d+\], 0x0$", l)]
        assert not scalar_data_loads, (n, scalar_data_loads)


def test_attention_entry_loads_its_arguments_once(kernels):
    """attn_decode3_kernel: one batch of kernel-argument loads, then the length; nothing is fetched behind the early-exit test."""
    for n, b in select(kernels, r"attn_decode3_kernel").items():
        first = min(i for i, l in enumerate(b) if l.startswith("global_load"))
        late = [l for l in b[first:] if l.startswith("s_load")]
        assert not late, (n, late)
        assert not count(b[:200], "v_rcp_iflag"), (n, "chunk bounds are shifts, not divisions")


def test_row_kernels_have_no_per_element_index_division(kernels):
    """Element-wise kernels of the per-layer prefill path take a row per block and a float4 per thread; a 64-bit index division
    per element made kv_scatter_half_kernel compute-bound (26 us per layer for 63 MB)."""
    limits = {r"kv_scatter_half_kernel": 140, r"split_rows_f16_kernel": 140, r"splitk_finish_kernelILi2E": 320}      # three slice-count forms (4 and 16 unrolled, generic loop)
    for pat, lim in limits.items():
        for n, b in select(kernels, pat).items():
            assert len(b) < lim, (n, len(b))


def test_flash_attn_hh_counted_waits_and_no_scratch(kernels):
    """flash_attn_hh_kernel counts its LDS-DMA pieces by hand (vmcnt(8) / (4) / (0), each fused with the s_barrier behind it in one
    asm statement): a scratch access or any other compiler-issued VMEM operation inside the tile loop would silently shift those
    counts (round-3 advisor)."""
    for n, b in select(kernels, r"flash_attn_hh_kernel").items():
        assert not [l for l in b if l.startswith(("scratch_", "buffer_"))], (n, "scratch / buffer access in the LDS-DMA attention kernel")
        dma = [i for i, l in enumerate(b) if l.startswith("global_load_lds_dwordx4")]
        assert len(dma) >= 8, n
        for cnt in (8, 4, 0):
            idx = [i for i, l in enumerate(b) if l.startswith(f"s_waitcnt vmcnt({cnt})") and i > dma[0]]
            assert idx, (n, f"no vmcnt({cnt}) wait behind the first LDS-DMA piece")
            assert any(b[i + 1].startswith("s_barrier") for i in idx if i + 1 < len(b)), (n, f"vmcnt({cnt}) is not followed by its barrier")
        # between the first DMA piece and the end of the loop the only vector-memory loads are the DMA pieces themselves
        loads = [l for l in b[dma[0]:dma[-1] + 1] if l.startswith("global_load") and not l.startswith("global_load_lds")]
        assert not loads, (n, loads[:3])


def test_prefill_tail_gemvs_are_in_the_code_object(kernels):
    """er_prefill hands the prefix's 1..8 ragged rows to gemv_kernel<float, KS, NB, RW, PRO_NONE, EPI> in groups of up to four rows
    (er_api.hip prefill_tail_rows / linear_tail): out_proj <1, NB, 1, 0, RESID, 4>, fc1 <1, NB, 2, 0, RELU, 4>, fc2 <4, NB, 2, 0, RESID, 4>
    for NB = 1..4, and none of them spills."""
    for ks, rw, epi in ((1, 1, 2), (1, 2, 1), (4, 2, 2)):
        for nb in (1, 2, 3, 4):
            sel = select(kernels, rf"gemv_kernelIfLi{ks}ELi{nb}ELi{rw}ELi0ELi{epi}ELi4EEE")
            assert sel, (ks, nb, rw, epi)       # (select() asserts too: an instantiation that is gone must not pass vacuously)
            for n, b in sel.items():
                assert not [l for l in b if l.startswith("scratch_")], (n, "scratch access")


def test_gemm_hh256_main_loop_shape(kernels):
    """The 256 x 256 / 8-wave LDS-DMA GEMM (k_gemm.h, round 5): no scratch at 200+ registers, the k-loop's 32 MFMAs fed by 24
    fragment reads with at most two full lgkmcnt(0) drains in front of MFMAs (fragments double-buffered: left alone hipcc put one in
    front of every four MFMAs), eight asm LDS-DMA pieces per k-step and ONE counted wait + raw barrier - no compiler-issued
    vmcnt wait anywhere in the loop (the pieces are invisible to its bookkeeping)."""
    for n, b in select(kernels, r"gemm_hh256_kernel").items():
        assert not [l for l in b if l.startswith("scratch_")], (n, "scratch access")
        mf = [i for i, l in enumerate(b) if l.startswith("v_mfma_f32_32x32x16_f16")]
        assert len(mf) == 32, (n, len(mf))
        loop = b[mf[0] - 40:mf[-1] + 6]
        assert count(loop, "ds_read_b128") == 24, (n, count(loop, "ds_read_b128"))
        drains = [l for l in loop if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l and "vmcnt" not in l]
        assert len(drains) <= 2, (n, drains)
        vm = [l for l in loop if l.startswith("s_waitcnt") and "vmcnt" in l]
        assert vm == ["s_waitcnt vmcnt(0) lgkmcnt(0)"], (n, vm)
        assert count(b, "global_load_lds_dwordx4") == 16, (n, "8 pieces in the prologue + 8 in the loop")


def test_codegen_flags_are_accepted_and_take_effect(kernels, tmp_path):
    """build.py compiles the whole library with two internal LLVM options (-amdgpu-kernarg-preload-count=16, -amdgpu-mfma-vgpr-form).
    A hipcc that no longer knows one of them must fail HERE with a clear message (not as an "Unknown command line argument" in the
    middle of build()), and each must still do what the kernels were tuned for: the MFMA kernels keep their accumulators in VGPRs (no
    v_accvgpr copies in the LDS-DMA attention kernel, no MFMA kernel spilling to scratch under the 256-VGPR cap the flag implies), and
    the single-row decode kernels receive their leading scalar arguments preloaded (no s_load of kernel arguments in front of the
    first vector load)."""
    from edgerunner_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "probe.hip"
    src.write_text("#include <hip/hip_runtime.h>\n__global__ void k(float* p, int n) { if ((int)threadIdx.x < n) p[threadIdx.x] = 1.f; }\n")
    for i, f in enumerate(B.FLAGS):
        if f != "-mllvm":
            continue
        opt = B.FLAGS[i + 1]
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "--cuda-device-only", "-c", "-mllvm", opt, "-o", str(tmp_path / "probe.o"), str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, f"this hipcc rejects `-mllvm {opt}` (edgerunner_amd/build.py FLAGS; INTEGRATION.md section 3): {r.stderr[-400:]}"
    # -amdgpu-mfma-vgpr-form: accumulators in VGPRs, and nothing spills under the cap
    for n, b in select(kernels, r"flash_attn_hh_kernel|flash_attn_f16_kernel").items():
        assert not count(b, "v_accvgpr_read") and not count(b, "v_accvgpr_write"), (n, "MFMA accumulators are copied through AGPRs again")
    for n, b in kernels.items():
        if not n.startswith("__") and count(b, "v_mfma_"):
            assert not [l for l in b if l.startswith("scratch_")], (n, "an MFMA kernel spills to scratch")
    # -amdgpu-kernarg-preload-count: plen / q / cache pointers of the balanced attention kernel and the single-row GEMVs arrive in SGPRs
    pre = kernels["__preload__"]
    hot = {n: v for n, v in pre.items() if re.search(r"attn_decode3_kernel|gemv_kernelI(f|DF16_)Li1E|outproj_merge_kernel", n)}
    assert hot and all(v >= 8 for v in hot.values()), ({n: v for n, v in hot.items() if v < 8}, "leading kernel arguments are not preloaded into SGPRs")


def test_gemm_hh_stream_kernel_shape(kernels):
    """k_gemm_stream.h: the loaders' LDS-DMA pieces are invisible to hipcc's wait bookkeeping, so the kernel must hold nothing that would
    shift the hand-counted vmcnt waits - no scratch - and the matrix waves' k-step must be the tight form the timeline was measured on:
    sixteen MFMAs fed by sixteen fragment reads with counted lgkmcnt waits (no lgkmcnt(0) drain between them) and ONE barrier."""
    for n, b in select(kernels, r"gemm_hh_stream_kernelILi5ELi0E").items():
        assert not [l for l in b if l.startswith("scratch_")], (n, "scratch access")
        assert count(b, "global_load_lds_dwordx4") >= 16, n
        for cnt in (16, 8, 0):
            assert [l for l in b if l.startswith(f"s_waitcnt vmcnt({cnt})")], (n, f"no counted vmcnt({cnt}) wait")
        mf = [i for i, l in enumerate(b) if l.startswith("v_mfma_f32_32x32x16_f16")]
        assert len(mf) == 16, (n, len(mf))
        # the k-step: from the barrier that closes the previous step to the one behind the last MFMA
        start = max(i for i, l in enumerate(b[:mf[0]]) if l.startswith("s_barrier"))
        loop = b[start:mf[-1] + 2]
        assert count(loop, "ds_read_b128") == 16, (n, count(loop, "ds_read_b128"))
        assert not [l for l in loop if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l], (n, "a full LDS drain inside the k-step")
