"""The C-ABI shared library loads and exports every symbol include/edgerunner_hip.h declares
(no compute is launched: this runs without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from edgerunner_amd import build, native
    build.build(verbose=False)          # cross-compiles for gfx950 when stale; no GPU needed
    return native.load_library()


def header_functions():
    src = open(os.path.join(ROOT, "include", "edgerunner_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(er_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from edgerunner_amd import native
    assert header_functions() == sorted(native.EXPORTS)


def test_all_symbols_exported(lib):
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in the header but not exported by the .so"


def test_trivial_calls(lib):
    from edgerunner_amd import native
    assert lib.er_abi_version() == 1
    names = [lib.er_kernel_kind_name(k).decode() for k in range(native.ER_NUM_KERNEL_KINDS)]
    assert names[:2] == ["qkv_gemv", "attn_decode"] and names[-1] == "sample_head"
    assert lib.er_kernel_kind_name(99).decode() == "?"
    assert isinstance(lib.er_last_error(), bytes)


def test_missing_library_fails_loudly(tmp_path):
    from edgerunner_amd import native
    with pytest.raises(native.NativeError, match="no CPU/eager fallback"):
        native.load_library(str(tmp_path / "nope.so"))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "edgerunner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "arae_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f
